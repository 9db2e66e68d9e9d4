#!/bin/bash
# Round 6: every gpurun call of the round as a subcommand (usage: gpurun -- "bash tools/gpu_r6.sh <letter>"); outputs under gpurun_out/r06/,
# the ones quoted in DESIGN.md copied to profiles/r06/ (profiles/README.md names the file -> subcommand).
O=gpurun_out/r06
mkdir -p $O
R=$GRAFT_REPO_ROOT
case "$1" in
a)
  # round 6, GPU call A: (1) the exact-lattice tests of the two up-conv kernels, (2) the dominant launch on random / GELU-shaped / zero
  # operands with the chip's power and clock sampled beside it (is the 0.53 a schedule limit or the power budget?), (3) the backbone
  # counter pass BEFORE this round's kernel work, (4) the headline line and the API-level probe on this box.
  # power / clock beside a sustained run of the dominant shape (400 launches ~ 0.9 s per data mode)
  # backbone counters, before
  # headline + API level on this box
  timeout 600 python -m pytest tests/test_lattice_gpu.py -q -m gpu -k "upconv" --timeout 500 > $O/a_lattice_upconv.txt 2>&1
  tail -5 $O/a_lattice_upconv.txt | cut -c1-300
  {
  for mode in random gelu zeros random; do
    echo "== operands: $mode"
    if [ $mode = random ]; then unset ESAM3_BENCH_DATA; else export ESAM3_BENCH_DATA=$mode; fi
    timeout 200 python tools/bench_gemm.py "up-conv,neck L0 3x3,head.3,ViT-H fc1" 2>&1 | grep -v "amdgpu.ids\|^lib\|^kernel"
  done
  unset ESAM3_BENCH_DATA
  } > $O/a_gemm_operands.txt 2>&1
  cat $O/a_gemm_operands.txt
  {
  for mode in random zeros; do
    if [ $mode = random ]; then unset ESAM3_BENCH_DATA; else export ESAM3_BENCH_DATA=$mode; fi
    ESAM3_BENCH_ITERS=1500 timeout 120 python tools/bench_gemm.py "up-conv" > $O/a_sustained_$mode.txt 2>&1 &
    BP=$!
    sleep 6
    for i in 1 2 3 4 5 6 7 8; do
      echo "-- $mode sample $i"; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | head -6
      sleep 0.4
    done
    wait $BP
    grep "up-conv" $O/a_sustained_$mode.txt
  done
  unset ESAM3_BENCH_DATA
  } > $O/a_power_clock.txt 2>&1
  tail -40 $O/a_power_clock.txt | cut -c1-200
  cd /tmp && export TMPDIR=/tmp
  P=$R/$O/pmc_before
  mkdir -p $P
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $P/sq1 -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace -d $P/sq2 -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R
  python tools/pmc_kernels.py --pass $P/sq1 --pass $P/sq2 --pass $P/fetch --pass $P/write \
    --match mbconv3s,mbconv3b,mla1,mla2d,kvprep,stem_dsconv,gemm256p,upconv_narrow,conv3x3_narrow,resize_shuffle --out $O/pmc_backbone_before.txt > /dev/null 2>&1
  head -60 $O/pmc_backbone_before.txt | cut -c1-260
  find $P -name "*.csv" -size +4M -delete
  ESAM3_BENCH_PROFILE_OUT=$O/a_bench_per_launch.json timeout 400 python bench.py > $O/a_bench.json 2> $O/a_bench.err
  tail -c 1500 $O/a_bench.json
  timeout 300 python tools/api_level_probe.py > $O/a_api_probe.txt 2>&1
  tail -15 $O/a_api_probe.txt | cut -c1-200

  ;;
b)
  # round 6, GPU call B: GPU timeline of API-level steps (where the device idles) + the double-buffered PCIe-inclusive leg
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- python $R/tools/api_timeline.py run > $R/$O/b_api_timeline_host.txt 2>&1
  cd $R
  grep "host step" $O/b_api_timeline_host.txt
  python tools/api_timeline.py report /tmp/tl > $O/b_api_timeline.txt 2>&1
  cat $O/b_api_timeline.txt | cut -c1-200
  timeout 400 python bench.py --no-cpu-baseline > $O/b_bench.json 2> $O/b_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/b_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s")})
PY

  ;;
c)
  # round 6, GPU call C: the API path after (1) the helper-thread staging of chunk 2 and (2) hand-backs started before any wait
  timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -k "batch or processor or api or host or alias or preprocess" --timeout 500 > $O/c_tests.txt 2>&1
  tail -4 $O/c_tests.txt | cut -c1-300
  timeout 300 python tools/api_level_probe.py 2>&1 | grep -v amdgpu > $O/c_api_probe.txt
  head -12 $O/c_api_probe.txt | cut -c1-200
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- python $R/tools/api_timeline.py run > $R/$O/c_api_timeline_host.txt 2>&1
  cd $R
  grep "host step" $O/c_api_timeline_host.txt | tail -3
  python tools/api_timeline.py report /tmp/tl > $O/c_api_timeline.txt 2>&1
  tail -30 $O/c_api_timeline.txt | cut -c1-160
  timeout 500 python bench.py --no-cpu-baseline > $O/c_bench.json 2> $O/c_bench.err
  tail -3 $O/c_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/c_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s") or "FAILED" in str(v)})
PY

  ;;
d)
  # round 6, GPU call D: the new parity pins (three students, scopes), then the whole GPU suite with its slowest tests listed
  timeout 900 python -m pytest tests/test_stage1_step.py tests/test_e2e_gpu.py tests/test_pcs.py -q -m gpu -rP --timeout 600 \
    -k "b0_training or repvit_m2_3 or tiny_vit_21m or profiler_scopes or (pcs_engine_vs_golden and bf16)" > $O/d_new_pins.txt 2>&1
  grep -E "^\[stage-1|gradients:|parameters after|BatchNorm running|passed|failed|^E " $O/d_new_pins.txt | cut -c1-330
  timeout 1500 python -m pytest tests -q -m gpu -x --durations=60 --timeout 900 > $O/d_full_suite.txt 2>&1
  tail -75 $O/d_full_suite.txt | cut -c1-200

  ;;
e)
  # round 6, GPU call E: idle time between the kernels of a headline step
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/e_bench_traced.json 2>/dev/null
  cd $R
  python tools/kernel_gaps.py /tmp/kt --out $O/e_kernel_gaps.txt | cut -c1-200

  ;;
f)
  # round 6, GPU call F: TinyViT MBConv on the mbconv3s design with GELU epilogues: op test, TinyViT parity tests, config-3 shard bench
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_students_gpu.py tests/test_bf16_distribution.py -q -m gpu -rP --timeout 500 \
    -k "mbconv3_gelu or tinyvit" > $O/f_tinyvit_tests.txt 2>&1
  grep -E "passed|failed|^E |\[dist" $O/f_tinyvit_tests.txt | cut -c1-300
  ESAM3_BENCH_PROFILE_OUT=$O/f_bench_tinyvit_per_launch.json timeout 400 python bench.py --backbone tinyvit --model 11m --no-cpu-baseline > $O/f_bench_tinyvit.json 2> $O/f_bench_tinyvit.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/f_bench_tinyvit.json").read().strip().splitlines()[-1])
print("tinyvit-11m", d["value"], d["ms_per_step"])
p = json.load(open("gpurun_out/r06/f_bench_tinyvit_per_launch.json"))["per_tag"]
for r in sorted(p, key=lambda r: -r["ms"])[:12]:
    print(f"  {r['ms']:.3f} x{r['launches']} {r['tag'][:90]}")
PY

  ;;
g)
  ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 timeout 300 python tools/evit_fused_bench.py tv.0 tv.0:v2 tv.hs s1.1 2>&1 | grep -v amdgpu | tee $O/g_tv_mbconv_ab.txt

  ;;
h)
  # round 6, GPU call H: host worker threads of the API path (PIL staging + mask widening): 8 / 16 / 32 / 64
  for n in 16 8 32 64 16; do
    echo "== ESAM3_HOST_THREADS=$n"
    ESAM3_HOST_THREADS=$n timeout 200 python tools/api_level_probe.py 2>&1 | grep -E "api step, results|keep previous|rep 1|stage 32 PIL images, rgbx np" | cut -c1-220
  done > $O/h_api_host_threads.txt 2>&1
  cat $O/h_api_host_threads.txt

  ;;
i)
  # round 6, GPU call I: per-operator roofline of a B1 batch-32 training step (+ the throughput lines of three students)
  timeout 400 python tools/stage1_step_roofline.py --model b1 --batch 32 2>&1 | grep -v amdgpu > $O/i_roofline_stage1_step_b1_b32.md
  head -40 $O/i_roofline_stage1_step_b1_b32.md | cut -c1-220
  timeout 300 python tools/bench_stage1_step.py --model b1 --batch 32 --steps 5 2>/dev/null | tail -1 > $O/i_bench_stage1_step_b1_b32.json
  python -c "
import json; d=json.loads(open('$O/i_bench_stage1_step_b1_b32.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"

  ;;
j)
  # round 6, GPU call J: two-caller test, chunked window-attention backward, per-layer SyncBatchNorm setting, training steps after those changes
  timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_train_tinyvit.py tests/test_stage1_step.py tests/test_dist_gloo.py -q -m gpu --timeout 600 \
    -k "two_callers or window_attention or training_step or rccl or tinyvit_block" > $O/j_tests.txt 2>&1
  tail -6 $O/j_tests.txt | cut -c1-300

  ;;
k)
  # round 6, GPU call K: does a power-bound GEMM launch need all 256 CUs?  (dev library, persistent grid capped)
  for g in 256 248 240 224 208 192 160 128 256; do
    echo "== ESAM3_P_GRID=$g"
    ESAM3_P_GRID=$g ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_BENCH_ITERS=40 timeout 120 python tools/bench_gemm.py "up-conv,neck L1 3x3,head.3" 2>&1 | grep "ms"
  done > $O/k_gemm_grid_cap.txt 2>&1
  cat $O/k_gemm_grid_cap.txt

  ;;
l)
  for g in 0 224 192 160 128; do
    ESAM3_LIB=build_dev/libesam3_dev.so ESAM3_P_GRID=$g timeout 300 python tools/two_stream_probe.py 2>&1 | grep "streams="
  done > $O/l_two_stream_probe.txt 2>&1
  cat $O/l_two_stream_probe.txt | cut -c1-120

  ;;
m)
  ( time python bench.py > $O/m_bench.json 2> $O/m_bench.err ) 2>&1 | grep real
  tail -2 $O/m_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/m_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s") or "FAILED" in str(v)})
PY

  ;;
al)
  # round 6, GPU call AL: kernel census of a RepViT-M1.1 and a TinyViT-11M training step
  mkdir -p $O
  R=$GRAFT_REPO_ROOT
  for m in repvit_m1_1 tiny_vit_11m; do
    rm -rf /tmp/al_prof
    ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/al_prof -o al --output-format csv -- python $R/tools/bench_stage1_step.py --model $m --batch 32 --steps 5 --warmup 2 > /dev/null 2>&1 )
    cp $(find /tmp/al_prof -name "*kernel_stats.csv" | head -1) $O/al_kernel_stats_stage1_step_${m}_b32.csv
  done
  ;;
ac)
  # round 6, GPU call AC: the training step after the gradient copies became one multi-tensor copy: tests, the B1 step, the kernel census of a step
  mkdir -p $O
  [ -n "$AC_SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_train_blocks.py tests/test_stage1_step.py tests/test_stage1.py -q -m gpu -x --timeout 600 > $O/ac_train_tests.txt 2>&1
  tail -3 $O/ac_train_tests.txt | cut -c1-300
  for m in b1 repvit_m1_1 tiny_vit_11m; do
    timeout 300 python tools/bench_stage1_step.py --model $m --batch 32 --steps 5 2>/dev/null | tail -1 > $O/ac_bench_stage1_step_${m}_b32.json
    python -c "import json; d=json.loads(open('$O/ac_bench_stage1_step_${m}_b32.json').read()); print('$m', d['value'], d['ms_per_step'])"
  done
  R=$GRAFT_REPO_ROOT
  ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/ac_prof -o ac --output-format csv -- python $R/tools/bench_stage1_step.py --model b1 --batch 32 --steps 5 --warmup 2 > $R/$O/ac_prof_stdout.txt 2>&1 )
  cp $(find /tmp/ac_prof -name "*kernel_stats.csv" | head -1) $O/ac_kernel_stats_stage1_step_b1_b32.csv
  python - <<PY
import csv, glob
f = glob.glob("/tmp/ac_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# the last 40 % of the trace = timed steps; busy fraction and the largest gaps there
n = len(rows); part = rows[int(n * 0.6):]
busy = sum(e - s for s, e, _ in part); span = part[-1][1] - part[0][0]
print("kernels", len(part), "busy ms", busy / 1e6, "span ms", span / 1e6, "busy frac", busy / span)
gaps = sorted(((part[i + 1][0] - part[i][1], part[i][2][:60], part[i + 1][2][:60]) for i in range(len(part) - 1)), reverse=True)[:25]
for g in gaps: print(g)
rr = list(csv.DictReader(open(f)))
sel = [r for r in rr if "bn_map_kernel<1, true" in r["Kernel_Name"]]
sel = sel[-35:]
print("bn_map BWD calls of the last step: (us, grid, kernel)")
for r in sel: print(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][40:75])
PY
  head -40 $O/ac_kernel_stats_stage1_step_b1_b32.csv | cut -c1-150
  ;;
p)
  # round 6, GPU call P: training kernels after the round's changes (stride-2 depthwise dgrad with LDS weights, vectorised colsum, stem im2col
  # kernel, depthwise wgrad without per-pixel divisions): their tests, the step pins of all students, the per-operator table again
  timeout 900 python -m pytest tests/test_train_blocks.py tests/test_train_repvit.py tests/test_train_tinyvit.py tests/test_stage1_step.py tests/test_stage1.py -q -m gpu -x --timeout 600 > $O/p_train_tests.txt 2>&1
  tail -5 $O/p_train_tests.txt | cut -c1-300
  timeout 400 python tools/stage1_step_roofline.py --model b1 --batch 32 --calls 40 2>&1 | grep -v amdgpu > $O/p_roofline_stage1_step_b1_b32.md
  head -32 $O/p_roofline_stage1_step_b1_b32.md | cut -c1-200
  for m in b1 repvit_m1_1 tiny_vit_11m; do
    timeout 300 python tools/bench_stage1_step.py --model $m --batch 32 --steps 5 2>/dev/null | tail -1 > $O/p_bench_stage1_step_${m}_b32.json
    python -c "import json; d=json.loads(open('$O/p_bench_stage1_step_${m}_b32.json').read()); print('$m', d['value'], d['ms_per_step'])"
  done
  ;;
q)
  # round 6, GPU call Q: phase ablation of mla1 (dev library; ESAM3_MLA1_ABL bits: 1 expand, 2 depthwise, 4 grouped, 8 relu(q) stores, 16 kv partials)
  export ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=50
  {
  for abl in 0 1 2 4 8 16 24 31 0; do
    echo "== ESAM3_MLA1_ABL=$abl"
    ESAM3_MLA1_ABL=$abl timeout 120 python tools/evit_fused_bench.py s2.ctx s3.ctx 2>&1 | grep op_timed
  done
  } > $O/q_mla1_ablation.txt 2>&1
  cat $O/q_mla1_ablation.txt
  ;;
r)
  # round 6, GPU call R: mla1v (depthwise + grouped conv in registers) -- exactness tests, then A/B against the round-4 kernel
  timeout 600 python -m pytest tests/test_lattice_gpu.py tests/test_ops_gpu.py -q -m gpu -k "mla or lite" --timeout 500 > $O/r_mla_tests.txt 2>&1
  tail -5 $O/r_mla_tests.txt | cut -c1-300
  export ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=50
  {
  for old in 1 0 1 0; do
    echo "== ESAM3_MLA1_OLD=$old"
    ESAM3_MLA1_OLD=$old timeout 120 python tools/evit_fused_bench.py s2.ctx s3.ctx 2>&1 | grep op_timed
  done
  for abl in 1 2 4 8 16 31; do
    echo "== new kernel, ESAM3_MLA1_ABL=$abl"
    ESAM3_MLA1_ABL=$abl timeout 120 python tools/evit_fused_bench.py s2.ctx s3.ctx 2>&1 | grep op_timed
  done
  } > $O/r_mla1_ab.txt 2>&1
  cat $O/r_mla1_ab.txt
  ;;
s)
  # round 6, GPU call S: persistent input-stem kernel -- op test (bit-equal to the round-2 kernel), then timing per workgroups-per-CU setting
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "stem" --timeout 500 > $O/s_stem_tests.txt 2>&1
  tail -5 $O/s_stem_tests.txt | cut -c1-300
  export ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=30
  {
  for wgs in 2 3 4 5; do
    echo "== ESAM3_STEM_WGS=$wgs (variant 0 = persistent, 1 = one tile per workgroup)"
    ESAM3_STEM_WGS=$wgs timeout 120 python tools/stem_bench.py 0 1 2>&1 | grep op_timed
  done
  } > $O/s_stem_ab.txt 2>&1
  cat $O/s_stem_ab.txt
  ;;
t)
  # round 6, GPU call T: mla1v + the 1024-thread kvprep + the persistent stem in the engine: exactness tests, the expand-write order A/B
  # of the fused MBConvs (flip by pixel bit 2 instead of bit 1), the headline line with its per-launch table
  timeout 900 python -m pytest tests/test_lattice_gpu.py tests/test_ops_gpu.py -q -m gpu -k "mla or lite or stem or mbconv" --timeout 500 > $O/t_tests.txt 2>&1
  tail -3 $O/t_tests.txt | cut -c1-300
  timeout 900 python -m pytest tests/test_e2e_gpu.py -q -m gpu -x --timeout 800 > $O/t_e2e.txt 2>&1
  tail -3 $O/t_e2e.txt | cut -c1-300
  export ESAM3_OP_REPEAT=50
  {
  for lib in dev flip4 dev flip4; do
    echo "== build_dev/libesam3_$lib.so"
    ESAM3_DEV_LIB=build_dev/libesam3_$lib.so timeout 200 python tools/evit_fused_bench.py s0.1 s1.1 s2.loc s3.loc s2.ctx s3.ctx 2>&1 | grep op_timed
  done
  } > $O/t_flip_ab.txt 2>&1
  cat $O/t_flip_ab.txt
  unset ESAM3_OP_REPEAT
  ESAM3_BENCH_PROFILE_OUT=$O/t_bench_per_launch.json timeout 400 python bench.py > $O/t_bench.json 2> $O/t_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/t_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s")})
PY
  python tools/roofline_table.py $O/t_bench_per_launch.json > $O/t_roofline_headline.md 2>/dev/null; head -40 $O/t_roofline_headline.md | cut -c1-160
  ;;
u)
  # round 6, GPU call U: the stride-2 blocks of stages 3 / 4 on the 8-wave persistent kernel (mbconv3b<S = 2>): lattice + op tests, A/B vs the generic kernel
  timeout 900 python -m pytest tests/test_lattice_gpu.py tests/test_ops_gpu.py -q -m gpu -k "mbconv" --timeout 500 > $O/u_tests.txt 2>&1
  tail -3 $O/u_tests.txt | cut -c1-300
  export ESAM3_OP_REPEAT=50 ESAM3_DEV_LIB=build_dev/libesam3_dev.so
  {
  for gen in 1 0 1 0; do
    echo "== ESAM3_MB3_GENERIC=$gen"
    ESAM3_MB3_GENERIC=$gen timeout 200 python tools/evit_fused_bench.py s2.0 s3.0 2>&1 | grep op_timed
  done
  } > $O/u_mb3b_s2_ab.txt 2>&1
  cat $O/u_mb3b_s2_ab.txt
  ;;
v)
  # round 6, GPU call V: A/B of the 64 -> 256 -> 64 stride-1 MBConv: 4-wave mbconv3s (two per CU) vs the 8-wave LDS-weight kernel mbconv3b<64> (one / two per CU)
  export ESAM3_OP_REPEAT=50 ESAM3_DEV_LIB=build_dev/libesam3_dev.so
  timeout 600 python -m pytest tests/test_lattice_gpu.py -q -m gpu -k "mbconv3" --timeout 500 > $O/v_tests_default.txt 2>&1; tail -1 $O/v_tests_default.txt
  {
  echo "== mbconv3s"; timeout 200 python tools/evit_fused_bench.py s1.1 tv.hs 2>&1 | grep op_timed
  for grid in 256 512; do
    echo "== mbconv3b<64>, grid $grid"
    ESAM3_MB3B_64=1 ESAM3_MB3B_GRID=$grid timeout 200 python tools/evit_fused_bench.py s1.1 tv.hs 2>&1 | grep op_timed
  done
  echo "== mbconv3s"; timeout 200 python tools/evit_fused_bench.py s1.1 tv.hs 2>&1 | grep op_timed
  } > $O/v_mb3b64_ab.txt 2>&1
  cat $O/v_mb3b64_ab.txt
  ;;
w)
  # round 6, GPU call W: the narrow MBConvs (Cin 16 / 32) on the 8-wave LDS-weight kernel at two workgroups per CU: exactness through the dev library, A/B vs mbconv3s
  export ESAM3_DEV_LIB=build_dev/libesam3_dev.so
  ESAM3_MB3B_SMALL=1 ESAM3_MB3B_64=1 ESAM3_MB3B_GRID=512 timeout 300 python tools/mbconv_variant_check.py 2>&1 | grep -v amdgpu > $O/w_variant_check.txt
  cat $O/w_variant_check.txt
  export ESAM3_OP_REPEAT=50
  {
  echo "== mbconv3s"; timeout 200 python tools/evit_fused_bench.py s0.0 s0.1 s1.0 s1.1 2>&1 | grep op_timed
  for grid in 256 512 768; do
    echo "== mbconv3b, grid $grid"
    ESAM3_MB3B_SMALL=1 ESAM3_MB3B_64=1 ESAM3_MB3B_GRID=$grid timeout 200 python tools/evit_fused_bench.py s0.0 s0.1 s1.0 s1.1 2>&1 | grep op_timed
  done
  echo "== mbconv3s"; timeout 200 python tools/evit_fused_bench.py s0.0 s0.1 s1.0 s1.1 2>&1 | grep op_timed
  } > $O/w_mb3b_small_ab.txt 2>&1
  cat $O/w_mb3b_small_ab.txt
  ;;
x)
  # round 6, GPU call X: every fused MBConv on the 8-wave LDS-weight kernel (narrow ones two per CU, TinyViT's GELU variant included): op / lattice /
  # student / e2e tests, headline + TinyViT-11M lines with their per-launch tables
  timeout 1200 python -m pytest tests/test_lattice_gpu.py tests/test_ops_gpu.py tests/test_students_gpu.py tests/test_e2e_gpu.py -q -m gpu --timeout 900 > $O/x_tests.txt 2>&1
  tail -4 $O/x_tests.txt | cut -c1-300
  ESAM3_BENCH_PROFILE_OUT=$O/x_bench_per_launch.json timeout 400 python bench.py > $O/x_bench.json 2> $O/x_bench.err
  ESAM3_BENCH_PROFILE_OUT=$O/x_bench_tinyvit_per_launch.json timeout 400 python bench.py --backbone tinyvit --model 11m --no-cpu-baseline > $O/x_bench_tinyvit.json 2> $O/x_bench_tinyvit.err
  python - <<'PY'
import json
for f in ("x_bench", "x_bench_tinyvit"):
    d = json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s")})
PY
  python tools/roofline_table.py $O/x_bench_per_launch.json > $O/x_roofline_headline.md 2>/dev/null; head -34 $O/x_roofline_headline.md | cut -c1-150
  python tools/roofline_table.py $O/x_bench_tinyvit_per_launch.json > $O/x_roofline_tinyvit_11m.md 2>/dev/null; head -12 $O/x_roofline_tinyvit_11m.md | cut -c1-150
  ;;
y)
  # round 6, GPU call Y: the backbone counter passes BEFORE / AFTER this round's kernel work under identical conditions: the dev library (same sources +
  # environment switches), headline step only (batch 32), 12 steps per pass; "before" = the round-5 dispatch restored by the switches
  # (ESAM3_MLA1_OLD, ESAM3_MB3S, ESAM3_STEM_OLD; the kvprep change cannot be switched back)
  cd /tmp && export TMPDIR=/tmp
  export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so
  for which in before after; do
    if [ $which = before ]; then export ESAM3_MLA1_OLD=1 ESAM3_MB3S=1 ESAM3_STEM_OLD=1; else unset ESAM3_MLA1_OLD ESAM3_MB3S ESAM3_STEM_OLD; fi
    P=$R/$O/pmc_$which
    rm -rf $P; mkdir -p $P
    CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only"
    rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $P/sq1 -o a --output-format csv -- $CMD > /dev/null 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace -d $P/sq2 -o b --output-format csv -- $CMD > /dev/null 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -o f --output-format csv -- $CMD > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -o w --output-format csv -- $CMD > /dev/null 2>&1
    (cd $R && python tools/pmc_kernels.py --pass $P/sq1 --pass $P/sq2 --pass $P/fetch --pass $P/write \
      --match mbconv3s,mbconv3b,mbconv3_,mla1,mla2d,kvprep,stem_dsconv --out $O/pmc_backbone_${which}_b32.txt > /dev/null 2>&1)
    echo "== $which"; grep -A2 "dispatches per pass" $R/$O/pmc_backbone_${which}_b32.txt | grep -v "^    [A-Z]" | grep -v "^--" | cut -c1-250
    find $P -name "*.csv" -size +2M -delete
  done
  ;;
z)
  # round 6, GPU call Z: is the wide MBConv (Cin 128 / 256) waiting for its weight DMA?  ESAM3_MB3_ABL=32 stops the stream after the first chunk (timing only)
  export ESAM3_OP_REPEAT=50 ESAM3_DEV_LIB=build_dev/libesam3_dev.so
  {
  for abl in 0 32 0 32; do
    echo "== ESAM3_MB3_ABL=$abl"
    ESAM3_MB3_ABL=$abl timeout 200 python tools/evit_fused_bench.py s2.loc s3.loc s2.0 s3.0 s1.1 2>&1 | grep op_timed
  done
  } > $O/z_mb3b_dma_abl.txt 2>&1
  cat $O/z_mb3b_dma_abl.txt
  ;;
aa)
  # round 6, GPU call AA: expand units split (pixel tile, channel half) over the 8 waves of mbconv3b: exactness + timing per threshold (0 = whole tiles everywhere)
  export ESAM3_OP_REPEAT=50
  {
  for v in u0 u16 u32 u64 u0; do
    echo "== build_dev/libesam3_$v.so"
    ESAM3_DEV_LIB=build_dev/libesam3_$v.so timeout 300 python tools/mbconv_variant_check.py 2>&1 | grep -E "ALL EXACT|DIFFER|differ" | grep -v " 0 of" 
    ESAM3_DEV_LIB=build_dev/libesam3_$v.so timeout 200 python tools/evit_fused_bench.py s0.0 s0.1 s1.0 s1.1 s2.0 tv.0 2>&1 | grep op_timed
  done
  } > $O/aa_units_ab.txt 2>&1
  cat $O/aa_units_ab.txt
  ;;
ab)
  # round 6, GPU call AB: this round's backbone kernels vs the ones they replace, bit for bit on real data (dev library switches), then the PCS
  # bf16 distribution test that moved (x1.33 -> x1.53 on one box) with the order-preserving kvprep
  ESAM3_DEV_LIB=build_dev/libesam3_dev.so timeout 600 python tools/ab_bitcompare.py 2>&1 | grep -v amdgpu > $O/ab_bitcompare.txt
  cat $O/ab_bitcompare.txt
  timeout 900 python -m pytest tests/test_pcs.py -q -m gpu -k "distribution" -rP --timeout 800 2>&1 | grep -E "dist pcs|passed|failed" > $O/ab_pcs_dist.txt
  grep -E "dog|passed|failed" $O/ab_pcs_dist.txt | cut -c1-200
  ;;
final)
  # round 6, final GPU call: the whole GPU suite, smoke(), the default bench line, the other configurations' lines
  timeout 1300 python -m pytest tests/ -q -m gpu --durations=8 > $O/final_suite.txt 2>&1; tail -14 $O/final_suite.txt | cut -c1-200
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.txt 2>&1; tail -2 $O/final_smoke.txt | cut -c1-300
  ESAM3_BENCH_PROFILE_OUT=$O/final_bench_per_launch.json timeout 400 python bench.py > $O/final_bench.json 2> $O/final_bench.err
  ESAM3_BENCH_PROFILE_OUT=$O/final_bench_tinyvit_per_launch.json timeout 400 python bench.py --backbone tinyvit --model 11m --no-cpu-baseline > $O/final_bench_tinyvit.json 2> $O/final_bench_tinyvit.err
  timeout 600 python bench.py --workload text --backbone sam3 --model vit_h --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/final_bench_text_cfg4.json 2> $O/final_bench_text_cfg4.err
  timeout 300 python tools/bench_stage1_step.py --model b1 --batch 32 --steps 5 2>/dev/null | tail -1 > $O/final_bench_stage1_step_b1_b32.json
  python - <<'PY'
import json
for f in ("final_bench", "final_bench_tinyvit", "final_bench_text_cfg4", "final_bench_stage1_step_b1_b32"):
    try:
        d = json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k: v for k, v in d.get("config", {}).items() if k.endswith("images_per_s")})
    except Exception as e:
        print(f, "FAILED", e)
PY
  python tools/roofline_table.py $O/final_bench_per_launch.json > $O/final_roofline_headline.md 2>/dev/null
  python tools/roofline_table.py $O/final_bench_tinyvit_per_launch.json --merge-layers > $O/final_roofline_tinyvit_11m.md 2>/dev/null
  ;;
*) echo "usage: $0 {a..z aa ab final}"; exit 2 ;;
esac
