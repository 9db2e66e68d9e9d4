#!/bin/bash
# round 6, GPU call A: (1) the exact-lattice tests of the two up-conv kernels, (2) the dominant launch on random / GELU-shaped / zero
# operands with the chip's power and clock sampled beside it (is the 0.53 a schedule limit or the power budget?), (3) the backbone
# counter pass BEFORE this round's kernel work, (4) the headline line and the API-level probe on this box.
O=gpurun_out/r06
mkdir -p $O
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
# power / clock beside a sustained run of the dominant shape (400 launches ~ 0.9 s per data mode)
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
# backbone counters, before
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
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
# headline + API level on this box
ESAM3_BENCH_PROFILE_OUT=$O/a_bench_per_launch.json timeout 400 python bench.py > $O/a_bench.json 2> $O/a_bench.err
tail -c 1500 $O/a_bench.json
timeout 300 python tools/api_level_probe.py > $O/a_api_probe.txt 2>&1
tail -15 $O/a_api_probe.txt | cut -c1-200
