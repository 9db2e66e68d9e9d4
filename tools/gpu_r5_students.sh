#!/bin/bash
# round 5, all nine stage-1 students: the TinyViT attention kernels after the phase split, every training-step test, the other stage-1 /
# training-block tests (regression: kernels_train.hip changed), smoke(), step timings of every student family, a kernel table of a TinyViT step
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_train_tinyvit.py tests/test_train_repvit.py -q -m gpu --timeout 250 2>&1 | tail -2
timeout 500 python -m pytest tests/test_stage1_step.py tests/test_stage1.py tests/test_train_blocks.py -q -m gpu -rP --timeout 400 > $O/students_steps.txt 2>&1; tail -3 $O/students_steps.txt | cut -c1-300
grep -h "^E  \|FAILED" $O/students_steps.txt | cut -c1-300 | head -20
grep -h "^\[stage-1 tiny\|^\[stage-1 rep" $O/students_steps.txt | cut -c1-250 | head -30
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
for m in tiny_vit_5m tiny_vit_11m tiny_vit_21m repvit_m2_3 b2 b0; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_$m.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_$m.json").read().strip().splitlines()[-1])
print("$m", d["value"], "images/s", d["ms_per_step"], "ms", d["roofline"]["achieved"], "TFLOP/s", d["config"]["forward_gflop_per_image"], "GFLOP/image fwd")
PY
done
timeout 200 python tools/bench_stage1_step.py --model tiny_vit_11m --batch 32 > $O/bench_stage1_step_tiny_vit_11m_b32.json 2>/dev/null; tail -c 300 $O/bench_stage1_step_tiny_vit_11m_b32.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof/stage1_tv2 -o s --output-format csv -- python $R/tools/bench_stage1_step.py --model tiny_vit_11m --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/$O/prof/stage1_tv2 -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stage1_step_tiny_vit_11m_kernel_stats.csv 2>/dev/null
find $R/$O/prof -name "*_kernel_trace.csv" -delete
head -8 $R/$O/stage1_step_tiny_vit_11m_kernel_stats.csv | cut -c1-150
