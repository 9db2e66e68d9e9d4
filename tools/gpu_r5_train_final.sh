#!/bin/bash
# round 5, the training path after the wgrad split choice (M, N, K) and the residual additions on esam3_channel_scale: every training test,
# step timings of one student per family (+ batch 32), kernel table of a B1 step
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_train_blocks.py tests/test_train_repvit.py tests/test_train_tinyvit.py tests/test_stage1_step.py tests/test_stage1.py -q -m gpu -rP --timeout 400 > $O/train_final_tests.txt 2>&1
tail -2 $O/train_final_tests.txt | cut -c1-200; grep -h "^E  \|FAILED" $O/train_final_tests.txt | cut -c1-300 | head -20
for m in b1 b2 b0 repvit_m0_9 repvit_m1_1 repvit_m2_3 tiny_vit_5m tiny_vit_11m tiny_vit_21m; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_${m}_final.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_${m}_final.json").read().strip().splitlines()[-1])
print("$m", d["value"], "images/s", d["ms_per_step"], "ms", d["roofline"]["achieved"], "TFLOP/s")
PY
done
for m in b1 repvit_m1_1 tiny_vit_11m; do
  timeout 200 python tools/bench_stage1_step.py --model $m --batch 32 > $O/bench_stage1_step_${m}_b32_final.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_${m}_b32_final.json").read().strip().splitlines()[-1])
print("$m b32", d["value"], "images/s", d["ms_per_step"], "ms", d["roofline"]["achieved"], "TFLOP/s")
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof/stage1_b1f -o s --output-format csv -- python $R/tools/bench_stage1_step.py --model b1 --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/$O/prof/stage1_b1f -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stage1_step_b1_kernel_stats_final.csv 2>/dev/null
find $R/$O/prof -name "*_kernel_trace.csv" -delete
head -10 $R/$O/stage1_step_b1_kernel_stats_final.csv | cut -c1-170
