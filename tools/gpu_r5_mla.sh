#!/bin/bash
# round 5: the LiteMLA token passes with four threads per token and no indexed register arrays (dim 32 spilled 12 KB per thread): the
# kernel tests, the EfficientViT training-step tests, step timings of B0 / B1 / B2, kernel table of a B2 step
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_train_blocks.py -q -m gpu --timeout 250 -k "mla or efficientvit_block or trunk_train" 2>&1 | tail -2
timeout 400 python -m pytest tests/test_stage1_step.py -q -m gpu -rP --timeout 300 -k "not repvit and not tinyvit" > $O/mla_steps.txt 2>&1; tail -2 $O/mla_steps.txt | cut -c1-200
grep -h "^\[stage-1\|^  gradients\|^E  " $O/mla_steps.txt | cut -c1-300 | head -12
for m in b0 b1 b2; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_${m}_mla4.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_${m}_mla4.json").read().strip().splitlines()[-1])
print("$m", d["value"], "images/s", d["ms_per_step"], "ms", d["roofline"]["achieved"], "TFLOP/s")
PY
done
timeout 200 python tools/bench_stage1_step.py --model b1 --batch 32 > $O/bench_stage1_step_b1_b32_mla4.json 2>/dev/null; tail -c 260 $O/bench_stage1_step_b1_b32_mla4.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof/stage1_b2b -o s --output-format csv -- python $R/tools/bench_stage1_step.py --model b2 --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/$O/prof/stage1_b2b -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stage1_step_b2_kernel_stats_mla4.csv 2>/dev/null
find $R/$O/prof -name "*_kernel_trace.csv" -delete
head -8 $R/$O/stage1_step_b2_kernel_stats_mla4.csv | cut -c1-170
