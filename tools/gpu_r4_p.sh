#!/bin/bash
# round 4, GPU call P: the stage-1 training step at larger batches (is it launch-bound?) + its rocprofv3 kernel stats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
for b in 16 32; do timeout 600 python tools/bench_stage1_step.py --dtype bf16 --batch $b --steps 4 --warmup 2 2>/dev/null | tail -1 | tee $O/bench_stage1_step_b$b.json | cut -c1-330; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stage1 -o stage1 --output-format csv -- python $R/tools/bench_stage1_step.py --dtype bf16 --batch 8 --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find $O/prof_stage1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $O/stage1_step_kernel_stats.csv && head -12 $O/stage1_step_kernel_stats.csv | cut -c1-150
rm -rf $O/prof_stage1
