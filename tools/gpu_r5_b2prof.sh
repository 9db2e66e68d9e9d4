#!/bin/bash
# round 5: rocprofv3 --stats of an EfficientViT-B2 training step (133 ms at batch 8: out of line with its FLOPs)
O=gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof/stage1_b2 -o s --output-format csv -- python $R/tools/bench_stage1_step.py --model b2 --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/$O/prof/stage1_b2 -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stage1_step_b2_kernel_stats.csv 2>/dev/null
find $R/$O/prof -name "*_kernel_trace.csv" -delete
head -16 $R/$O/stage1_step_b2_kernel_stats.csv | cut -c1-200
