#!/bin/bash
# round 5, TinyViT students of the stage-1 trainer: LayerNorm / window-attention training kernels and layers against autograd, the
# training steps against the reference's own runs (5m, 11m with its DropPath factors), step timings, a kernel table of a step
O=gpurun_out/r05; mkdir -p $O
timeout 400 python -m pytest tests/test_train_tinyvit.py -q -m gpu -rP --timeout 300 > $O/tv_blocks.txt 2>&1; tail -3 $O/tv_blocks.txt | cut -c1-300
grep -h "^E  \|FAILED" $O/tv_blocks.txt | cut -c1-300 | head -40
timeout 400 python -m pytest tests/test_stage1_step.py -q -m gpu -rP --timeout 300 -k "tinyvit" > $O/tv_steps.txt 2>&1; tail -3 $O/tv_steps.txt | cut -c1-300
grep -h "^\[stage-1\|^  gradients\|^  parameters\|^  BatchNorm\|^E  " $O/tv_steps.txt | cut -c1-420 | head -40
for m in tiny_vit_11m tiny_vit_5m; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_$m.json 2>$O/bench_stage1_step_$m.err; tail -c 600 $O/bench_stage1_step_$m.json; tail -3 $O/bench_stage1_step_$m.err | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof/stage1_tv -o s --output-format csv -- python $R/tools/bench_stage1_step.py --model tiny_vit_11m --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/$O/prof/stage1_tv -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stage1_step_tiny_vit_11m_kernel_stats.csv 2>/dev/null
find $R/$O/prof -name "*_kernel_trace.csv" -size +8M -delete
head -14 $R/$O/stage1_step_tiny_vit_11m_kernel_stats.csv | cut -c1-150
