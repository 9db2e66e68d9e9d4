#!/bin/bash
# round 4, GPU call M: training-block tests with the autocast yardstick + the real-shape trunk test, the stage-1 step bench + its kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
timeout 1200 python -m pytest tests/test_train_blocks.py -q -s -k "mbconv_block or efficientvit_block or trunk_train" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-400 | tee $O/tests_m.txt
timeout 600 python tools/bench_stage1_step.py --dtype bf16 2>/dev/null | tail -1 | tee $O/bench_stage1_step.json
timeout 600 python tools/bench_stage1_step.py --dtype f32 2>/dev/null | tail -1 | tee $O/bench_stage1_step_f32.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stage1 -o stage1 -- python $R/tools/bench_stage1_step.py --dtype bf16 --steps 2 --warmup 1 > /dev/null 2>&1
f=$(ls $O/prof_stage1/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" > $O/stage1_step_kernel_stats.csv
rm -rf $O/prof_stage1
