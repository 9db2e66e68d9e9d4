#!/bin/bash
# round 5, GPU call G: B2 training trunk localisation, the token-split LiteMLA backward + side-stream H2D through their tests, stage-1 step
# timings + kernel table, API A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
timeout 300 python tools/trunk_train_layer_diff.py b2 > $O/g_layer_diff_b2.txt 2>&1; grep "fwd\|bwd" $O/g_layer_diff_b2.txt | awk '{print $2, $(NF)}' | head -70
timeout 500 python -m pytest tests/test_train_blocks.py tests/test_stage1_step.py tests/test_stage1.py tests/test_e2e_gpu.py tests/test_facade_eval_coco.py -q -m gpu --timeout 400 \
  -k "not test_b2 and (train or stage1 or batch or release or eval_coco or mla or dwconv or step or update or paired)" > $O/g_tests.txt 2>&1; tail -4 $O/g_tests.txt | cut -c1-200
timeout 150 python tools/bench_stage1_step.py > $O/g_stage1_step.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/g_stage1_step.json').read().strip().splitlines()[-1]); print('stage1 b8', d['value'], d['ms_per_step'])"
timeout 150 python tools/bench_stage1_step.py --batch 32 > $O/g_stage1_step_b32.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/g_stage1_step_b32.json').read().strip().splitlines()[-1]); print('stage1 b32', d['value'], d['ms_per_step'])"
timeout 200 python tools/api_ab.py > $O/g_api_ab.txt 2>&1; grep "round 1" $O/g_api_ab.txt
timeout 200 python bench.py --no-cpu-baseline > $O/g_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/g_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'api', d['config']['api_level_images_per_s'], 'pcie', d['config']['pcie_inclusive_images_per_s'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof/stage1 -o s --output-format csv -- python $R/tools/bench_stage1_step.py --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $O/prof/stage1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/stage1_step_kernel_stats.csv 2>/dev/null
find $O/prof -name "*_kernel_trace.csv" -size +8M -delete
head -14 $O/stage1_step_kernel_stats.csv | cut -c1-150
