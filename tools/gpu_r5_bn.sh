#!/bin/bash
# round 5: the BatchNorm kernels after the split entry points / 16-wave finalize / 1/n as a kernel argument -- every test that runs them,
# the SyncBatchNorm path on a one-rank RCCL group, step timings
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_stage1.py tests/test_train_blocks.py tests/test_train_repvit.py tests/test_train_tinyvit.py tests/test_stage1_step.py -q -m gpu --timeout 400 2>&1 | tail -2
timeout 300 python -m pytest tests/test_dist_gloo.py -q -m gpu -rP --timeout 250 -k "sync_batchnorm or one_rank or allreducer" > $O/bn_sync_tests.txt 2>&1; tail -2 $O/bn_sync_tests.txt; grep -h "sync-bn one rank\|^E  " $O/bn_sync_tests.txt | cut -c1-300
for m in b1 repvit_m1_1 tiny_vit_11m; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_${m}_bn16.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_${m}_bn16.json").read().strip().splitlines()[-1])
print("$m", d["value"], "images/s", d["ms_per_step"], "ms")
PY
done
timeout 200 python tools/bench_stage1_step.py --model b1 --batch 32 > $O/bench_stage1_step_b1_b32_bn16.json 2>/dev/null; tail -c 200 $O/bench_stage1_step_b1_b32_bn16.json
