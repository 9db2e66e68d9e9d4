#!/bin/bash
# round 5: BatchNorm + activation in the same passes -- the fused-vs-separate test, every training test, step timings
O=gpurun_out/r05; mkdir -p $O
timeout 200 python -m pytest tests/test_train_blocks.py -q -m gpu --timeout 150 -k "bn_act" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_stage1.py tests/test_train_blocks.py tests/test_train_repvit.py tests/test_train_tinyvit.py tests/test_stage1_step.py -q -m gpu --timeout 400 2>&1 | tail -2
for m in b1 repvit_m1_1 tiny_vit_11m; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_${m}_bnact.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_${m}_bnact.json").read().strip().splitlines()[-1])
print("$m", d["value"], "images/s", d["ms_per_step"], "ms")
PY
done
timeout 200 python tools/bench_stage1_step.py --model b1 --batch 32 > $O/bench_stage1_step_b1_b32_bnact.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_b1_b32_bnact.json").read().strip().splitlines()[-1])
print("b1 b32", d["value"], "images/s", d["ms_per_step"], "ms")
PY
