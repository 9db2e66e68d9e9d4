#!/bin/bash
# round 5: the 3x3 conv weight gradient as one gathered launch (head conv, patch-embedding conv) -- its test, every training test, step timings,
# then the headline bench line + smoke as the last sanity of the round's library
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_train_repvit.py -q -m gpu --timeout 250 -k "conv3x3" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_stage1.py tests/test_train_blocks.py tests/test_train_repvit.py tests/test_train_tinyvit.py tests/test_stage1_step.py -q -m gpu --timeout 400 2>&1 | tail -2
for m in b1 repvit_m1_1 tiny_vit_11m; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_${m}_wgrad9.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_${m}_wgrad9.json").read().strip().splitlines()[-1])
print("$m", d["value"], "images/s", d["ms_per_step"], "ms")
PY
done
timeout 200 python tools/bench_stage1_step.py --model b1 --batch 32 > $O/bench_stage1_step_b1_b32_wgrad9.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_stage1_step_b1_b32_wgrad9.json").read().strip().splitlines()[-1])
print("b1 b32", d["value"], "images/s", d["ms_per_step"], "ms")
PY
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_last_sanity.json; python - <<PY
import json
d = json.loads(open("$O/bench_last_sanity.json").read())
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["api_level_images_per_s"])
PY
