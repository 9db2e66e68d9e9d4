#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
( time python bench.py > $O/m_bench.json 2> $O/m_bench.err ) 2>&1 | grep real
tail -2 $O/m_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/m_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s") or "FAILED" in str(v)})
PY
