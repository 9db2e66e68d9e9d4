#!/bin/bash
# headline bench with the per-launch table; prints the bench line and the launches matching $1 (regex) or the top 25
mkdir -p gpurun_out/r02
ESAM3_BENCH_PROFILE_OUT=gpurun_out/r02/headline_per_launch.json timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>gpurun_out/r02/headline.err | tail -1 > gpurun_out/r02/headline.json
python - "$1" <<'P'
import json, re, sys
b = json.load(open("gpurun_out/r02/headline.json"))
print("value", b["value"], "ms_per_step", b["ms_per_step"], "roofline", b["roofline"].get("achieved"), b["roofline"].get("frac"))
d = json.load(open("gpurun_out/r02/headline_per_launch.json"))
rows = sorted(d["per_tag"], key=lambda r: -r["ms"])
print("total ms", round(sum(r["ms"] for r in rows), 3))
pat = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else None
for r in rows if pat else rows[:25]:
    if pat and not re.search(pat, r["tag"]): continue
    print(f'{r["ms"]:8.4f} x{r["launches"]:<3d} {r["tag"][-70:]:70s} {(r.get("kernel") or "")[:32]}')
P
