#!/bin/bash
# config 4 (ViT-H + text + detector, batch 8): bench line + aggregated per-launch table, for each environment setting given
mkdir -p gpurun_out/r02
for e in "$@"; do
  echo "== $e"
  env $e ESAM3_BENCH_PROFILE_OUT=gpurun_out/r02/cfg4_pl.json python bench.py --workload text --backbone sam3 --model vit_h --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02/cfg4_line.json
  python - <<'P'
import json, collections, re
b = json.load(open("gpurun_out/r02/cfg4_line.json"))
print(b["value"], b["ms_per_step"], b["config"]["kernel_ms_per_step_by_stage"])
d = json.load(open("gpurun_out/r02/cfg4_pl.json"))
agg = collections.Counter(); cnt = collections.Counter()
for r in d["per_tag"]:
    k = re.sub(r"\.\d+\.", ".N.", r["tag"]); agg[k] += r["ms"]; cnt[k] += r["launches"]
for k, v in agg.most_common(14):
    print(f"   {v:7.3f} {cnt[k]:4d} {k[:90]}")
P
done
