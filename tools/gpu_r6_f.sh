#!/bin/bash
# round 6, GPU call F: TinyViT MBConv on the mbconv3s design with GELU epilogues: op test, TinyViT parity tests, config-3 shard bench
O=gpurun_out/r06
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_students_gpu.py tests/test_bf16_distribution.py -q -m gpu -rP --timeout 500 \
  -k "mbconv3_gelu or tinyvit" > $O/f_tinyvit_tests.txt 2>&1
grep -E "passed|failed|^E |\[dist" $O/f_tinyvit_tests.txt | cut -c1-300
ESAM3_BENCH_PROFILE_OUT=$O/f_bench_tinyvit_per_launch.json timeout 400 python bench.py --backbone tinyvit --model 11m --no-cpu-baseline > $O/f_bench_tinyvit.json 2> $O/f_bench_tinyvit.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/f_bench_tinyvit.json").read().strip().splitlines()[-1])
print("tinyvit-11m", d["value"], d["ms_per_step"])
p = json.load(open("gpurun_out/r06/f_bench_tinyvit_per_launch.json"))["per_tag"]
for r in sorted(p, key=lambda r: -r["ms"])[:12]:
    print(f"  {r['ms']:.3f} x{r['launches']} {r['tag'][:90]}")
PY
