#!/bin/bash
# round 4, GPU call A: first run of the fused EfficientViT kernels (op checks with error localisation, stage parity, bench A/B)
mkdir -p gpurun_out/r04
timeout 600 python tools/evit_fused_check.py 2>&1 | tee gpurun_out/r04/evit_fused_check.txt | tail -60
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "stages or complete or smoke or case" 2>&1 | tail -15 | tee gpurun_out/r04/e2e_subset.txt
ESAM3_BENCH_PROFILE_OUT=gpurun_out/r04/bench_a_per_launch.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_a.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_a.json')); print('bench', b['value'], b['ms_per_step'], b['config'].get('kernel_ms_per_step_by_stage'))
d=json.load(open('gpurun_out/r04/bench_a_per_launch.json'))
for r in sorted(d['per_tag'], key=lambda r:-r['ms'])[:40]:
    print(f"{r['ms']:.3f} x{r['launches']} {r['tag'][-60:]}")
PY
if [ -f build_dev/libesam3_dev.so ]; then
  cp efficientsam3_amd/libesam3_hip.so /tmp/prod.so; cp build_dev/libesam3_dev.so efficientsam3_amd/libesam3_hip.so
  for e in "X=1" "ESAM3_MB_V2=1" "ESAM3_NO_MLA_FUSED=1" "ESAM3_MB_V2=1 ESAM3_NO_MLA_FUSED=1"; do
    echo "== $e: $(env $e python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("kernel_ms_per_step_by_stage"))')"
  done | tee gpurun_out/r04/bench_a_ab.txt
  cp /tmp/prod.so efficientsam3_amd/libesam3_hip.so
fi
