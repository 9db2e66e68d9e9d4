#!/bin/bash
# round 4, GPU call D: fused EfficientViT kernels after the instruction-count / prefetch pass: op checks, op timings, the op + e2e
# GPU tests that exercise the backbone, the bench line with per-launch table
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
timeout 600 python tools/evit_fused_check.py 2>&1 | grep -v "^   per" | tee $O/evit_fused_check_d.txt | tail -22
( export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20; python tools/evit_fused_bench.py 2>&1 | grep -a op_timed ) | tee $O/evit_fused_bench_d.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -q -x 2>&1 | tail -8 | tee $O/tests_d.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_d_per_launch.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_d.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_d.json')); print('bench', b['value'], b['ms_per_step'], b['config'].get('kernel_ms_per_step_by_stage'))
d=json.load(open('gpurun_out/r04/bench_d_per_launch.json'))
for r in sorted(d['per_tag'], key=lambda r:-r['ms'])[:32]:
    print(f"{r['ms']:.3f} x{r['launches']} {r['tag'][-60:]}")
PY
