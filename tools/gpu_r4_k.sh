#!/bin/bash
# round 4, GPU call K3: fused i2t block + merged k/v projection + MFMA token -> image attention: op tests, op timings, the
# decode / pcs / student tests, bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
true
true
timeout 2400 python -m pytest tests/test_pcs.py tests/test_e2e_gpu.py tests/test_students_gpu.py -q 2>&1 | tail -8 | tee $O/tests_k.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_k_per_launch.json timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_k.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_k.json')); print('bench', b['value'], b['ms_per_step'], b['config'].get('kernel_ms_per_step_by_stage'), b.get('step_roofline_frac'), b['config'].get('launches_per_step'))
PY
