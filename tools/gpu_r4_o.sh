#!/bin/bash
# round 4, GPU call O: after the i2t rework (split accumulation chains, next rows prefetched) and the rowlin256 kernel: the decoder op tests,
# the e2e / student / pcs tests, smoke, the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "i2t or rowlin or t2i or attention or resize" 2>&1 | tail -2 | tee $O/tests_o.txt
timeout 1800 python -m pytest tests/test_e2e_gpu.py tests/test_students_gpu.py tests/test_pcs.py -q -m gpu 2>&1 | tail -4 | tee -a $O/tests_o.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/tests_o.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_o_per_launch.json timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_o.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_o.json')); c=b['config']; print('bench', b['value'], b['ms_per_step'], c['kernel_ms_per_step_by_stage'], b['step_roofline_frac'], b['roofline']['frac'], 'api', c['api_level_images_per_s'], 'pcie', c['pcie_inclusive_images_per_s'], b['cpu_baseline']['value'])
PY
