#!/bin/bash
# round 4, GPU call R: hole filling with path halving + shared walks: the e2e cases that run it, smoke, the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -k "predict_inst_vs_golden or full_batch or batch_api" 2>&1 | tail -2 | tee $O/tests_r.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/tests_r.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_r_per_launch.json timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_r.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_r.json')); c=b['config']; print('bench', b['value'], b['ms_per_step'], c['kernel_ms_per_step_by_stage'], b['step_roofline_frac'], b['roofline']['frac'], 'api', c['api_level_images_per_s'], 'pcie', c['pcie_inclusive_images_per_s'], b['cpu_baseline']['value'])
PY
