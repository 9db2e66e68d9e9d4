#!/bin/bash
# round 4, GPU call N: after the host-side API fixes (numpy widening instead of torch's OpenMP copy, pinned route for the low-res logits),
# the pivoted BatchNorm statistics and the in-process store of forced one-rank groups: the tests that touch them + the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_stage1.py tests/test_dist_gloo.py tests/test_facade_eval_coco.py tests/test_stage1_step.py -q -m gpu 2>&1 | tail -5 | tee $O/tests_n.txt
timeout 300 python -m pytest tests/test_pcs.py -q -m gpu -k video 2>&1 | tail -2 | tee -a $O/tests_n.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_n_per_launch.json timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_n.json')); c=b['config']; print('bench', b['value'], b['ms_per_step'], c['kernel_ms_per_step_by_stage'], b['step_roofline_frac'], 'api', c['api_level_images_per_s'], 'pcie', c['pcie_inclusive_images_per_s'], b['cpu_baseline']['value'])
PY
