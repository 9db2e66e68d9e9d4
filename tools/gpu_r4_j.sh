#!/bin/bash
# round 4, GPU call J: full GPU suite + smoke + bench after the training step, the video path and the resize_shuffle rewrite
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/tests_j.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/tests_j.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_j_per_launch.json timeout 900 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_j.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_j.json')); print('bench', b['value'], b['ms_per_step'], b['config'].get('kernel_ms_per_step_by_stage'), b.get('step_roofline_frac'), b['roofline']['frac'], b.get('cpu_baseline'))
PY
