#!/bin/bash
# round-4 artifacts, part 1: full GPU suite + smoke, the default bench line with its per-launch table, rocprofv3 stats + PMC passes of the
# dominant launch (tools/gpu_profile_round.sh), the per-launch roofline table
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
python -m pytest tests/ -q -m gpu 2>&1 | tail -25 | tee $O/full_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/full_gpu_suite.log
ESAM3_BENCH_PROFILE_OUT=$O/bench_headline_per_launch.json timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_headline.json; cut -c1-300 $O/bench_headline.json; python -c "
import json; b=json.load(open('$O/bench_headline.json')); print(json.dumps(b['roofline'])); print(b['cpu_baseline']); print(b['step_roofline_frac'], b['config']['kernel_ms_per_step_by_stage'], b['config']['launches_per_step'], b['config']['api_level_images_per_s'], b['config']['pcie_inclusive_images_per_s'])"
ROUND=r04 bash tools/gpu_profile_round.sh 2>&1 | tail -30
cd $R
python tools/roofline_table.py $O/bench_headline_per_launch.json --merge-layers > $O/roofline_headline.md 2>/dev/null; head -30 $O/roofline_headline.md
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("torchrun x1:", d["value"], d["n_gpus"], d["config"]["collective_backend"], d["config"]["ranks_in_process_group"], d["config"]["collective_error"])' | tee -a $O/full_gpu_suite.log
