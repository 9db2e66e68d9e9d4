#!/bin/bash
# round-5 artifacts after the last kernel changes: default bench line (+ per-launch + roofline tables), rocprofv3 stats + PMC passes of the
# dominant launch, a one-rank torch.distributed.run launch of the bench, the stage-1 step lines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
ESAM3_BENCH_PROFILE_OUT=$O/bench_headline_per_launch.json timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_headline.json
python -c "
import json; b=json.load(open('$O/bench_headline.json')); print(b['value'], b['ms_per_step'], json.dumps(b['roofline'])[:300]); print(b['cpu_baseline']['value']); print(b['step_roofline_frac'], b['config']['kernel_ms_per_step_by_stage'], b['config']['launches_per_step'], b['config']['api_level_images_per_s'], b['config']['pcie_inclusive_images_per_s'])"
ROUND=r05 bash tools/gpu_profile_round.sh 2>&1 | tail -6
cd $R
python tools/roofline_table.py $O/bench_headline_per_launch.json --merge-layers > $O/roofline_headline.md 2>/dev/null; head -12 $O/roofline_headline.md
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("torchrun x1:", d["value"], d["n_gpus"], d["config"]["collective_backend"], d["config"]["ranks_in_process_group"], d["config"]["collective_error"])' | tee $O/torchrun_x1.txt
timeout 200 python tools/bench_stage1_step.py > $O/bench_stage1_step.json 2>/dev/null; tail -c 200 $O/bench_stage1_step.json
timeout 200 python tools/bench_stage1_step.py --batch 32 > $O/bench_stage1_step_b32.json 2>/dev/null; tail -c 200 $O/bench_stage1_step_b32.json
