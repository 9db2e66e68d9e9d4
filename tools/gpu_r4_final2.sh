#!/bin/bash
# round-4 artifacts, part 2: the other configurations (config 3 shard, RV-M, config 4), the one-rank RCCL dry run, the stage-1 lines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
for cfg in "tinyvit 11m 32 interactive" "repvit m1.1 32 interactive" "sam3 vit_h 8 text"; do
  set -- $cfg
  ESAM3_BENCH_PROFILE_OUT=$O/bench_$1_$2_per_launch.json timeout 900 python bench.py --backbone $1 --model $2 --batch $3 --workload $4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$1_$2.json
  python -c "
import json; b=json.load(open('$O/bench_$1_$2.json')); print('$1 $2', b['value'], b['ms_per_step'], b['config']['kernel_ms_per_step_by_stage'])"
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dry-collective 2>/dev/null | tail -1 > $O/bench_dry_collective.json; python -c "
import json; b=json.load(open('$O/bench_dry_collective.json')); c=b['config']; print('dry-collective', b['value'], c['collective_backend'], c['ranks_in_process_group'], c['side_stream_gathers'], c['collective_error'])"
timeout 600 python tools/stage1_forward_bench.py --batch 8 --steps 5 2>/dev/null | tail -1 > $O/bench_stage1_paired.json; cut -c1-300 $O/bench_stage1_paired.json
