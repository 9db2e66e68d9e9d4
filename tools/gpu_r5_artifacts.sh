#!/bin/bash
# round-5 artifacts in one call: the default bench line (+ per-launch table + roofline table), rocprofv3 stats + PMC passes of the
# dominant launch, a one-rank torch.distributed.run launch of the bench, the other configurations, the stage-1 step with its kernel
# table, the API A/B and the API-level probe.  (The round's intermediate A/B calls were one-off scripts and are not kept; what they
# measured is under profiles/r05/ with the tool named in profiles/README.md.)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
ESAM3_BENCH_PROFILE_OUT=$O/bench_headline_per_launch.json timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_headline.json
python -c "
import json; b=json.load(open('$O/bench_headline.json')); print(b['value'], b['ms_per_step'], json.dumps(b['roofline'])[:400]); print(b['cpu_baseline']); print(b['step_roofline_frac'], b['config']['kernel_ms_per_step_by_stage'], b['config']['launches_per_step'], b['config']['api_level_images_per_s'], b['config']['pcie_inclusive_images_per_s'])"
ROUND=r05 bash tools/gpu_profile_round.sh 2>&1 | tail -12
cd $R
python tools/roofline_table.py $O/bench_headline_per_launch.json --merge-layers > $O/roofline_headline.md 2>/dev/null; head -24 $O/roofline_headline.md
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("torchrun x1:", d["value"], d["n_gpus"], d["config"]["collective_backend"], d["config"]["ranks_in_process_group"], d["config"]["collective_error"])' | tee $O/torchrun_x1.txt
for cfg in "tinyvit 11m 32 interactive" "repvit m1.1 32 interactive" "sam3 vit_h 8 text"; do
  set -- $cfg
  ESAM3_BENCH_PROFILE_OUT=$O/bench_$1_$2_per_launch.json timeout 400 python bench.py --backbone $1 --model $2 --batch $3 --workload $4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$1_$2.json
  python -c "
import json; b=json.load(open('$O/bench_$1_$2.json')); print('$1 $2', b['value'], b['ms_per_step'], b['config']['kernel_ms_per_step_by_stage'])"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dry-collective 2>/dev/null | tail -1 > $O/bench_dry_collective.json; python -c "
import json; b=json.load(open('$O/bench_dry_collective.json')); c=b['config']; print('dry-collective', b['value'], c['collective_backend'], c['ranks_in_process_group'], c['side_stream_gathers'], c['collective_error'])"
timeout 200 python tools/bench_stage1_step.py > $O/bench_stage1_step.json 2>/dev/null; tail -c 300 $O/bench_stage1_step.json
timeout 200 python tools/bench_stage1_step.py --batch 32 > $O/bench_stage1_step_b32.json 2>/dev/null; tail -c 300 $O/bench_stage1_step_b32.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof/stage1 -o s --output-format csv -- python $R/tools/bench_stage1_step.py --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $O/prof/stage1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/stage1_step_kernel_stats.csv 2>/dev/null
find $O/prof -name "*_kernel_trace.csv" -size +8M -delete
cd $R
head -12 $O/stage1_step_kernel_stats.csv | cut -c1-160
timeout 200 python tools/stage1_forward_bench.py --batch 8 --steps 5 2>/dev/null | tail -1 > $O/bench_stage1_paired.json; cut -c1-200 $O/bench_stage1_paired.json
timeout 200 python tools/api_ab.py > $O/api_ab.txt 2>&1; grep "round 1" $O/api_ab.txt
timeout 200 python tools/api_level_probe.py > $O/api_level_probe.txt 2>&1; grep "^rep\|^step\|api step" $O/api_level_probe.txt | cut -c1-260
