#!/bin/bash
# round 4, GPU call F: LiteMLA with the tile-blocked q layout + prefetched grouped weights: checks + timings
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
timeout 600 python tools/evit_fused_check.py 2>&1 | grep -v "^   per" | grep "lite_mla\|branch" | tee $O/evit_fused_check_f.txt
( export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20; python tools/evit_fused_bench.py s2.ctx s3.ctx 2>&1 | grep -a op_timed ) | tee $O/evit_fused_bench_f.txt
cd /tmp && export TMPDIR=/tmp ESAM3_OP_REPEAT=5 ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so
rocprofv3 --kernel-trace --stats -d $O/trace_f -o t --output-format csv -- python $R/tools/evit_fused_bench.py s2.ctx s3.ctx > $O/trace_f.log 2>&1
cd $R && python - <<'P' | tee -a $O/evit_fused_bench_f.txt
import csv, glob
for f in glob.glob("gpurun_out/r04/trace_f/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mla" in r["Name"]:
            print("stats", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
P
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "lite_mla" 2>&1 | tail -3
find $O -name "*_kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
