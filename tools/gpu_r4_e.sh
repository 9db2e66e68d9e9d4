#!/bin/bash
# round 4, GPU call E: phase ablation of mla1 and the kernel split of the fused LiteMLA block
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20
( for abl in 0 1 2 4 8 16 3 31; do echo "== mla1 ablation mask $abl"; ESAM3_MLA1_ABL=$abl python tools/evit_fused_bench.py s2.ctx s3.ctx 2>&1 | grep -a op_timed; done ) | tee $O/mla_ablation_e.txt
cd /tmp && export TMPDIR=/tmp ESAM3_OP_REPEAT=5
rocprofv3 --kernel-trace --stats -d $O/trace_e -o t --output-format csv -- python $R/tools/evit_fused_bench.py s2.ctx s3.ctx > $O/trace_e.log 2>&1
cd $R && python - <<'P' | tee -a $O/mla_ablation_e.txt
import csv, glob
for f in glob.glob("gpurun_out/r04/trace_e/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mla" in r["Name"]:
            print("stats", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
P
find $O -name "*_kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
