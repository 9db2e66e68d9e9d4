#!/bin/bash
# round 4, GPU call C: persistent / prefetching MBConv variants (mbconv3s, mbconv3b), LiteMLA with DMA-staged weights:
# op checks, per-shape timings, kernel trace + SQ counters of the fused operators
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
timeout 600 python tools/evit_fused_check.py 2>&1 | grep -v "^   per" | tee $O/evit_fused_check_c.txt | tail -40
export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20
( echo "== all shapes"; python tools/evit_fused_bench.py 2>&1 | grep -a op_timed
  echo "== generic mbconv3 (ESAM3_MB3_GENERIC=1)"; ESAM3_MB3_GENERIC=1 python tools/evit_fused_bench.py s0.1 s1.1 s2.loc s3.loc 2>&1 | grep -a op_timed
  for gd in 256 768 1024; do echo "== mbconv3s grid $gd"; ESAM3_MB3S_GRID=$gd python tools/evit_fused_bench.py s0.0 s0.1 s1.0 s1.1 2>&1 | grep -a op_timed; done
  for abl in 2 4 8 16 14; do
    echo "== ablation mask $abl (3b shapes only honour nothing: generic kernels)"; ESAM3_MB3_GENERIC=1 ESAM3_MB3_ABL=$abl python tools/evit_fused_bench.py s2.0 s3.0 2>&1 | grep -a "op_timed"
  done ) | grep -v amdgpu.ids | tee $O/evit_fused_bench_c.txt
cd /tmp && export TMPDIR=/tmp
export ESAM3_OP_REPEAT=3
SH="s0.0 s0.1 s1.1 s2.loc s3.loc s2.ctx s3.ctx"
rocprofv3 --kernel-trace --stats -d $O/trace_c -o t --output-format csv -- python $R/tools/evit_fused_bench.py $SH > $O/trace_c.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_c1 -o a --output-format csv -- python $R/tools/evit_fused_bench.py $SH > $O/pmc_c1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $O/pmc_c2 -o b --output-format csv -- python $R/tools/evit_fused_bench.py $SH > $O/pmc_c2.log 2>&1
tail -3 $O/pmc_c2.log
cd $R && python - <<'P' | tee $O/pmc_c_summary.txt
import csv, glob, collections
O = "gpurun_out/r04"
for f in glob.glob(O + "/trace_c/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mbconv3" in r["Name"] or "mla" in r["Name"]:
            print("stats", r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for tag in ("pmc_c1", "pmc_c2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + f"/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mbconv3" in k or "mla" in k:
                agg[(k[:80], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        print(tag, k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
P
find $O -name "*_kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
