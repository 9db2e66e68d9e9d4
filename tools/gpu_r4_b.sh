#!/bin/bash
# round 4, GPU call B: where does the time of the fused EfficientViT kernels go?  per-shape timings, phase ablations, SQ counters
mkdir -p gpurun_out/r04
export ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
( echo "== all shapes, v3"; python tools/evit_fused_bench.py 2>&1 | grep op_timed -B0
  echo "== v2 for the shapes it has"; python tools/evit_fused_bench.py s0.0:v2 s0.1:v2 s1.0:v2 s1.1:v2 s2.0:v2 2>&1 | grep -a "op_timed"
  for abl in 1 2 4 8 16 6 14 30; do
    echo "== ablation mask $abl"; ESAM3_MB3_ABL=$abl python tools/evit_fused_bench.py s0.1 s1.1 s2.loc s3.loc 2>&1 | grep -a "op_timed"
  done ) | grep -v amdgpu.ids | tee $O/evit_fused_bench_b.txt
cd /tmp && export TMPDIR=/tmp
export ESAM3_OP_REPEAT=3
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_b1 -o a --output-format csv -- python $R/tools/evit_fused_bench.py s0.1 s1.1 s2.loc s3.loc s2.ctx s3.ctx > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $O/pmc_b2 -o b --output-format csv -- python $R/tools/evit_fused_bench.py s0.1 s1.1 s2.loc s3.loc s2.ctx s3.ctx > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA --kernel-trace -d $O/pmc_b3 -o c --output-format csv -- python $R/tools/evit_fused_bench.py s0.1 s1.1 s2.loc s3.loc s2.ctx s3.ctx > /dev/null 2>&1
cd $R && python - <<'P' | tee $O/pmc_b_summary.txt
import csv, glob, collections
O = "gpurun_out/r04"
for tag in ("pmc_b1", "pmc_b2", "pmc_b3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(O + f"/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mbconv3" in k or "mla" in k:
                agg[(k[:80], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(O + f"/{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mbconv3" in k or "mla" in k:
                dur[(k[:80], r["Grid_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, cs in sorted(agg.items()):
        d = dur.get(k, [0])
        print(tag, k, "avg_us", round(sum(d) / max(len(d), 1) / 1e3, 1), {c: round(sum(v) / len(v)) for c, v in cs.items()})
P
find $O -name "*_kernel_trace.csv" -size +4M -delete; find $O -name "*.db" -delete
