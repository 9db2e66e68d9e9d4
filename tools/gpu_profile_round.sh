#!/bin/bash
# Round profile of the headline step: rocprofv3 kernel stats + the PMC passes of the dominant launch (separate runs, as
# MI355X_MICROARCH.md prescribes), condensed by tools/rocprof_summary.py into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r06}
O=$R/gpurun_out/$ROUND/prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d $O/sq1 -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $O/sq2 -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2>&1
cd $R && python tools/rocprof_summary.py --stats $O/stats --fetch $O/fetch --write $O/write --sq $O/sq1 --sq $O/sq2 --steps 6 --round $ROUND 2>&1 | tail -40
cp profiles/${ROUND}_kernel_stats.csv profiles/pmc_dominant_kernel.json profiles/pmc_dominant_kernel_sq.json gpurun_out/$ROUND/ 2>/dev/null
# keep the merge small: drop the raw traces
find $O -name "*_kernel_trace.csv" -size +8M -delete
