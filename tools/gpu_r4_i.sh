#!/bin/bash
# round 4, GPU call I5: resize_shuffle row kernel ablations (2: no stores, 8: stores linearised per wave)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20
for a in 0 2 8; do echo "== abl $a"; ESAM3_RS_ABL=$a python tools/neck_ops_bench.py 2>&1 | grep -a resize | head -3; done | tee $O/resize_shuffle_row_abl.txt
