#!/bin/bash
# round 6, GPU call K: does a power-bound GEMM launch need all 256 CUs?  (dev library, persistent grid capped)
O=gpurun_out/r06
mkdir -p $O
for g in 256 248 240 224 208 192 160 128 256; do
  echo "== ESAM3_P_GRID=$g"
  ESAM3_P_GRID=$g ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_BENCH_ITERS=40 timeout 120 python tools/bench_gemm.py "up-conv,neck L1 3x3,head.3" 2>&1 | grep "ms"
done > $O/k_gemm_grid_cap.txt 2>&1
cat $O/k_gemm_grid_cap.txt
