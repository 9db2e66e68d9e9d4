#!/bin/bash
# A/B of the ablation builds (tools/dev_variants.sh) on the key shapes, two interleaved rounds
mkdir -p gpurun_out/r02
for round in 1 2; do
  for v in "$@"; do
    ESAM3_DEV_LIB=build_dev/libesam3_$v.so python tools/bench_gemm.py "neck L0 3x3,neck L1 3x3,convT0,ViT-H qkv,ViT-H fc2" 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r02/variants_$(date +%H%M%S).log
