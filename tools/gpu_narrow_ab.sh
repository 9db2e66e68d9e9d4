#!/bin/bash
# A/B of conv3x3_narrow ablation builds (SRC=conv3x3_narrow tools/dev_variants.sh ...): timing, then parity of the listed ones
mkdir -p gpurun_out/r02
for round in 1 2; do
  for v in "$@"; do
    ESAM3_DEV_LIB=build_dev/libesam3_$v.so python tools/bench_gemm.py "sam2 L" 2>&1 | grep -v "amdgpu.ids\|^kernel:"
  done
done | tee gpurun_out/r02/narrow_ab_$(date +%H%M%S).log
for v in $PARITY; do
  cp build_dev/libesam3_$v.so efficientsam3_amd/libesam3_hip.so
  echo "parity $v: $(timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k conv3x3_padded 2>&1 | tail -1)"
done
