#!/bin/bash
# one gpurun call: GEMM parity tests, then the timing of the 256x256 kernels (product, classic, no-epilogue ablation)
mkdir -p gpurun_out/r02
python -m pytest tests/test_ops_gpu.py -x -q -k "linear or conv or gemm256p" 2>&1 | tail -5 | tee gpurun_out/r02/gemm_tests.log
for round in 1 2; do
python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
ESAM3_DEV_LIB=build_dev/libesam3_nostore.so python tools/bench_gemm.py "neck L0 3x3,convT0,ViT-H qkv,ViT-H fc2" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02/bench_gemm_p2.log
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r02/bench_second.log
