#!/bin/bash
# one gpurun call: GEMM parity tests, then the A/B timing of the two 256x256 kernels
set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_ops_gpu.py -x -q -k "linear or conv or gemm256p" 2>&1 | tail -15 | tee gpurun_out/r02/gemm_tests.log
python tools/bench_gemm.py 2>&1 | tee gpurun_out/r02/bench_gemm_p.log
ESAM3_GEMM256_CLASSIC=1 python tools/bench_gemm.py 2>&1 | tee gpurun_out/r02/bench_gemm_classic.log
python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/r02/bench_first.log
