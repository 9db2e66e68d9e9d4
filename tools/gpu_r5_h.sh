#!/bin/bash
# round 5, GPU call H: the hand-pipelined expand phase of mbconv3s -- exact-lattice tests, then A/B timings (previous / new) on one box
O=gpurun_out/r05; mkdir -p $O
timeout 200 python -m pytest tests/test_lattice_gpu.py tests/test_ops_gpu.py -q -m gpu -k "mbconv3" --timeout 150 2>&1 | tail -2
for lib in head dev head dev; do
  ESAM3_DEV_LIB=build_dev/libesam3_$lib.so ESAM3_OP_REPEAT=30 timeout 100 python tools/evit_fused_bench.py s0.0 s0.1 s1.0 s1.1 2>&1 | grep op_timed | sed "s/^/$lib /" | cut -c1-110 | tee -a $O/h_evit_ab.txt
done
