#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 timeout 300 python tools/evit_fused_bench.py tv.0 tv.0:v2 tv.hs s1.1 2>&1 | grep -v amdgpu | tee $O/g_tv_mbconv_ab.txt
