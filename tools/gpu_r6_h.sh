#!/bin/bash
# round 6, GPU call H: host worker threads of the API path (PIL staging + mask widening): 8 / 16 / 32 / 64
O=gpurun_out/r06
mkdir -p $O
for n in 16 8 32 64 16; do
  echo "== ESAM3_HOST_THREADS=$n"
  ESAM3_HOST_THREADS=$n timeout 200 python tools/api_level_probe.py 2>&1 | grep -E "api step, results|keep previous|rep 1|stage 32 PIL images, rgbx np" | cut -c1-220
done > $O/h_api_host_threads.txt 2>&1
cat $O/h_api_host_threads.txt
