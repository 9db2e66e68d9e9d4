#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
for g in 0 224 192 160 128; do
  ESAM3_LIB=build_dev/libesam3_dev.so ESAM3_P_GRID=$g timeout 300 python tools/two_stream_probe.py 2>&1 | grep "streams="
done > $O/l_two_stream_probe.txt 2>&1
cat $O/l_two_stream_probe.txt | cut -c1-120
