#!/bin/bash
# round 5, GPU call D: config 4 / EV-M detector distribution tests after the pooled-prompt change, the API pipeline A/B, host ceiling facts
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 200 python tools/api_ab.py > $O/d_api_ab.txt 2>&1; grep "round" $O/d_api_ab.txt
timeout 120 python tools/host_ceiling.py --ranks 8 --steps 20 > $O/d_host_ceiling.txt 2>&1; grep -v amdgpu.ids $O/d_host_ceiling.txt | cut -c1-400
timeout 120 python tools/host_ceiling.py --ranks 4 --steps 20 > $O/d_host_ceiling4.txt 2>&1; grep "rank(s)\|ceiling" $O/d_host_ceiling4.txt | cut -c1-300
timeout 500 python -m pytest tests/test_pcs.py -q -m gpu -rP --timeout 400 -k "config4 or pcs_bf16_distribution" > $O/d_tests.txt 2>&1
tail -4 $O/d_tests.txt | cut -c1-200; grep -h "FAIL" $O/d_tests.txt | cut -c1-260
