#!/bin/bash
# round 5: the whole GPU suite as the driver runs it (+ the printed margins of every parity test), smoke(), the default bench line
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 1300 python -m pytest tests -q -m gpu -rP --durations=15 --timeout 900 > $O/full_gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/full_gpu_suite.txt | tail -3
grep -h "FAIL\|^E  " $O/full_gpu_suite.txt | cut -c1-260 | head -30
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt | cut -c1-400
