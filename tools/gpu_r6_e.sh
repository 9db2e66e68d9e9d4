#!/bin/bash
# round 6, GPU call E: idle time between the kernels of a headline step
O=gpurun_out/r06
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/e_bench_traced.json 2>/dev/null
cd $R
python tools/kernel_gaps.py /tmp/kt --out $O/e_kernel_gaps.txt | cut -c1-200
