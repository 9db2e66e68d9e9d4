#!/bin/bash
# round 4, GPU call Q: fixed-order reductions of the training kernels parallelised (bn_finalize, wgrad_reduce, dw_wgrad_finalize): their tests + the step bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests/test_stage1.py tests/test_train_blocks.py tests/test_stage1_step.py -q -m gpu 2>&1 | tail -3 | tee $O/tests_q.txt
timeout 600 python tools/bench_stage1_step.py --dtype bf16 2>/dev/null | tail -1 | tee $O/bench_stage1_step_q.json | cut -c1-300
timeout 600 python tools/bench_stage1_step.py --dtype bf16 --batch 32 --steps 4 2>/dev/null | tail -1 | tee $O/bench_stage1_step_q_b32.json | cut -c1-300
