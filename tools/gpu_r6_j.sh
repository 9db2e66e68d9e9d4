#!/bin/bash
# round 6, GPU call J: two-caller test, chunked window-attention backward, per-layer SyncBatchNorm setting, training steps after those changes
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_train_tinyvit.py tests/test_stage1_step.py tests/test_dist_gloo.py -q -m gpu --timeout 600 \
  -k "two_callers or window_attention or training_step or rccl or tinyvit_block" > $O/j_tests.txt 2>&1
tail -6 $O/j_tests.txt | cut -c1-300
