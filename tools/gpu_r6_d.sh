#!/bin/bash
# round 6, GPU call D: the new parity pins (three students, scopes), then the whole GPU suite with its slowest tests listed
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_stage1_step.py tests/test_e2e_gpu.py tests/test_pcs.py -q -m gpu -rP --timeout 600 \
  -k "b0_training or repvit_m2_3 or tiny_vit_21m or profiler_scopes or (pcs_engine_vs_golden and bf16)" > $O/d_new_pins.txt 2>&1
grep -E "^\[stage-1|gradients:|parameters after|BatchNorm running|passed|failed|^E " $O/d_new_pins.txt | cut -c1-330
timeout 1500 python -m pytest tests -q -m gpu -x --durations=60 --timeout 900 > $O/d_full_suite.txt 2>&1
tail -75 $O/d_full_suite.txt | cut -c1-200
