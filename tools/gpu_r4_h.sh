#!/bin/bash
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1500 python -m pytest tests/test_stage1_step.py -q -s -k "two_training or bf16_training" 2>&1 | grep -v "^$" | tail -30 | tee $O/stage1_step_h.txt
timeout 900 python -m pytest tests/test_pcs.py -x -q -k "video_grounding" 2>&1 | tail -25 | tee -a $O/stage1_step_h.txt
