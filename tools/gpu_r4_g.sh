#!/bin/bash
# round 4, GPU call G: the stage-1 training step (device-weight operators, head, two whole iterations vs the reference fixture),
# the composed trunk in training mode at real size, the video-path entry around the real detector
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1500 python -m pytest tests/test_stage1_step.py -x -q -s 2>&1 | grep -v "^$" | tail -30 | tee $O/stage1_step_g.txt
timeout 900 python -m pytest tests/test_pcs.py -x -q -k "video_grounding" 2>&1 | tail -5 | tee -a $O/stage1_step_g.txt
timeout 900 python -m pytest tests/test_train_blocks.py tests/test_stage1.py -x -q 2>&1 | tail -4 | tee -a $O/stage1_step_g.txt
for a in "f32 128" "bf16 128" "f32 1008"; do timeout 900 python tools/trunk_train_gpu_check.py $a 2>&1 | grep -v amdgpu | tail -12; done | tee $O/trunk_train_check_g.txt
