#!/bin/bash
# round 5, RepViT students of the stage-1 trainer: the new kernels and layers against autograd, the training steps against the reference's
# own runs, a regression pass over the kernels whose file changed, step timings
O=gpurun_out/r05; mkdir -p $O
timeout 400 python -m pytest tests/test_train_repvit.py -q -m gpu -rP --timeout 300 > $O/rv_blocks.txt 2>&1; tail -3 $O/rv_blocks.txt | cut -c1-300
grep -h "^E  \|FAILED" $O/rv_blocks.txt | cut -c1-300 | head -40
timeout 400 python -m pytest tests/test_stage1_step.py -q -m gpu -rP --timeout 300 -k "repvit" > $O/rv_steps.txt 2>&1; tail -3 $O/rv_steps.txt | cut -c1-300
grep -h "^\[stage-1\|^  gradients\|^  parameters\|^  BatchNorm\|^E  " $O/rv_steps.txt | cut -c1-420 | head -40
timeout 300 python -m pytest tests/test_train_blocks.py -q -m gpu --timeout 250 -k "activation or colsum or linear_wgrad or mbconv_block" 2>&1 | tail -2
for m in repvit_m1_1 repvit_m0_9; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_$m.json 2>$O/bench_stage1_step_$m.err; tail -c 700 $O/bench_stage1_step_$m.json; tail -3 $O/bench_stage1_step_$m.err | cut -c1-300
done
timeout 200 python tools/bench_stage1_step.py --model repvit_m1_1 --batch 32 > $O/bench_stage1_step_repvit_m1_1_b32.json 2>/dev/null; tail -c 400 $O/bench_stage1_step_repvit_m1_1_b32.json
