#!/bin/bash
# round 6, GPU call I: per-operator roofline of a B1 batch-32 training step (+ the throughput lines of three students)
O=gpurun_out/r06
mkdir -p $O
timeout 400 python tools/stage1_step_roofline.py --model b1 --batch 32 2>&1 | grep -v amdgpu > $O/i_roofline_stage1_step_b1_b32.md
head -40 $O/i_roofline_stage1_step_b1_b32.md | cut -c1-220
timeout 300 python tools/bench_stage1_step.py --model b1 --batch 32 --steps 5 2>/dev/null | tail -1 > $O/i_bench_stage1_step_b1_b32.json
python -c "
import json; d=json.loads(open('$O/i_bench_stage1_step_b1_b32.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
