#!/bin/bash
# BASELINE configs 3 (TV-M shard of 32) and 4 (ViT-H + text, batch 8): bench line + per-launch table each
mkdir -p gpurun_out/r02
ESAM3_BENCH_PROFILE_OUT=gpurun_out/r02/bench_tinyvit_11m_per_launch.json python bench.py --backbone tinyvit --model 11m --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02/bench_tinyvit_11m.json
ESAM3_BENCH_PROFILE_OUT=gpurun_out/r02/bench_text_cfg4_per_launch.json python bench.py --workload text --backbone sam3 --model vit_h --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02/bench_text_cfg4.json
ESAM3_BENCH_PROFILE_OUT=gpurun_out/r02/bench_repvit_m1.1_per_launch.json python bench.py --backbone repvit --model m1.1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02/bench_repvit_m1.1.json
for f in tinyvit_11m text_cfg4 repvit_m1.1; do cut -c1-160 gpurun_out/r02/bench_$f.json; done
