#!/bin/bash
# Round 5, the stage-1 TRAINING path on one MI355X: every training test with its printed margins, the step timing of every student (or of
# the ones named: MODELS="b1 tiny_vit_11m"), batch-32 timings of one student per family, and a rocprofv3 --stats kernel table of a step of
# PROFILE_MODEL (default b1).  TAG names the output files (gpurun_out/r05/*_<TAG>.*).  This one script replaces the per-change call
# scripts of the round (RepViT, TinyViT, all students, LiteMLA token passes, BatchNorm halves, one-launch 3x3 wgrad, BatchNorm + activation):
# what each of those runs produced is under profiles/r05/ with the tag it was given, see profiles/README.md.
#   TAG=final MODELS="b1 b2" PROFILE_MODEL=b2 bash tools/gpu_r5_training.sh
O=gpurun_out/r05; mkdir -p $O
TAG=${TAG:-run}
MODELS=${MODELS:-"b0 b1 b2 repvit_m0_9 repvit_m1_1 repvit_m2_3 tiny_vit_5m tiny_vit_11m tiny_vit_21m"}
PROFILE_MODEL=${PROFILE_MODEL:-b1}
timeout 900 python -m pytest tests/test_stage1.py tests/test_train_blocks.py tests/test_train_repvit.py tests/test_train_tinyvit.py tests/test_stage1_step.py \
  -q -m gpu -rP --timeout 600 > $O/training_tests_$TAG.txt 2>&1
tail -2 $O/training_tests_$TAG.txt | cut -c1-200; grep -h "^E  \|FAILED" $O/training_tests_$TAG.txt | cut -c1-300 | head -20
timeout 300 python -m pytest tests/test_dist_gloo.py -q -m gpu -rP --timeout 250 -k "sync_batchnorm or one_rank or allreducer" > $O/training_dist_tests_$TAG.txt 2>&1
tail -1 $O/training_dist_tests_$TAG.txt
report() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], "images/s", d["ms_per_step"], "ms", d["roofline"]["achieved"], "TFLOP/s")
PY
}
for m in $MODELS; do
  timeout 200 python tools/bench_stage1_step.py --model $m > $O/bench_stage1_step_${m}_$TAG.json 2>/dev/null; report $O/bench_stage1_step_${m}_$TAG.json $m
done
for m in b1 repvit_m1_1 tiny_vit_11m; do
  timeout 200 python tools/bench_stage1_step.py --model $m --batch 32 > $O/bench_stage1_step_${m}_b32_$TAG.json 2>/dev/null; report $O/bench_stage1_step_${m}_b32_$TAG.json "$m b32"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof/stage1_$TAG -o s --output-format csv -- python $R/tools/bench_stage1_step.py --model $PROFILE_MODEL --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/$O/prof/stage1_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stage1_step_${PROFILE_MODEL}_kernel_stats_$TAG.csv 2>/dev/null
find $R/$O/prof -name "*_kernel_trace.csv" -delete
head -12 $R/$O/stage1_step_${PROFILE_MODEL}_kernel_stats_$TAG.csv | cut -c1-170
