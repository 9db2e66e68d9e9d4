#!/bin/bash
# round 5, GPU call C: the PCS precision changes and the trimmed distribution tests, the host-path changes (chunked D2H + native
# widening, uneven chunks) through the batch tests and the bench's API leg, the stage-1 depthwise data gradient, the host ceiling
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 700 python -m pytest tests/test_pcs.py tests/test_bf16_distribution.py tests/test_stage1_step.py tests/test_train_blocks.py tests/test_e2e_gpu.py tests/test_facade_eval_coco.py \
  -q -m gpu -rP --durations=8 --timeout 600 \
  -k "pcs_bf16_distribution or config4 or geometric or pcs_engine or efficientvit-b0 or two_training_steps or dwconv or mbconv_block or batch or release or eval_coco or predict_inst_vs_golden" > $O/c_tests.txt 2>&1
tail -14 $O/c_tests.txt | cut -c1-200
grep -h "FAIL" $O/c_tests.txt | cut -c1-260
timeout 200 python bench.py --no-cpu-baseline > $O/c_bench.json 2> $O/c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/c_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], "api", d["config"]["api_level_images_per_s"], "pcie", d["config"]["pcie_inclusive_images_per_s"])
PY
timeout 150 python tools/bench_stage1_step.py > $O/c_stage1_step.json 2> $O/c_stage1_step.err; tail -c 400 $O/c_stage1_step.json
timeout 150 python tools/bench_stage1_step.py --batch 32 > $O/c_stage1_step_b32.json 2>> $O/c_stage1_step.err; tail -c 300 $O/c_stage1_step_b32.json
timeout 200 python tools/host_ceiling.py --ranks 8 --steps 30 > $O/c_host_ceiling.txt 2>&1; cat $O/c_host_ceiling.txt
