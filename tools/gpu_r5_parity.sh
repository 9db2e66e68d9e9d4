#!/bin/bash
# round 5, GPU call A: the new parity gates (exact-lattice tests, distribution tests, PCS yardstick tests, the many-point decode after
# release_host_weights) and the stage-1 step test with its printed margins
mkdir -p gpurun_out/r05
python -m pytest tests/test_lattice_gpu.py tests/test_bf16_distribution.py tests/test_pcs.py tests/test_stage1_step.py \
  "tests/test_e2e_gpu.py::test_release_host_weights_keeps_results" "tests/test_e2e_gpu.py::test_predict_inst_vs_golden" \
  -q -m gpu -rP --timeout 1500 > gpurun_out/r05/parity_a.txt 2>&1
tail -5 gpurun_out/r05/parity_a.txt
grep -c "ok  \|FAIL" gpurun_out/r05/parity_a.txt
