#!/bin/bash
# parity tests with their printed per-case tables
mkdir -p gpurun_out/r02
python -m pytest tests/test_e2e_gpu.py tests/test_students_gpu.py tests/test_pcs.py -q -m gpu -s 2>&1 | grep -v "^$" | grep "^\[\|passed\|failed\|FAILED\|Error\|assert" | cut -c1-420 | tee gpurun_out/r02/parity_$(date +%H%M%S).log
