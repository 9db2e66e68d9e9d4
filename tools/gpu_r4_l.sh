#!/bin/bash
# round 4, GPU call L: token-side fused kernels: A/B of the decoder variants on the same inputs, the decode tests, bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
export ESAM3_DEV_LIB=$R/build_dev/libesam3_dev.so
python tools/decoder_fused_check.py run /tmp/all.npz 2>&1 | tail -2
ESAM3_NO_TOK_FUSED=1 python tools/decoder_fused_check.py run /tmp/notok.npz 2>&1 | tail -1
ESAM3_NO_TOK_FUSED=1 ESAM3_NO_I2T_FUSED=1 ESAM3_NO_T2I_MFMA=1 python tools/decoder_fused_check.py run /tmp/r3.npz 2>&1 | tail -1
( python tools/decoder_fused_check.py cmp /tmp/all.npz /tmp/notok.npz; python tools/decoder_fused_check.py cmp /tmp/notok.npz /tmp/r3.npz; python tools/decoder_fused_check.py cmp /tmp/all.npz /tmp/r3.npz ) | tee $O/decoder_fused_check_l.txt
unset ESAM3_DEV_LIB
timeout 2400 python -m pytest tests/test_e2e_gpu.py tests/test_students_gpu.py -q -x 2>&1 | tail -6 | tee $O/tests_l.txt
ESAM3_BENCH_PROFILE_OUT=$O/bench_l_per_launch.json timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_l.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04/bench_l.json')); print('bench', b['value'], b['ms_per_step'], b['config'].get('kernel_ms_per_step_by_stage'), b.get('step_roofline_frac'), b['config'].get('launches_per_step'))
PY
