#!/bin/bash
# round 5, GPU call B: (1) the new kernels (clamp-modifier Hardswish, mla2d, tiled hole filling) against the exact-lattice and op tests,
# (2) A/B timings through the dev libraries, (3) the headline bench line, (4) the parity gates that failed or changed in call A
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 420 python -m pytest tests/test_lattice_gpu.py tests/test_ops_gpu.py -q -m gpu -k "lattice or one_hot or mbconv3 or lite_mla_block or fill_holes or fused_mlp" --timeout 300 > $O/b_kernels.txt 2>&1
tail -3 $O/b_kernels.txt
for lib in hswc dev; do
  ESAM3_DEV_LIB=build_dev/libesam3_$lib.so ESAM3_OP_REPEAT=20 timeout 200 python tools/evit_fused_bench.py > $O/b_evit_$lib.txt 2>&1
done
ESAM3_MLA2_OLD=1 ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 timeout 120 python tools/evit_fused_bench.py s2.ctx s3.ctx > $O/b_evit_mla2old.txt 2>&1
ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 timeout 120 python tools/cc_bench.py > $O/b_cc_new.txt 2>&1
ESAM3_CC_OLD=1 ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 timeout 120 python tools/cc_bench.py > $O/b_cc_old.txt 2>&1
grep -h "op_timed\|bit-exact" $O/b_evit_*.txt $O/b_cc_*.txt | cut -c1-160
ESAM3_BENCH_PROFILE_OUT=$O/b_bench_per_launch.json timeout 300 python bench.py --no-cpu-baseline > $O/b_bench.json 2> $O/b_bench.err
tail -c 600 $O/b_bench.json
timeout 900 python -m pytest tests/test_bf16_distribution.py tests/test_pcs.py tests/test_stage1_step.py -q -m gpu -rP --durations=12 --timeout 600 \
  -k "efficientvit or tinyvit-11m or pcs_bf16_distribution or config4 or two_training_steps or geometric" > $O/b_parity.txt 2>&1
tail -25 $O/b_parity.txt | cut -c1-200
