#!/bin/bash
# Build ablation variants of gemm256p.hip into build_dev/libesam3_<name>.so (git-ignored, shipped by gpurun).
# usage: tools/dev_variants.sh name1:"-DFLAG ..." name2:"..."      then   ESAM3_DEV_LIB=build_dev/libesam3_name1.so python tools/bench_gemm.py
set -e
cd "$(dirname "$0")/../efficientsam3_amd/csrc"
make -s
mkdir -p ../../build_dev
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable $flags -c gemm256p.hip -o ../../build_dev/gemm256p_$name.o
  objs=$(ls build/*.o | grep -v gemm256p.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_dev/libesam3_$name.so $objs ../../build_dev/gemm256p_$name.o
  echo "built build_dev/libesam3_$name.so ($flags)"
done
