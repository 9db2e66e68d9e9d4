#!/bin/bash
# Build ablation variants of one kernel source (SRC=gemm256p by default, e.g. SRC=conv3x3_narrow) into build_dev/libesam3_<name>.so (git-ignored, shipped by gpurun).
# usage: tools/dev_variants.sh name1:"-DFLAG ..." name2:"..."      then   ESAM3_DEV_LIB=build_dev/libesam3_name1.so python tools/bench_gemm.py
set -e
cd "$(dirname "$0")/../efficientsam3_amd/csrc"
make -s
mkdir -p ../../build_dev
SRC=${SRC:-gemm256p}
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable $flags -c $SRC.hip -o ../../build_dev/${SRC}_$name.o
  objs=$(ls build/*.o | grep -v "/$SRC.o" | grep -v "/dev_")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_dev/libesam3_$name.so $objs ../../build_dev/${SRC}_$name.o
  echo "built build_dev/libesam3_$name.so ($flags)"
done
