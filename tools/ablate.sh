#!/bin/bash
# Builds timing-ablation variants of the library (results intentionally wrong) into build_exp/ — run on the build host;
# then on the GPU box: tools/ablate.sh run
cd "$(dirname "$0")/.."
SRC=redclust.jl_amd/csrc/redclust_hip.hip
if [ "$1" = "run" ]; then
  for v in base noatomic nolog nodir2 nolog_nodir2 nolog_nodir2_noatomic; do
    echo "== $v"; RC_DEBUG_FLAGS=2 RC_LIB_PATH=$PWD/build_exp/lib_$v.so python3 tools/time_sweeps.py 8192 50 64 300 2>&1 | tail -1
  done
  exit 0
fi
mkdir -p build_exp
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -o build_exp/lib_$1.so $SRC & }
build base ""
build noatomic "-DRC_ABL_NOATOMIC"
build nolog "-DRC_ABL_NOLOG"
build nodir2 "-DRC_ABL_NODIR2"
build nolog_nodir2 "-DRC_ABL_NOLOG -DRC_ABL_NODIR2"
build nolog_nodir2_noatomic "-DRC_ABL_NOLOG -DRC_ABL_NODIR2 -DRC_ABL_NOATOMIC"
wait
ls -la build_exp
