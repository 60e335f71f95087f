#!/bin/bash
# chaos builds (block-dependent random delays inside the resolver's rounds) against the moving-regime parity tests
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "== chaos sites mask $v"
  RC_LIB_PATH=$PWD/build_exp/lib_chaos$v.so timeout 600 python -m pytest tests/test_gpu_derived_log.py tests/test_gpu_parity.py -x -q -m gpu -k "derived_sweeps or derived_equals or many_small or synthetic_moving or long_trajectory or golden_sweeps" 2>&1 | tail -3
done
