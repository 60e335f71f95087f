#!/bin/bash
# which allocation is read before it is written?  (-DRC_POISON build; RC_POISON_ONLY selects allocation sites by text)
cd "$(dirname "$0")/.."
export RC_LIB_PATH=$PWD/build_exp/lib_poison.so
T="timeout 300 python -m pytest tests/test_gpu_derived_log.py -x -q -m gpu"
for site in "$@"; do
  printf "%-22s " "$site"; RC_POISON_ONLY="$site" $T 2>&1 | tail -1
done
