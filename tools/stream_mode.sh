#!/bin/bash
# sweep rate with the resolvers on one stream (RC_RES_ONE_STREAM=1) or on the sweep's parity stream (0), by problem size
cd "$(dirname "$0")/.."
for cfg in "1000 10" "2000 20" "3000 25" "4096 30" "6000 40" "8192 50"; do
  set -- $cfg
  for m in 0 1; do
    echo -n "RC_RES_ONE_STREAM=$m  "; RC_RES_ONE_STREAM=$m python3 tools/time_sweeps.py $1 $2 64 400 | tail -1
  done
done
