#!/bin/bash
# sweep rate over problem sizes (stationary regime)
cd "$(dirname "$0")/.."
for cfg in "100 10" "1000 10" "2000 20" "4096 30" "6000 40" "8192 50" "12000 80" "16384 100"; do
  set -- $cfg
  env "${@:3}" python3 tools/time_sweeps.py $1 $2 64 600 | tail -1
done
