#!/bin/bash
# A/B on ONE box: usage  bash tools/ab.sh libA.so libB.so   ("in-tree" = the in-tree build).  Alternates the two libraries:
# stationary headline rate (2000 sweeps), moving regime full and incremental
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = "in-tree" ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$PWD/$lib; fi
  echo "== $lib (rep $rep)"
  python tools/time_sweeps.py 8192 50 64 2000 | tail -1
  for m in full incremental; do MODE=$m python tools/moving_rate.py | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['mode'], 'sweeps/s %.0f' % d['sweeps_per_s'], ['%.0f' % r for r in d['rates']], 'blocking %.0f' % d['blocking_sweeps_per_s'], d['kernel'], 'reduction %.0f us' % d['reduction_us'], d['checksum'])"; done
done; done 2>&1 | tee $O/ab.txt
