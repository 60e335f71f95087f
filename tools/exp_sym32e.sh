#!/bin/bash
# experiment: work-item height of k_bulk_sym32 (direction-1 flushes per item) on config 5, diag build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
export RC_LIB_PATH=$PWD/build_r4/lib_diag.so
run() { echo "== $*"; env "$@" timeout 300 python tools/config5_rate.py 80 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config5 sweeps/s %.0f  launch %.3f ms  frac %.3f %s' % (d['sweeps_per_s'], d['avg_launch_ms'], d['frac_of_8TBps'], d['kernel']))"; }
for rep in 1 2 3 4; do
run RC_SYM_ITEM_TILES=8
run RC_SYM_ITEM_TILES=16
run RC_SYM_ITEM_TILES=32
run RC_SYM_ITEM_TILES=4
done 2>&1 | tee $O/sym32e.txt
