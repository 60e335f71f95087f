#!/bin/bash
# A/B on one box: k_bulk_sym32 (16-row tiles) with / without the merged direction-2 flushes and the scalar row bases
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 300 python tools/config5_rate.py 60 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config5 sweeps/s %.0f  launch %.3f ms  frac %.3f %s' % (d['sweeps_per_s'], d['avg_launch_ms'], d['frac_of_8TBps'], d['kernel']))"; }
for rep in 1 2 3; do for lib in diag diag_nocomb diag_nosaddr diag_nocomb_nosaddr; do
export RC_LIB_PATH=$PWD/build_r4/lib_$lib.so
echo "#### $lib (rep $rep)"
run RC_SYM32_TR=16 RC_SYM32_BPC=4
run RC_SYM32_TR=16 RC_SYM32_BPC=3
done; done 2>&1 | tee $O/sym32d.txt
