#!/bin/bash
# experiment: what the flushes of k_bulk_sym32 cost (timing ablations of the diag build: ABL=32 drops the direction-1 flushes, 64 the direction-2 ones; wrong sums), blocks per CU, item size
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
export RC_LIB_PATH=$PWD/build_r4/lib_diag.so
run() { echo "== $*"; env "$@" timeout 300 python tools/config5_rate.py 60 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config5 sweeps/s %.0f  launch %.3f ms  frac %.3f %s' % (d['sweeps_per_s'], d['avg_launch_ms'], d['frac_of_8TBps'], d['kernel']))"; }
for rep in 1 2; do
run RC_SYM32_TR=16 RC_SYM32_BPC=4
run RC_SYM32_TR=16 RC_SYM32_BPC=4 ABL=32
run RC_SYM32_TR=16 RC_SYM32_BPC=4 ABL=64
run RC_SYM32_TR=16 RC_SYM32_BPC=4 ABL=96
run RC_SYM32_TR=16 RC_SYM32_BPC=4 RC_SYM_ITEM_TILES=16
run RC_SYM32_TR=16 RC_SYM32_BPC=4 RC_SYM_ITEM_TILES=4
run RC_SYM32_TR=16 RC_SYM32_BPC=3
run RC_SYM32_TR=16 RC_SYM32_BPC=3 ABL=96
done 2>&1 | tee $O/sym32b.txt
