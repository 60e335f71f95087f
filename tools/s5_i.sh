#!/bin/bash
# full-mode moving regime: resolver block size x row-reduction blocks per CU while labels move (diag build)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
export RC_LIB_PATH=$PWD/build_r4/lib_diag.so
for cfg in "512:2" "768:1" "768:2" "1024:1" "512:1"; do
  export RC_RES_THREADS=${cfg%%:*} RC_S2_ALT_PER_CU=${cfg#*:}
  MODE=full timeout 300 python tools/moving_rate.py | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('res threads $RC_RES_THREADS, reduction blocks per CU $RC_S2_ALT_PER_CU:', 'sweeps/s %.0f' % d['sweeps_per_s'], ['%.0f' % r for r in d['rates']], 'blocking %.0f' % d['blocking_sweeps_per_s'], d['kernel'], 'reduction %.0f us' % d['reduction_us'], d['checksum'])"
done 2>&1 | tee $O/full_mode_geometry.txt
