#!/bin/bash
# experiment: row pitch off the power of two (RC_LD_PAD, diag build): config 5 (HBM-resident, k_bulk_sym32) and the headline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
export RC_LIB_PATH=$PWD/build_r4/lib_diag.so
for rep in 1 2; do for pad in 0 64 256 576; do
  echo "== pad $pad (rep $rep)"
  RC_LD_PAD=$pad timeout 300 python tools/config5_rate.py 60 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config5 sweeps/s %.0f  launch %.3f ms  frac %.3f %s' % (d['sweeps_per_s'], d['avg_launch_ms'], d['frac_of_8TBps'], d['kernel']))"
  RC_LD_PAD=$pad timeout 300 python tools/time_sweeps.py 8192 50 64 2000 | tail -1
done; done 2>&1 | tee $O/ldpad.txt
RC_LD_PAD=256 timeout 600 python tests/fuzz_parity.py 60 61000 2>&1 | tail -1
