#!/bin/bash
# moving regime in full mode with the widened re-layout rule: the new test, the rates (in-tree against the previous build), the suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "relayout" 2>&1 | tail -3
for rep in 1 2 3; do for lib in build_r4/lib_base.so in-tree; do
  if [ "$lib" = "in-tree" ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$PWD/$lib; fi
  echo "== $lib (rep $rep)"
  python tools/moving_rate.py | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['mode'], 'sweeps/s %.0f' % d['sweeps_per_s'], ['%.0f' % r for r in d['rates']], 'blocking %.0f' % d['blocking_sweeps_per_s'], d['kernel'], 'reduction %.0f us' % d['reduction_us'], 'K', d['K'], d['checksum'])"
done; done 2>&1 | tee $O/moving_relayout_ab.txt
unset RC_LIB_PATH
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -2
timeout 900 python tests/fuzz_parity.py 150 83000 2>&1 | tail -1
