#!/bin/bash
# k_bulk_sym32 with 16-row tiles as the default: GPU suite, randomised checks (a third of the cases use 32-bit storage), A/B against the
# round-1 tile shape on one box (diag build)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -3
timeout 900 python tests/fuzz_parity.py 150 64000 2>&1 | tail -1
timeout 900 python tests/fuzz_parity.py 16 65000 large 2>&1 | tail -1
run() { echo "== $*"; env "$@" timeout 300 python tools/config5_rate.py 60 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config5 sweeps/s %.0f  launch %.3f ms  frac %.3f %s' % (d['sweeps_per_s'], d['avg_launch_ms'], d['frac_of_8TBps'], d['kernel']))"; }
export RC_LIB_PATH=$PWD/build_r4/lib_diag.so
for rep in 1 2 3; do
run RC_SYM32_TR=32
run RC_SYM32_TR=16
done 2>&1 | tee $O/sym32_final_ab.txt
