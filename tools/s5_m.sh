#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; B=$PWD/build_r4
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
timeout 900 python tests/fuzz_parity.py 200 57000 2>&1 | tail -1
timeout 1500 python tests/fuzz_parity.py 30 58000 large 2>&1 | tail -1
RC_LIB_PATH=$B/lib_chaos15.so timeout 900 python tests/fuzz_parity.py 200 59000 2>&1 | tail -1
for m in full incremental; do MODE=$m python tools/moving_rate.py | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['mode'], 'sweeps/s %.0f' % d['sweeps_per_s'], ['%.0f' % r for r in d['rates']], d['kernel'], d['checksum'])"; done
python tools/uniform_init.py | tail -1
python tools/time_sweeps.py 8192 50 64 2000 | tail -1
