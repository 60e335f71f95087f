#!/bin/bash
# after a kernel change (round 4): GPU suite, short randomised checks (in-tree + chaos build, incl. the large leg), A/B of the stationary
# and moving rates against another build on the same box, phase tables.  usage: bash tools/check_r04.sh [other_lib.so]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; B=$PWD/build_r4
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
timeout 900 python tests/fuzz_parity.py ${FUZZ_N:-200} 57000 2>&1 | tail -1
timeout 1500 python tests/fuzz_parity.py 30 58000 large 2>&1 | tail -1
RC_LIB_PATH=$B/lib_chaos15.so timeout 900 python tests/fuzz_parity.py ${FUZZ_N:-200} 59000 2>&1 | tail -1
if [ -n "$1" ]; then bash tools/ab.sh "$1" in-tree; fi
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | head -12
python tools/uniform_init.py | tail -1
