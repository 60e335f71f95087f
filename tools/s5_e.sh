#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee $O/pytest_gpu.txt
timeout 900 python tests/fuzz_parity.py ${FUZZ_N:-250} 53000 2>&1 | tail -2 | tee $O/fuzz.txt
RC_LIB_PATH=$B/lib_chaos15.so timeout 900 python tests/fuzz_parity.py ${FUZZ_N:-250} 54000 2>&1 | tail -2 | tee -a $O/fuzz.txt
bash tools/ab.sh build_r4/lib_head.so in-tree
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental.txt
RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_random.py 2>&1 | tee $O/phases_random.txt
python tools/uniform_init.py | tail -1 | tee $O/uniform_init.json
