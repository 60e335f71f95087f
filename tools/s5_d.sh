#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
python -m pytest tests/test_gpu_logs.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
bash tools/ab.sh build_r4/lib_b6a4593.so in-tree
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental.txt
RC_PROF_COMMIT=1 RC_LIB_PATH=$B/lib_profcommit.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental_commit.txt
