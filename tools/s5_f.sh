#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
RC_PROF_COMMIT=1 RC_LIB_PATH=$B/lib_profcommit.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental_commit.txt
RC_PROF_EVAL=1 RC_LIB_PATH=$B/lib_profeval.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental_eval.txt
