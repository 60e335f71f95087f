#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
RC_PROF_COMMIT=1 RC_LIB_PATH=$B/lib_profcommit.so python tools/prof_resolve_random.py 2>&1 | tee $O/phases_random_commit.txt
RC_PROF_EVAL=1 RC_LIB_PATH=$B/lib_profeval.so python tools/prof_resolve_random.py 2>&1 | tee $O/phases_random_eval.txt
