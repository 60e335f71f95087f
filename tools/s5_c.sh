#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
RC_LIB_PATH=$B/lib_prof.so python tools/exp_tentative.py incremental 2>&1 | tee $O/exp_tentative_incremental.txt
RC_LIB_PATH=$B/lib_prof.so python tools/exp_tentative.py 2>&1 | tee $O/exp_tentative_full.txt
