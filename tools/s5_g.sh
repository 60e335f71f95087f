#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_capacity.py tests/test_gpu_derived_log.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
timeout 900 python tests/fuzz_parity.py ${FUZZ_N:-150} 55000 2>&1 | tail -2 | tee $O/fuzz.txt
bash tools/ab.sh build_r4/lib_head.so in-tree
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental.txt
RC_PROF_EVAL=1 RC_LIB_PATH=$B/lib_profeval.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental_eval.txt
