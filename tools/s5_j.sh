#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O; B=$PWD/build_r4
timeout 1500 python tests/fuzz_parity.py 40 42000 large 2>&1 | tail -2 | tee $O/verify_large.log
RC_LIB_PATH=$B/lib_chaos15.so timeout 1200 python tests/fuzz_parity.py 20 46000 large 2>&1 | tail -2 | tee -a $O/verify_large.log
timeout 900 python -m pytest tests/test_gpu_capacity.py tests/test_gpu_wide.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tee -a $O/verify_large.log
