#!/bin/bash
# Round-4 verification on the GPU box: the whole GPU suite, the chaos build (block-dependent random delays inside the resolver's
# rounds: -DRC_CHAOS=15) on the moving-regime tests and the multi-thread tests, and the randomised differential checks.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O; LOG=$O/verify.log; : > $LOG
B=$PWD/build_r4
echo "== pytest -m gpu (in-tree build)" | tee -a $LOG
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee -a $LOG
echo "== chaos build: moving-regime parity subset + multi-thread tests" | tee -a $LOG
RC_LIB_PATH=$B/lib_chaos15.so timeout 1200 python -m pytest tests/test_gpu_derived_log.py tests/test_gpu_parity.py tests/test_gpu_threads.py tests/test_gpu_capacity.py -x -q -m gpu -k "derived_sweeps or derived_equals or many_small or synthetic_moving or long_trajectory or golden_sweeps or concurrent or resumed or record_sample" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee -a $LOG
echo "== fuzz (tests/fuzz_parity.py): sweeps, large, chains pipelined vs synchronous, chains vs the oracle's loop" | tee -a $LOG
for args in "500 71000" "40 72000 large" "400 73000 chains" "200 74000 oracle_chains" "12 75000 wide"; do
  timeout 1500 python tests/fuzz_parity.py $args 2>&1 | tail -2 | tee -a $LOG
done
echo "== fuzz on the chaos build" | tee -a $LOG
RC_LIB_PATH=$B/lib_chaos15.so timeout 1200 python tests/fuzz_parity.py 250 76000 2>&1 | tail -2 | tee -a $LOG
RC_LIB_PATH=$B/lib_chaos15.so timeout 1200 python tests/fuzz_parity.py 20 77000 large 2>&1 | tail -2 | tee -a $LOG
