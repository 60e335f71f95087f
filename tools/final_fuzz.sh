#!/bin/bash
# last randomised campaign of the round on the final library (new seeds)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O; LOG=$O/final_fuzz.log; : > $LOG
for args in "700 91000" "30 92000 large" "300 93000 chains" "150 94000 oracle_chains" "10 95500 wide"; do
  timeout 1200 python tests/fuzz_parity.py $args 2>&1 | tail -1 | tee -a $LOG
done
RC_LIB_PATH=$PWD/build_r4/lib_chaos15.so timeout 900 python tests/fuzz_parity.py 200 96000 2>&1 | tail -1 | tee -a $LOG
