#!/bin/bash
# long randomised campaign of round 4 (final build): sweeps, large, chains pipelined vs synchronous, chains vs the oracle's loop; chaos build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O; LOG=$O/soak.log; : > $LOG; B=$PWD/build_r4
for args in "1500 61000" "60 62000 large" "800 63000 chains" "400 64000 oracle_chains"; do
  timeout 3000 python tests/fuzz_parity.py $args 2>&1 | tail -1 | tee -a $LOG
done
echo "== chaos build" | tee -a $LOG
RC_LIB_PATH=$B/lib_chaos15.so timeout 2000 python tests/fuzz_parity.py 600 65000 2>&1 | tail -1 | tee -a $LOG
RC_LIB_PATH=$B/lib_chaos15.so timeout 2000 python tests/fuzz_parity.py 30 66000 large 2>&1 | tail -1 | tee -a $LOG
