#!/bin/bash
# narrowing of a wide context under rc_run_chain: the wide tests, randomised wide chains, then the whole GPU suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide.py -m gpu -x -q 2>&1 | tail -5
timeout 1200 python tools/fuzz_wide_chains.py ${1:-20} ${2:-95000} 2>&1 | tee $O/fuzz_wide_chains.txt | tail -24
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -3
