#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; B=$PWD/build_r4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_capacity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
timeout 900 python tests/fuzz_parity.py ${FUZZ_N:-150} 56000 2>&1 | tail -2
bash tools/ab.sh build_r4/lib_head.so in-tree
echo "== stationary resolver phases"; RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve.py 2>&1 | tail -7
