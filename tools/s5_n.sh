#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; B=$PWD/build_r4
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_derived_log.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for rep in 1 2 3; do for lib in build_r4/lib_head.so in-tree; do
  if [ "$lib" = "in-tree" ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$PWD/$lib; fi
  echo -n "$lib: "; python tools/time_sweeps.py 8192 50 64 3000 | tail -1
done; done
