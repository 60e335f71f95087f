#!/bin/bash
# A/B: k_bulk_sym (64-bit storage, the caller's logD) before / after the two-barrier + early-issue change, n = 8192 and 16384
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
for rep in 1 2 3; do for lib in build_r4/lib_base.so in-tree; do
  if [ "$lib" = "in-tree" ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$PWD/$lib; fi
  echo "== $lib (rep $rep)"
  STORED=1 python tools/time_sweeps.py 8192 50 64 1000 | tail -1
  STORED=1 python tools/time_sweeps.py 8192 50 32 1000 | tail -1
done; done 2>&1 | tee $O/sym64_ab.txt
unset RC_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python tests/fuzz_parity.py 150 81000 2>&1 | tail -1
