#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; B=$PWD/build_r4
for lib in lib_prof_b6a4593.so lib_prof.so; do
  echo "== $lib (stationary resolver phases, pipelined)"; RC_LIB_PATH=$B/$lib python tools/prof_resolve.py 2>&1 | tail -14
  echo "== $lib blocking"; BLOCKING=1 RC_LIB_PATH=$B/$lib python tools/prof_resolve.py 2>&1 | tail -7
done | tee $O/stationary_phases.txt
