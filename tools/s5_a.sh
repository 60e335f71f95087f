#!/bin/bash
# session baseline: GPU suite, resolver phase tables (moving regime, both modes), first sweep from random labels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
B=$PWD/build_r4
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental.txt
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 2>&1 | tee $O/phases_full.txt
RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_random.py 2>&1 | tee $O/phases_random.txt
for m in full incremental; do MODE=$m python tools/moving_rate.py | tail -1 | tee -a $O/moving_rate.jsonl; done
python tools/uniform_init.py | tail -1 | tee $O/uniform_init.json
