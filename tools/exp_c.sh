#!/bin/bash
# fat-wave experiment: separate launches (RC_FUSED=0), row reduction capped per CU by an LDS pad, resolver at 128 VGPRs
cd "$(dirname "$0")/.."
T="python3 tools/time_sweeps.py 8192 50 64 300"
export RC_FUSED=0
run() { echo "== $1"; shift; env "$@" $T | tail -1; }
run "a(LOGS8,170v) percu2 pad26K" RC_LIB_PATH=$PWD/build_exp/lib_fat_a.so RC_SYMW_PER_CU=2 RC_SYML_PAD=26624
run "a percu2 nopad"              RC_LIB_PATH=$PWD/build_exp/lib_fat_a.so RC_SYMW_PER_CU=2
run "b(LOGS4,150v) percu3 pad4K"  RC_LIB_PATH=$PWD/build_exp/lib_fat_b.so RC_SYMW_PER_CU=3 RC_SYML_PAD=4096
run "b percu2 pad26K"             RC_LIB_PATH=$PWD/build_exp/lib_fat_b.so RC_SYMW_PER_CU=2 RC_SYML_PAD=26624
run "b percu3 nopad"              RC_LIB_PATH=$PWD/build_exp/lib_fat_b.so RC_SYMW_PER_CU=3
run "d(LOGS2,128v,res128) percu4" RC_LIB_PATH=$PWD/build_exp/lib_fat_d.so
run "d percu2 pad4K"              RC_LIB_PATH=$PWD/build_exp/lib_fat_d.so RC_SYMW_PER_CU=2 RC_SYML_PAD=4096
