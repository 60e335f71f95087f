#!/bin/bash
# experiment (A): resolver at 128 VGPRs + row reduction capped at 3 blocks per CU by an LDS pad (room for a resolver block)
cd "$(dirname "$0")/.."
T="python3 tools/time_sweeps.py 8192 50 64 300"
echo "== base";                RC_LIB_PATH=$PWD/build_exp/lib_base.so $T | tail -1
echo "== res128";              RC_LIB_PATH=$PWD/build_exp/lib_res128.so $T | tail -1
echo "== res128 percu3";       RC_SYMW_PER_CU=3 RC_LIB_PATH=$PWD/build_exp/lib_res128.so $T | tail -1
echo "== res128 percu3 pad4.5K (3 bulk blocks + resolver per CU)"; RC_SYMW_PER_CU=3 RC_SYML_PAD=4608 RC_LIB_PATH=$PWD/build_exp/lib_res128.so $T | tail -1
echo "== res128 percu3 pad4.5K res256"; RC_RES_THREADS=256 RC_SYMW_PER_CU=3 RC_SYML_PAD=4608 RC_LIB_PATH=$PWD/build_exp/lib_res128.so $T | tail -1
echo "== res128 percu2 pad"; RC_SYMW_PER_CU=2 RC_SYML_PAD=4608 RC_LIB_PATH=$PWD/build_exp/lib_res128.so $T | tail -1
echo "== base percu3 pad4.5K"; RC_SYMW_PER_CU=3 RC_SYML_PAD=4608 RC_LIB_PATH=$PWD/build_exp/lib_base.so $T | tail -1
