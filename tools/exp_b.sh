#!/bin/bash
cd "$(dirname "$0")/.."
T="python3 tools/time_sweeps.py 8192 50 64 300"
export RC_LIB_PATH=$PWD/build_exp/lib_base.so
echo "== base";                $T | tail -1
echo "== one stream";          RC_ONE_STREAM=1 $T | tail -1
echo "== no prefetch";         RC_NO_PREFETCH=1 $T | tail -1
echo "== one bulk stream";     RC_ONE_BULK_STREAM=1 $T | tail -1
echo "== one stream, res256";  RC_ONE_STREAM=1 RC_RES_THREADS=256 $T | tail -1
echo "== one stream, res512";  RC_ONE_STREAM=1 RC_RES_THREADS=512 $T | tail -1
