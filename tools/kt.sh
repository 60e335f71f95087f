#!/bin/bash
# kernel trace of bench.py (no stats) + steady-state window.  usage: tools/kt.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/kt_$TAG; mkdir -p $O
env RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/bench.json 2> $O/err.log
head -3 $O/k_kernel_stats.csv | cut -c1-150
python3 $GRAFT_REPO_ROOT/tools/timeline3.py $O/k_kernel_trace.csv
