#!/bin/bash
# kernel-trace stats of bench.py under the given environment.  usage: tools/ktrace.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/kt_$TAG; mkdir -p $O
env RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/bench.json 2> $O/err.log
head -4 $O/k_kernel_stats.csv | cut -c1-160
