#!/bin/bash
# kernel trace of plain sweeps at a small size.  usage: tools/kt_small.sh <n> <K> [steps]
cd /tmp && export TMPDIR=/tmp
N=$1; K=$2; S=${3:-400}
O=$GRAFT_REPO_ROOT/gpurun_out/kt_small_$N; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python3 $GRAFT_REPO_ROOT/tools/time_sweeps.py $N $K 64 $S > $O/out.log 2> $O/err.log
tail -1 $O/out.log
python3 $GRAFT_REPO_ROOT/tools/timeline3.py $O/k_kernel_trace.csv
