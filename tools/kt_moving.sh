#!/bin/bash
# kernel trace of the moving regime.  usage: tools/kt_moving.sh [sigma] [kcap]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt_moving; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python3 $GRAFT_REPO_ROOT/tools/moving_sweeps.py ${1:-0.2} ${2:-512} 200 > $O/out.log 2> $O/err.log
tail -1 $O/out.log
python3 $GRAFT_REPO_ROOT/tools/timeline3.py $O/k_kernel_trace.csv
