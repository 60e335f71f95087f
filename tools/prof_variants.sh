#!/bin/bash
# usage: tools/prof_variants.sh "0 1 2 3 4 7"   — rocprofv3 kernel stats of bench.py under RC_DEBUG_FLAGS ablations
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in $1; do
  export RC_DEBUG_FLAGS=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$f -o b -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/abl_$f.json 2> $R/gpurun_out/abl_$f.err
  echo "== flags=$f"; grep -E "k_resolve|k_bulk|k_zero" $R/gpurun_out/abl_$f/b_kernel_stats.csv | cut -d, -f1-4,6,7
  python3 -c "import json;d=json.load(open('$R/gpurun_out/abl_$f.json'));print('sweeps/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'bulk_ms',round(d['roofline']['avg_launch_ms'],4))"
done
