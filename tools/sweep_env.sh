#!/bin/bash
# usage: tools/sweep_env.sh "VAR=a VAR2=b" "VAR=c" ...   — bench.py under different env settings (no cpu baseline)
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sweeps/s %.0f  us/step %.1f  bulk_us %.1f (n=%d) frac_sweep %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['launches'], d['sweep_frac_of_hbm_peak']))"
done
