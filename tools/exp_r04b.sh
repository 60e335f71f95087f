#!/bin/bash
# round-4 experiment batch B: births per batch (first sweep from random labels) and the moving regime under the same builds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04c; mkdir -p $O
B=$PWD/build_r4
for lib in cur birth24 birth72 birth96 birth144; do
  export RC_LIB_PATH=$B/lib_$lib.so
  python tools/uniform_init.py | tee -a $O/uniform_init.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'first sweep ms', d['first_sweep_ms'], d['first_sweeps'][0], d['checksum'])"
  MODE=incremental python tools/moving_rate.py | tee -a $O/moving_rate.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['mode'], 'sweeps/s %.0f' % d['sweeps_per_s'], 'rounds %.2f' % d['rounds_per_sweep'], d['checksum'])"
done
RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_random.py 2>&1 | tee $O/phases_random.txt
unset RC_LIB_PATH
MODE=full python tools/moving_rate.py | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-tree', d['mode'], 'sweeps/s %.0f' % d['sweeps_per_s'], 'rounds %.2f' % d['rounds_per_sweep'], d['checksum'])"
