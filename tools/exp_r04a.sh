#!/bin/bash
# round-4 experiment batch A (run on the GPU box): LDS variants of k_bulk_syml2, resolver block size, death short path of batch_sim
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O
B=$PWD/build_r4
echo "== parity subset (in-tree build)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -x -q -m gpu -k "moving or births or many_small or golden or synthetic or headline_config or long_traj" 2>&1 | tail -3
echo "== k_bulk_syml2 variants"; python tools/syml_variants.py $B/lib_base.so $B/lib_swz.so $B/lib_rep2.so $B/lib_rep4.so $B/lib_exp16.so | tee $O/syml_variants.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d.get('lib','')[-16:], 'sweeps/s %.0f' % d.get('sweeps_per_s', 0), 'kernel pipeline %.1f us alone %.1f us' % (d.get('kernel_us_pipeline', 0), d.get('kernel_us_alone', 0)), d.get('checksum'), d.get('error', ''))"
echo "== moving regime"
for cfg in "in-tree:full" "in-tree:incremental" "$B/lib_res1024.so:incremental" "$B/lib_res1024.so:full"; do lib=${cfg%%:*}; mode=${cfg#*:}; if [ "$lib" = "in-tree" ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$lib; fi; MODE=$mode python tools/moving_rate.py | tee -a $O/moving_rate.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['lib'][-16:], d['mode'], 'sweeps/s %.0f' % d['sweeps_per_s'], 'blocking %.0f' % d['blocking_sweeps_per_s'], 'changes %.1f rounds %.2f' % (d['changes_per_sweep'], d['rounds_per_sweep']), d['checksum'])"; done
unset RC_LIB_PATH
echo "== resolver phases"
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental.txt
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof.so python tools/prof_resolve_moving.py 0.2 0 2>&1 | tee $O/phases_full.txt
RC_PROF_SIM=1 RC_LIB_PATH=$B/lib_prof1024.so python tools/prof_resolve_moving.py 0.2 0 incremental 2>&1 | tee $O/phases_incremental_1024.txt
