#!/bin/bash
# Round-2 evidence for the default configuration: kernel-trace stats, PMC HBM traffic (separate passes), FETCH_SIZE
# calibration on a known byte count, instruction counters, and the bench line.  Run on the GPU box: tools/prof_r02.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
# headline configuration only (the moving-regime and incremental-mode legs of bench.py launch the same kernel names on
# other contexts and would be averaged in), then the whole default run
RC_BENCH_NO_INCREMENTAL=1 RC_BENCH_NO_MOVING=1 RC_BENCH_NO_DEFAULTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_full_under_rocprof.json 2> $O/stats_full.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 RC_BENCH_NO_MOVING=1 RC_BENCH_NO_DEFAULTS=1 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $O/pmc_$ctr -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 > $O/pmc_$ctr.json 2> $O/pmc_$ctr.err
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $O/calib_$ctr -o c -- $R/tools/calib_fetch 1024 > $O/calib_$ctr.txt 2> $O/calib_$ctr.err
done
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 RC_BENCH_NO_MOVING=1 RC_BENCH_NO_DEFAULTS=1 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $O/sq_$name -o r -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/sq_$name.log 2>&1
done
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python3 - <<PY
import glob, csv, collections, json
O = "$O"
def per_kernel(pattern):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(O + "/" + pattern + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[(k, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: (v / c, c) for k, (v, c) in acc.items()}
res = {}
for pat in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "calib_FETCH_SIZE", "calib_WRITE_SIZE"):
    for (k, c), (v, cnt) in per_kernel(pat).items():
        if any(x in k for x in ("k_bulk", "k_resolve", "calib_")): res[f"{pat}:{k}:{c}_per_launch"] = [v, cnt]
json.dump(res, open(O + "/pmc_summary_raw.json", "w"), indent=1)
for k, v in sorted(res.items()): print(k, v)
with open(O + "/sq_summary.txt", "w") as fh:
    for (k, c), (v, cnt) in sorted(per_kernel("sq_*").items()):
        if "k_bulk" in k or "k_resolve" in k:
            line = f"{k:42s} {c:24s} per-launch {v:14.1f}  (launches {cnt})"; print(line); fh.write(line + "\\n")
PY
tail -c 400 $O/bench.json
