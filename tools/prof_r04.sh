#!/bin/bash
# Round-4 evidence (run on the GPU box: bash tools/prof_r04.sh).  The program stands directly behind `--` in every rocprofv3 call; PMC
# counters are collected in their own passes with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
export RC_BENCH_NO_INCREMENTAL=1 RC_BENCH_NO_DEFAULTS=1 RC_BENCH_NO_KCAP512=1 RC_BENCH_NO_OTHER_CONFIGS=1
# 1. kernel-trace stats: headline leg alone (the driver's form: --steps 20), the moving regime, config 5
RC_BENCH_NO_MOVING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_headline_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_moving -o b -- python3 $R/tools/moving_rate.py > $O/moving_under_rocprof.json 2> $O/stats_moving.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o b -- python3 $R/tools/config5_rate.py 40 > $O/config5_under_rocprof.json 2> $O/stats_c5.err
# 2. PMC, one group per pass: the headline leg (k_bulk_syml2 + the stationary k_resolve)
export RC_BENCH_NO_TIMING=1 RC_BENCH_WINDOWS=2
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  RC_BENCH_NO_MOVING=1 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $O/pmc_headline_$name -o p -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/pmc_headline_$name.log 2>&1
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $O/pmc_moving_$name -o p -- python3 $R/tools/moving_rate.py > $O/pmc_moving_$name.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $O/pmc_c5_$name -o p -- python3 $R/tools/config5_rate.py 20 > $O/pmc_c5_$name.log 2>&1
done
# 3. FETCH_SIZE / WRITE_SIZE calibration on known byte counts
if [ ! -x $R/tools/calib_fetch ]; then hipcc --offload-arch=gfx950 -O3 -o $R/tools/calib_fetch $R/tools/calib_fetch.hip; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $O/calib_$ctr -o c -- $R/tools/calib_fetch 1024 > $O/calib_$ctr.txt 2> $O/calib_$ctr.err
done
unset RC_BENCH_NO_TIMING RC_BENCH_WINDOWS RC_BENCH_NO_INCREMENTAL RC_BENCH_NO_DEFAULTS RC_BENCH_NO_KCAP512 RC_BENCH_NO_OTHER_CONFIGS
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
for leg in ("headline", "moving", "c5"):
    with open(O + f"/counters_{leg}.txt", "w") as fh:
        for (k, c), (v, cnt) in sorted(per_kernel(f"pmc_{leg}_*").items()):
            if "k_bulk" in k or "k_resolve" in k:
                line = f"{k:42s} {c:24s} per-launch {v:14.1f}  (launches {cnt})"; print(leg, line); fh.write(line + "\\n")
                res[f"{leg}:{k}:{c}"] = [v, cnt]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for (k, c), (v, cnt) in per_kernel(f"calib_{ctr}").items():
        if "calib_" in k: res[f"calib:{k}:{c}"] = [v, cnt]
json.dump(res, open(O + "/pmc_summary_raw.json", "w"), indent=1)
PY
