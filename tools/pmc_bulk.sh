#!/bin/bash
# PMC counters of the row-reduction kernel (separate passes; kernel-trace only).  usage: tools/pmc_bulk.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  env RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 "$@" rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/$name -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $OUT/$name.log 2>&1
done
python3 - <<PY
import glob, csv, collections
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        if "k_bulk" not in k and "k_resolve" not in k: continue
        a = acc[(k, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for (k, c), (v, cnt) in sorted(acc.items()):
        print(f"{k:42s} {c:24s} per-launch {v / cnt:14.1f}  (launches {cnt})")
PY
