#!/bin/bash
# PMC counters of the row-reduction kernel and the resolver (separate passes; kernel-trace only).  usage: tools/pmc_r3.sh <tag> [VAR=value ...]
# (program directly after `--`: no env / bash hop under rocprofv3; the variables are exported here)
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
export RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 RC_BENCH_NO_MOVING=${RC_BENCH_NO_MOVING:-1} RC_BENCH_NO_DEFAULTS=1 RC_BENCH_NO_KCAP512=1 RC_BENCH_WINDOWS=2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/$name -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $OUT/$name.log 2>&1
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
