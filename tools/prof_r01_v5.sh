#!/bin/bash
# Round-1 final evidence for the default configuration (derived logD, k_bulk_syml): kernel-trace stats, PMC HBM traffic,
# and the bench line.  Run on the GPU box: tools/prof_r01_v5.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v5; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $O/pmc_$ctr -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 > $O/pmc_$ctr.json 2> $O/pmc_$ctr.err
done
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
