#!/bin/bash
# round-4: the GPU suite, then the evidence run (rocprofv3 stats + PMC) and the bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_gpu.txt
bash tools/prof_r04.sh > $O/prof_r04.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench steps20 rc=$?"
tail -3 $O/prof_r04.log
