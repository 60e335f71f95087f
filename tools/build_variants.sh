#!/bin/bash
# diagnostic builds of the library into build_r4/ (git-ignored; travels to the GPU box): bash tools/build_variants.sh [names...]
# names: prof (per-phase stamps + batch_sim counters), chaos15 (block-dependent random delays in the resolver's rounds), simold
# (the one-entry-at-a-time batch simulation of round 3 everywhere), profold (prof + simold), profcommit (column 2 = table rebuild inside the commit)
cd "$(dirname "$0")/../redclust.jl_amd/csrc" || exit 1
B=../../build_r4; mkdir -p $B
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
for v in "${@:-prof chaos15}"; do for name in $v; do
  case $name in
    prof) D="-DRC_DIAG -DRC_PROF_SYML -DRC_PROF_SIM";;
    profold) D="-DRC_DIAG -DRC_PROF_SYML -DRC_PROF_SIM -DRC_SIM_OLD";;
    chaos15) D="-DRC_DIAG -DRC_CHAOS=15";;
    diag) D="-DRC_DIAG";;
    profcommit) D="-DRC_DIAG -DRC_PROF_SYML -DRC_PROF_COMMIT";;
    profeval) D="-DRC_DIAG -DRC_PROF_SYML -DRC_PROF_EVAL";;
    simold) D="-DRC_SIM_OLD";;
    *) echo "unknown variant $name"; exit 1;;
  esac
  $CC $D -o $B/lib_$name.so redclust_hip.hip 2>&1 | grep -v "hip-link" &
done; done
wait
ls -la $B
