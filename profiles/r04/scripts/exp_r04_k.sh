#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04k; mkdir -p $OUT; cd $R
for rep in 1 2; do
  for sep in 1 0; do
  AB_SEP=$sep AB_ONLY=classic AB_TAG="dma before slots, sep=$sep" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_SEP=$sep SAFEOPT_HIP_LIB=scripts/dev/ab/dmatop.so AB_ONLY=classic AB_TAG="dma at top, sep=$sep" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
