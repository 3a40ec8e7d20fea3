#!/bin/bash
# round 4, call B: narrow groups in the 4-wave sweep -- tests + A/B at config 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
true

for rep in 1 2; do
  for v in 1 2 0; do
    SGP_NO_NARROW=$v AB_ONLY=classic AB_TAG="no_narrow=$v" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
