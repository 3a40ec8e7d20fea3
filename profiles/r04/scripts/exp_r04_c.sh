#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
for rep in 1 2; do
  SGP_NO_NARROW=1 AB_ONLY=classic AB_TAG="cur no_narrow" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_ONLY=classic AB_TAG="cur early" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  SAFEOPT_HIP_LIB=scripts/dev/ab/anlate.so AB_ONLY=classic AB_TAG="late" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
done | tee $OUT/ab.txt
