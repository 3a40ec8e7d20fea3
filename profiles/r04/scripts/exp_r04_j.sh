#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04j; mkdir -p $OUT; cd $R
for rep in 1 2; do
  AB_ONLY=classic AB_TAG="tables early" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  SAFEOPT_HIP_LIB=scripts/dev/ab/anlate.so AB_ONLY=classic AB_TAG="tables late" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  SGP_NO_NARROW=1 AB_ONLY=classic AB_TAG="tables no-narrow" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  SGP_NO_NARROW=1 AB_SEP=0 AB_ONLY=classic AB_TAG="generic no-narrow" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
done | tee $OUT/ab.txt
