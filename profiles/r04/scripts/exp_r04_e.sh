#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04e; mkdir -p $OUT; cd $R
for v in 1 0; do
SGP_NO_NARROW=$v SAFEOPT_HIP_LIB=scripts/dev/ab/stamps4.so AB_ONLY=classic AB_TAG="stamps no_narrow=$v" python scripts/dev/ab_sweep.py 2 2>&1 | tail -3
done | tee $OUT/stamps.txt
