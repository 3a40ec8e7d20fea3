#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04i; mkdir -p $OUT; cd $R
for sep in 0 1; do
AB_SEP=$sep SAFEOPT_HIP_LIB=scripts/dev/ab/stamps4.so AB_ONLY=classic AB_TAG="stamps sep=$sep" python scripts/dev/ab_sweep.py 2 2>&1 | tail -2
done | tee $OUT/stamps.txt
