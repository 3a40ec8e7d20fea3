#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04g; mkdir -p $OUT; cd $R
for rep in 1 2; do
  AB_ONLY=classic AB_TAG="cur" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  for v in salu32 salu64 valu32; do
  SAFEOPT_HIP_LIB=scripts/dev/ab/$v.so AB_ONLY=classic AB_TAG="$v" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
