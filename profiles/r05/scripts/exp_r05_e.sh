#!/bin/bash
# round 5, block e: what the cache's own memory instructions cost (timing variants: wrong results)
cd "$(dirname "$0")/../../.."
OUT=gpurun_out/exp_r05_e.txt; : > $OUT
export AB_ONLY=pair
for rep in 1 2; do
for v in cur kcnl kcns kcnn; do
  lib=$PWD/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$PWD/safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 5 2>&1 | grep "^cfg" >> $OUT
done
SGP_COV_CACHE=0 AB_TAG=off timeout 200 python scripts/dev/ab_sweep.py 3 5 2>&1 | grep "^cfg" >> $OUT
done
cat $OUT
