#!/bin/bash
# round 5, block f: cache loads that hit the L2 (every slot aliased to slot 0; wrong results)
cd "$(dirname "$0")/../../.."
OUT=gpurun_out/exp_r05_f.txt; : > $OUT
export AB_ONLY=pair
for rep in 1 2; do
for v in cur kcal kcalns; do
  lib=$PWD/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$PWD/safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 5 2>&1 | grep "^cfg" >> $OUT
done
done
cat $OUT
