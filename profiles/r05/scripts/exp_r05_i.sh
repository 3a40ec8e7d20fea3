#!/bin/bash
# round 5, block i: s_nop 1 in front of a wave's first slot only (paired kernel)
cd "$(dirname "$0")/../../.."
export AB_ONLY=pair
for rep in 1 2 3; do for v in base nop0; do
  SAFEOPT_HIP_LIB=$PWD/scripts/dev/ab/$v.so AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg"
done; done
