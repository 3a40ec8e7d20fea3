#!/bin/bash
# round 5, block j: where do the riders' 0.67 ms go?  (config 3 with the shared factor: one leader + two riders in
# k_sweep_pair<2,0,true,2,0> against ONE GP on its own in <2,0,true,0,0>); compile-time ablation of the rider instance
cd "$(dirname "$0")/../../.."
export AB_ONLY=pair
L=$PWD/scripts/dev/ab
SAFEOPT_HIP_LIB=$L/nop0.so AB_G=1 AB_TAG="one GP alone" timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg"
SAFEOPT_HIP_LIB=$L/nop0.so AB_G=2 AB_SHARE=1 AB_TAG="shared, leader + 1 rider" timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg"
SAFEOPT_HIP_LIB=$L/nop0.so AB_SHARE=1 AB_TAG="shared, leader + 2 riders" timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg"
for m in 1 2 4 16 32 36; do
  SAFEOPT_HIP_LIB=$L/abl$m.so AB_G=1 AB_TAG="one GP alone, ablate $m" timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg"
  SAFEOPT_HIP_LIB=$L/abl$m.so AB_SHARE=1 AB_TAG="shared, ablate $m" timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg"
done
