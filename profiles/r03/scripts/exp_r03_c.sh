#!/bin/bash
# (record of a round-3 experiment: the -DPGP_* switches it builds with were removed from sweep_pair.hip after commit d1566ec;
#  check that commit out to re-run it -- results in profiles/r03/experiments.txt)
# round 3, experiment C: LDS-DMA issue interleaved with the slot sequence
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for v in cur dm2 dm2p3 dm1o1 dm1o1p3 dm1o0; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib timeout 300 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | tail -3 | sed "s/$/  [$v]/"
done; done
SAFEOPT_HIP_LIB=scripts/dev/ab/dm2s.so AB_ONLY=pair AB_TAG=dm2s timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | tail -3
