#!/bin/bash
# (record of a round-3 experiment: the -DPGP_* switches it builds with were removed from sweep_pair.hip after commit d1566ec;
#  check that commit out to re-run it -- results in profiles/r03/experiments.txt)
# round 3, experiment B: s_setprio for the evaluation phase / the younger half
cd "$(dirname "$0")/../.."
export AB_ONLY=pair
for rep in 1 2; do
for v in cur order2 o0p3 o1p3 o2p3 o2p3h1 o1h1; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | tail -3
done; done
SAFEOPT_HIP_LIB=scripts/dev/ab/o2p3s.so AB_TAG=o2p3s timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | tail -3
