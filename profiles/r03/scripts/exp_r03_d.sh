#!/bin/bash
# (record of a round-3 experiment: the -DPGP_* switches it builds with were removed from sweep_pair.hip after commit d1566ec;
#  check that commit out to re-run it -- results in profiles/r03/experiments.txt)
# round 3, experiment D: on top of H0-only interleaved DMA + eval prio 3: order / prio / mapping
cd "$(dirname "$0")/../.."
export AB_ONLY=pair
for rep in 1 2; do
for v in cur n_o1 n_o2 n_p1 n_p2 n_h1p1 n_adj; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 300 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | tail -3
done; done
