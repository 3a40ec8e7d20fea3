#!/bin/bash
# (record of a round-3 experiment: the -DPGP_* switches it builds with were removed from sweep_pair.hip after commit d1566ec;
#  check that commit out to re-run it -- results in profiles/r03/experiments.txt)
# round 3, experiment F: both halves evaluate first (barrier-synchronous VALU burst), operands before the evaluation
cd "$(dirname "$0")/../.."
for v in cur oe o0; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib timeout 300 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | tail -3 | sed "s/$/  [$v]/"
done
export AB_ONLY=pair
for v in cur oe o0; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 300 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | tail -3
done
for v in cur_s oe_s; do SAFEOPT_HIP_LIB=scripts/dev/ab/$v.so AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | tail -3; done
