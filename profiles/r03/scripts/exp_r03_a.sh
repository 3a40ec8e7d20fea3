#!/bin/bash
# (record of a round-3 experiment: the -DPGP_* switches it builds with were removed from sweep_pair.hip after commit d1566ec;
#  check that commit out to re-run it -- results in profiles/r03/experiments.txt)
# round 3, experiment A: phase order / pair mapping / DMA placement of the paired sweep, and its ablation
cd "$(dirname "$0")/../.."
export AB_ONLY=pair
for rep in 1 2; do
for v in cur order1 order2 adj dmalate; do
  lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | tail -3
done; done
for m in 0 1 2 4 8 16 32 7 15; do
  SAFEOPT_HIP_LIB=scripts/dev/ab/instr.so SGP_ABLATE=$m AB_TAG="ablate $m" timeout 200 python scripts/dev/ab_sweep.py 3 2>&1 | tail -1
done
