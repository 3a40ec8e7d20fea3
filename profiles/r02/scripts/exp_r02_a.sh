#!/bin/bash
# Round-2 experiment A (GPU box): does a start-up phase offset between the two
# workgroups of a CU change the sweep time?  + the re-evaluate-vs-cache A/B.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_a
mkdir -p $OUT
SGP_HIPCC_FLAGS=-DSGP_INSTRUMENT python -m safeopt_amd.build --force > /dev/null || exit 1
for c in 3 2; do for k in 0 16 32 48 64 96; do
  echo -n "skew $k: "; SGP_SKEW=$k timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
done; done | tee $OUT/skew.txt
python -m safeopt_amd.build --force > /dev/null
for c in 3 4 5; do
  echo -n "cache: ";  timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
  echo -n "re-eval: "; SGP_NO_KVCACHE=1 timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
done | tee $OUT/cache_ab.txt
