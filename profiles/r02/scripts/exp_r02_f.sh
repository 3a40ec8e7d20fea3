#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_f
mkdir -p $OUT
SGP_HIPCC_FLAGS=-DSGP_SWEEP_WAVES8 python -m safeopt_amd.build --force > /dev/null || exit 1
for c in 3 2 4 5; do for w in 4 8; do
  echo -n "waves $w: "; SGP_SWEEP_WAVES=$w timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
done; done | tee $OUT/waves.txt
python -m safeopt_amd.build --force > /dev/null
