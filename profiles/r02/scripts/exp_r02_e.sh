#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_e
mkdir -p $OUT
SGP_HIPCC_FLAGS=-DSGP_INSTRUMENT python -m safeopt_amd.build --force > /dev/null || exit 1
for c in 3 2; do for a in 0 1 2 4 8 10 12 15 32; do
  SGP_ABLATE=$a timeout 120 python scripts/ablate.py $c 3 2>&1 | tail -1
done; done | tee $OUT/ablate.txt
python -m safeopt_amd.build --force > /dev/null
