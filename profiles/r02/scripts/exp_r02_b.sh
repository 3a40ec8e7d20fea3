#!/bin/bash
# Round-2 experiment B (GPU box): per-phase cycle stamps of the sweep's stage loop.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_b
mkdir -p $OUT
SGP_HIPCC_FLAGS=-DSGP_INSTRUMENT python -m safeopt_amd.build --force > /dev/null || exit 1
for c in 3 2 4; do for a in 0 8 4; do
  echo "== cfg $c ablate $a"; SGP_ABLATE=$a SGP_STAMPS=1 timeout 120 python scripts/ablate.py $c 1 2>&1 | tail -3
done; done 2>&1 | tee $OUT/stamps.txt
python -m safeopt_amd.build --force > /dev/null
