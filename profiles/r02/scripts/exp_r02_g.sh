#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_g
mkdir -p $OUT
for c in 3 4 5; do for sl in 16 32; do
  echo -n "slots $sl: "; SGP_SWEEP_SLOTS=$sl timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
done; done | tee $OUT/slots.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest.txt
