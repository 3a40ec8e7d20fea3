#!/bin/bash
# Round-2: first run of the rewritten sweep: parity tests, then sweep timings.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_d
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
for c in 2 3 4 5; do timeout 300 python scripts/ablate.py $c 4 2>&1 | tail -1; done | tee $OUT/sweep.txt
