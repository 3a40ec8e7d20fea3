#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_h
mkdir -p $OUT
SGP_HIPCC_FLAGS=-DSGP_SWEEP_WAVES8 python -m safeopt_amd.build --force > /dev/null || exit 1
for c in 3 2 4; do
  echo -n "4 waves x2: "; timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
  echo -n "8 waves   : "; SGP_SWEEP_WAVES=8 timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
  echo -n "ping-pong : "; SGP_SWEEP_WAVES=8 SGP_SWEEP_PP=1 timeout 120 python scripts/ablate.py $c 4 2>&1 | tail -1
done | tee $OUT/pp.txt
SGP_SWEEP_WAVES=8 SGP_SWEEP_PP=1 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_pp.txt
python -m safeopt_amd.build --force > /dev/null
