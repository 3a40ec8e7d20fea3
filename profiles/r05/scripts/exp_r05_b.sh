#!/bin/bash
# round 5, block b: attribution of the paired kernel's time beyond eval / barrier / DMA
# (SGP_INSTRUMENT build; bits: 1 barrier, 2 DMA, 4 evaluation, 8 MFMA, 64 A-operand reads of
# the slots, 128 B-operand reads)
cd "$(dirname "$0")/../../.."
OUT=gpurun_out/exp_r05_b.txt; : > $OUT
export SAFEOPT_HIP_LIB=$PWD/scripts/dev/ab/instr5.so
for c in 3 4; do for a in 0 7 64 128 192 71 135 199 207 15; do
  SGP_ABLATE=$a timeout 120 python scripts/ablate.py $c 2>&1 | tail -1 >> $OUT
done; done
cat $OUT
