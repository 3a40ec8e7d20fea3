#!/bin/bash
# what a separable-RBF path on a tensor grid could gain: covariances = product of D
# prefetched table entries instead of the evaluation (-DSGP_SEP_PROBE; results wrong)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_j
mkdir -p $OUT
for c in 2 4; do echo -n "evaluated : "; timeout 120 python scripts/ablate.py $c 8 2>&1 | tail -1; done | tee $OUT/sep.txt
SGP_HIPCC_FLAGS=-DSGP_SEP_PROBE python -m safeopt_amd.build --force > /dev/null || exit 1
for c in 2 4; do echo -n "table probe: "; timeout 120 python scripts/ablate.py $c 8 2>&1 | tail -1; done | tee -a $OUT/sep.txt
python -m safeopt_amd.build --force > /dev/null
