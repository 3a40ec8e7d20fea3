#!/bin/bash
# round 5, block d: covariance cache of the paired kernel (evaluate a block once per tile; later
# chunks / GPs with the same inputs and kernel read it back).  head = HEAD of round 4.
cd "$(dirname "$0")/../../.."
OUT=gpurun_out/exp_r05_d.txt; : > $OUT
export AB_ONLY=pair
echo "== pair-kernel tests on the new library" >> $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "both_kernels or split_remainder or shared_factor or tensor_grid or full_size_configs_3 or config5 or predict_noiseless" 2>&1 | tail -5 >> $OUT
for rep in 1 2; do
  SAFEOPT_HIP_LIB=$PWD/scripts/dev/ab/head.so AB_TAG=head timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg" >> $OUT
  AB_TAG=kc timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg" >> $OUT
  SGP_COV_CACHE=0 AB_TAG=kc-off timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg" >> $OUT
  AB_SEP=0 AB_TAG=kc-nosep timeout 200 python scripts/dev/ab_sweep.py 4 2>&1 | grep "^cfg" >> $OUT
  AB_SEP=0 SGP_COV_CACHE=0 AB_TAG=kc-off-nosep timeout 200 python scripts/dev/ab_sweep.py 4 2>&1 | grep "^cfg" >> $OUT
done
cat $OUT
