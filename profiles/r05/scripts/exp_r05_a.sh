#!/bin/bash
# round 5, block a: progress flags instead of the stage barrier in the paired kernel
# (variants built with -DPGP_FLAGS / -DPGP_PAIRSYNC from the working tree)
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
OUT=gpurun_out/exp_r05_a.txt
: > $OUT
export AB_ONLY=pair
echo "== correctness of f1 (flags) on the pair-kernel tests" >> $OUT
SAFEOPT_HIP_LIB=$PWD/scripts/dev/ab/f1.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "both_kernels or split_remainder or shared_factor or tensor_grid or full_size_configs_3 or config5" 2>&1 | tail -5 >> $OUT
for rep in 1 2; do
for v in cur f0 f1 f1p f0p; do
  lib=$PWD/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$PWD/safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg" >> $OUT
done; done
echo "== shared factor (riders), config 3" >> $OUT
for v in cur f1; do
  lib=$PWD/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$PWD/safeopt_amd/libsafeopt_hip.so
  AB_SHARE=1 SAFEOPT_HIP_LIB=$lib AB_TAG=$v-share timeout 200 python scripts/dev/ab_sweep.py 3 5 2>&1 | grep "^cfg" >> $OUT
done
cat $OUT
