#!/bin/bash
# round 5, block c: compile-time ablation of the paired kernel (-DPGP_CT_ABL=mask: no run-time
# cost next to what is measured).  bits: 1 barrier, 2 LDS-DMA, 4 evaluation, 8 MFMA,
# 64 A-operand reads of the slots, 128 B-operand reads.  Results are wrong by design.
cd "$(dirname "$0")/../../.."
OUT=gpurun_out/exp_r05_c.txt; : > $OUT
export AB_ONLY=pair
for v in cur a0 a1 a2 a4 a7 a64 a71 a128 a135 a192 a199 a8; do
  lib=$PWD/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$PWD/safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_TAG=$v timeout 200 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg" >> $OUT
done
cat $OUT
