#!/bin/bash
# dynamic instruction mix of the paired kernel for library variants (rocprofv3 --pmc, kernel trace only):
#   scripts/dev/pmc_ab.sh "r05 cur two" "pair-unmerged pair" 3
cd "$(dirname "$0")/../.."
R=$(pwd); VARS=$1; MODES=${2:-pair}; CFG=${3:-3}
OUT=$R/gpurun_out/pmc_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  lib=$R/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$R/safeopt_amd/libsafeopt_hip.so
  for m in $MODES; do
    for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS"; do
      tag=$(echo $set | cut -c1-12 | tr ' ' '_')
      d=$OUT/${v}_${m}_$tag; rm -rf $d
      SAFEOPT_HIP_LIB=$lib AB_ONLY=$m timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python $R/scripts/dev/ab_sweep.py $CFG > $d.log 2>&1
      f=$(find $d -name "*counter_collection.csv" | head -1)
      python3 - "$f" "$v/$m" <<'P'
import csv,sys,collections
f,tag=sys.argv[1:3]
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if 'k_sweep_pair' not in r['Kernel_Name']: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print(tag, ' '.join('%s=%.4g'%(k,acc[k]/n[k]) for k in sorted(acc)))
P
    done
  done
done
