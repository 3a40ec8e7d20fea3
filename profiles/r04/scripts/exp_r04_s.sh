#!/bin/bash
# A-chunk LDS-DMA with the non-temporal policy: do the factor tables stay in L2 then?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04s; mkdir -p $OUT; cd $R
for rep in 1 2; do
  for v in cur dmant; do
    lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
    SAFEOPT_HIP_LIB=$lib AB_ONLY=pair AB_TAG=" [$v]" timeout 300 python scripts/dev/ab_sweep.py 4 3 5 2>&1 | grep "^cfg"
    SAFEOPT_HIP_LIB=$lib AB_ONLY=classic AB_TAG=" [$v]" timeout 300 python scripts/dev/ab_sweep.py 2 2>&1 | grep "^cfg"
  done
done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in cur dmant; do
  lib=$R/scripts/dev/ab/$v.so; [ $v = cur ] && lib=$R/safeopt_amd/libsafeopt_hip.so
  SAFEOPT_HIP_LIB=$lib AB_ONLY=pair rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --output-format csv -d $OUT/pmc_$v -- python $R/scripts/dev/ab_sweep.py 4 > $OUT/pmc_$v.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for v in ("cur","dmant"):
    f=glob.glob("gpurun_out/r04s/pmc_%s/*/*_counter_collection.csv"%v)
    agg=collections.defaultdict(list)
    for x in csv.DictReader(open(f[0])):
        if "k_sweep_pair" in x["Kernel_Name"]: agg[x["Counter_Name"]].append(float(x["Counter_Value"]))
    m={k:sum(a)/len(a) for k,a in agg.items()}
    print(v, "fabric read requests per 1e6-row launch %.4g, %.0f cycles each"%(m["TCC_EA0_RDREQ_sum"], m["TCC_EA0_RDREQ_LEVEL_sum"]/m["TCC_EA0_RDREQ_sum"]))
PY
