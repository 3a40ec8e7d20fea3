#!/bin/bash
# where the L2 misses of the sweeps are served: average fabric read latency of the sweep
# kernels against an HBM-streaming and an Infinity-Cache-resident reference kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04r; mkdir -p $OUT; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/probe_mall scripts/dev/probe_mall.hip
/tmp/probe_mall > $OUT/probe_plain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CNT="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/ealat_probe -- /tmp/probe_mall > $OUT/ealat_probe.log 2>&1
SHORT="--steps 3 --warmup 1 --profile-steps 1 --no-extras --no-cpu-baseline --no-check-chosen --no-shared-pass"
for c in 3 4 5 2; do
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/ealat_cfg$c -- python $R/bench.py --config $c $SHORT > $OUT/ealat_cfg$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
out=[]
for name in ("probe","cfg3","cfg4","cfg5","cfg2"):
    f=glob.glob("gpurun_out/r04r/ealat_%s/*/*_counter_collection.csv"%name)
    if not f: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(f[0])):
        k=x["Kernel_Name"]
        if "k_sweep" in k or "k_lat" in k:
            agg[k.split("(")[0][-60:]][x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k,v in agg.items():
        m={n:sum(a)/len(a) for n,a in v.items()}
        lat=m["TCC_EA0_RDREQ_LEVEL_sum"]/max(m["TCC_EA0_RDREQ_sum"],1)
        out.append("%-8s %-58s requests %.4g  latency %.0f cycles  (32B %.3g, 128B %.3g)"%(name,k,m["TCC_EA0_RDREQ_sum"],lat,m["TCC_EA0_RDREQ_32B_sum"],m["TCC_BUBBLE_sum"]))
open("gpurun_out/r04r/served_by.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
cat $OUT/probe_plain.txt
