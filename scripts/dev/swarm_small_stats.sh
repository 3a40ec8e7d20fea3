#!/bin/bash
# per-kernel time of SafeOptSwarm.optimize() with the default swarm (20 particles) at n = 2000
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/scripts/dev/swarm_small.py 2>&1 | tail -8
cat > /tmp/one.py <<PY
import os, sys, numpy as np
sys.path.insert(0, "$R")
import bench, safeopt_amd, safeopt_amd.gpy as gpy
cfg = bench.make_config(5)
gps = bench.build_gps(cfg, gpy)
opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * 4, threshold=0.2, pso="device")
np.random.seed(0)
for _ in range(3):
    opt.optimize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw -o sw -- python /tmp/one.py > $R/gpurun_out/sw.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/sw/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-50s %6s %9.2f %6s%%"%(r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
