#!/bin/bash
# randomised whole optimize() steps on the round-4 paths: few observations (VALU kernel),
# factors resident in LDS, factor tables on tensor grids with contexts, products
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04v; mkdir -p $OUT; cd $R
python - <<'PY' 2>&1 | tee $OUT/fuzz.txt
import importlib.util, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("dev_fuzz", os.path.join(ROOT, "scripts", "dev", "fuzz.py"))
f = importlib.util.module_from_spec(spec); spec.loader.exec_module(f)
t0 = time.time()
print("fuzz.run(trials=1200, dmax=4, Gmax=3, nmax=140, seed0=940000, products=True, grids=True)")
f.run(trials=1200, dmax=4, Gmax=3, nmax=140, seed0=940000, verbose=True, products=True, grids=True)
print("  (%.0f s)" % (time.time() - t0)); t0 = time.time()
print("fuzz.run(trials=300, dmax=3, Gmax=2, nmax=700, seed0=950000, products=True, grids=True)")
f.run(trials=300, dmax=3, Gmax=2, nmax=700, seed0=950000, verbose=True, products=True, grids=True)
print("  (%.0f s)" % (time.time() - t0))
PY
