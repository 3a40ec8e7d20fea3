#!/bin/bash
# round 5, block k: randomised whole optimize() steps against the oracle on the final source -- the range of
# the resident-factor kernel (49 .. 128 observations; in passes up to 256 on tensor grids with factor tables)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
python - <<'PY' 2>&1 | tee $OUT/fuzz_r05.txt
import importlib.util, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("dev_fuzz", os.path.join(ROOT, "scripts", "dev", "fuzz.py"))
f = importlib.util.module_from_spec(spec); spec.loader.exec_module(f)
t0 = time.time()
print("fuzz.run(trials=700, dmax=4, Gmax=3, nmax=270, seed0=960000, products=True, grids=True)")
f.run(trials=700, dmax=4, Gmax=3, nmax=270, seed0=960000, verbose=True, products=True, grids=True)
print("  (%.0f s)" % (time.time() - t0))
PY
