"""Whole optimize() steps at d up to 8 and up to 8 GPs (every instance of the pass kernels), GP and
Lipschitz certificates, against the oracle."""
import importlib.util, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("dev_fuzz", os.path.join(ROOT, "scripts", "dev", "fuzz.py"))
f = importlib.util.module_from_spec(spec); spec.loader.exec_module(f)
t0 = time.time()
for lip in (0.0, 1.0):
    bad, worst = f.run(trials=200, dmax=8, Gmax=8, nmax=300, seed0=990000, verbose=False, products=True,
                       grids=True, lipschitz=lip)
    print("fuzz.run(trials=200, dmax=8, Gmax=8, nmax=300, seed0=990000, products=True, grids=True, lipschitz=%.0f): "
          "%d mismatches, max |Q_dev - Q_oracle| = %.3g  (%.0f s)" % (lip, bad, worst, time.time() - t0), flush=True)
