"""The randomised device-vs-oracle runs of round 6's final source (profiles/r06/fuzz.txt):
whole SafeOpt.optimize() steps and _compute_particle_fitness calls on seeded random problems."""
import importlib.util, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
def mod(name):
    spec = importlib.util.spec_from_file_location("dev_" + name, os.path.join(ROOT, "scripts", "dev", name + ".py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
t0 = time.time()
f = mod("fuzz")
print("fuzz.run(trials=1200, dmax=5, Gmax=4, nmax=600, seed0=960000, products=True, grids=True)  -- whole SafeOpt.optimize() steps")
print("  (random n, d, G, kernels, products of two parts, tensor grids with factor tables, thresholds, fmin incl. -inf):")
bad, worst = f.run(trials=1200, dmax=5, Gmax=4, nmax=600, seed0=960000, verbose=False, products=True, grids=True)
print("  1200 trials, %d mismatches, max |Q_dev - Q_oracle| = %.3g  (%.0f s)" % (bad, worst, time.time() - t0)); t0 = time.time()
print("fuzz.run(trials=600, dmax=5, Gmax=4, nmax=400, seed0=980000, products=True, grids=True, lipschitz=1.0)  -- the same with")
print("  Lipschitz certificates in every trial (gp_opt.py:558-576; sgp_grid_lipschitz_pass behind the first candidate):")
bad, worst = f.run(trials=600, dmax=5, Gmax=4, nmax=400, seed0=980000, verbose=False, products=True, grids=True, lipschitz=1.0)
print("  600 trials, %d mismatches, max |Q_dev - Q_oracle| = %.3g  (%.0f s)" % (bad, worst, time.time() - t0)); t0 = time.time()
s = mod("fuzz_swarm")
print("fuzz_swarm.run(trials=400, nmax=700, pmax=8000, seed0=970000, products=True)  -- _compute_particle_fitness, 4 swarm types:")
r = s.run(trials=400, nmax=700, pmax=8000, seed0=970000, verbose=False, products=True)
print("  result %s  (%.0f s)" % (r, time.time() - t0))
