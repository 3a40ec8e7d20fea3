import importlib.util, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
def mod(name):
    spec = importlib.util.spec_from_file_location("dev_" + name, os.path.join(ROOT, "scripts", "dev", name + ".py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
t0 = time.time()
f = mod("fuzz")
print("fuzz.run(trials=1500, dmax=5, Gmax=4, nmax=600, seed0=910000, products=True)  -- whole SafeOpt.optimize() steps, both sweep kernels,")
print("  products of two parts (overlapping column sets) in 30 % of the GPs of d >= 2:")
f.run(trials=1500, dmax=5, Gmax=4, nmax=600, seed0=910000, verbose=True, products=True)
print("  (%.0f s)" % (time.time() - t0)); t0 = time.time()
s = mod("fuzz_swarm")
print("fuzz_swarm.run(trials=500, nmax=700, pmax=8000, seed0=920000, products=True)  -- _compute_particle_fitness, 4 swarm types:")
s.run(trials=500, nmax=700, pmax=8000, seed0=920000, verbose=True, products=True)
print("  (%.0f s)" % (time.time() - t0))
