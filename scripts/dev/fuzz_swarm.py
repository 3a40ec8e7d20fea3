"""Randomised device-vs-oracle comparison of SafeOptSwarm._compute_particle_fitness
(all four swarm types) and of the device PSO against the host loop."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import safeopt_amd, safeopt_amd.gpy as gpy
from oracle import gp_numpy as gpn, safeopt_numpy as son

KINDS = ["RBF", "Matern32", "Matern52"]


def run(trials=100, nmax=400, pmax=2000, seed0=5000, verbose=True, products=False):
  """(mismatches, max relative fitness error) over seeded random swarms."""
  bad, worst = 0, 0.0
  for t in range(trials):
    rng = np.random.default_rng(seed0 + t)
    n, d, G = int(rng.integers(1, nmax)), int(rng.integers(1, 6)), int(rng.integers(1, 4))
    P = int(rng.integers(1, pmax))
    X = rng.uniform(-2, 2, size=(n, d))
    gps, gos = [], []
    for g in range(G):
        kind = KINDS[int(rng.integers(0, 3))]
        ls = list(rng.uniform(0.5, 2.0, size=d))
        var = float(rng.uniform(0.5, 3.0))
        y = (np.sin(X.sum(1) + g) + 0.5 + 0.3 * rng.normal(size=n))[:, None]
        noise = float(rng.uniform(0.02, 0.2)) ** 2
        def kern(ns):
            return getattr(ns, kind)(d, variance=var, lengthscale=ls, ARD=True)
        if products and d >= 2 and rng.random() < 0.35:
            # a product of two parts on random (possibly overlapping) column sets that
            # together cover every column
            c0 = np.sort(rng.choice(d, size=int(rng.integers(1, d + 1)), replace=False))
            rest = np.setdiff1d(np.arange(d), c0)
            extra = rng.choice(d, size=int(rng.integers(0, 2)), replace=False)
            c1 = np.unique(np.concatenate([rest, extra])).astype(int)
            if c1.size == 0:
                c1 = np.array([int(rng.integers(0, d))])
            cols = [c0, c1]
            kinds2 = [KINDS[int(rng.integers(0, 3))] for _ in range(2)]
            ls2 = [rng.uniform(0.5, 2.0, size=len(c)) for c in cols]

            def kern(ns):
                parts = [getattr(ns, kk)(len(c), variance=var ** 0.5, lengthscale=l, ARD=True,
                                         active_dims=list(c))
                         for kk, c, l in zip(kinds2, cols, ls2)]
                return parts[0] * parts[1]
        gps.append(gpy.models.GPRegression(X, y, kern(gpy.kern), noise_var=noise))
        gos.append(gpn.GPRegression(X, y, kern(gpn), noise_var=noise))
    fmin = [float(rng.uniform(-0.5, 0.8)) if (g == 0 or rng.random() < 0.7) else -np.inf for g in range(G)]
    opt = safeopt_amd.SafeOptSwarm(gps if G > 1 else gps[0], fmin, bounds=[(-3., 3.)] * d,
                                   threshold=0.1, pso="host")
    opt.best_lower_bound = float(rng.uniform(-0.5, 0.5))
    parts = rng.uniform(-3, 3, size=(P, d))
    for st in ["greedy", "maximizers", "expanders", "safe_set"]:
        v, s = opt._compute_particle_fitness(st, parts)
        vo, so = son.swarm_fitness(gos, parts, st, 2., fmin, opt.scaling, opt.best_lower_bound)
        err = float(np.max(np.abs(v - vo) / (1.0 + np.abs(vo))))
        worst = max(worst, err)
        if not (np.array_equal(s, so) and err < 1e-5):
            print("trial %d %s n=%d d=%d G=%d P=%d: err %.2g safe equal %s  MISMATCH"
                  % (t, st, n, d, G, P, err, np.array_equal(s, so)))
            bad += 1
  if verbose:
    print("%d trials x 4 swarm types, %d mismatches, max relative fitness error %.3g" % (trials, bad, worst))
  return bad, worst


if __name__ == "__main__":
    run(*[int(v) for v in sys.argv[1:3]])

