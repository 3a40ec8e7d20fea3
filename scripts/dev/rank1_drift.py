"""Soak of the incremental path: BO iterations with bordered updates and rank-1
posterior refreshes (full sweep every 16) against a fresh fit + full sweep of the
same data at every iteration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, safeopt_amd, safeopt_amd.gpy as gpy


def run(iters=120, n0=30, config=2, side=200, verbose=True):
    """(same query point every iteration, max |Q_incremental - Q_refit|)."""
    cfg = bench.make_config(config, side=side)
    rng = np.random.default_rng(0)
    X, Y = cfg["X"][:n0], cfg["Y"][:n0, :1]
    kind = cfg["kernels"][0][0]["kind"]

    def make(incremental):
        k = getattr(gpy.kern, kind)(cfg["d"], variance=2., lengthscale=1., ARD=True)
        gp = gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
        gp.incremental = incremental
        o = safeopt_amd.SafeOpt(gp, cfg["grid"], 0., threshold=0.2)
        o._backend.incremental = incremental
        return o

    a, b = make(True), make(False)
    worst_q, same = 0.0, True
    for it in range(iters):
        xa, xb = a.optimize(), b.optimize()
        same &= bool(np.array_equal(xa, xb))
        worst_q = max(worst_q, float(np.max(np.abs(a.Q - b.Q))))
        y = bench._bumps(np.atleast_2d(xa), 101)[0] + 1.0 + 0.05 * rng.normal()
        a.add_new_data_point(xa, y)
        b.add_new_data_point(xa, y)
    if verbose:
        print("%d iterations from n = %d: same query point every iteration: %s; "
              "max |Q_incremental - Q_refit| = %.3g" % (iters, n0, same, worst_q))
    return same, worst_q


if __name__ == "__main__":
    run(*[int(v) for v in sys.argv[1:3]])
