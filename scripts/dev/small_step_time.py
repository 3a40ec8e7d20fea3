"""What a SafeOpt.optimize() of the reference's own problem sizes costs (BASELINE.json
config 1 and neighbours): the one-launch step (sgp_grid_step_small) against the large-grid
path on the same object, plus a profile of the host side."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import safeopt_amd as sa, safeopt_amd.gpy as gpy


def run(d, sides, n, G=1, kind="RBF", steps=2000):
    grid = sa.linearly_spaced_combinations([(-5., 5.)] * d, sides)
    rng = np.random.default_rng(0)
    X = rng.uniform(-2, 2, size=(n, d))
    gps = [gpy.models.GPRegression(X, (1.0 + np.exp(-(X ** 2).sum(1)) + 0.1 * g)[:, None],
                                   getattr(gpy.kern, kind)(d, 2., [1.] * d, ARD=True),
                                   noise_var=0.05 ** 2) for g in range(G)]
    opt = sa.SafeOpt(gps if G > 1 else gps[0], grid, [0.] * G if G > 1 else 0., threshold=0.2)
    out = []
    for small in (True, False):
        opt.small_step = small
        for _ in range(100):
            x = opt.optimize()
        t0 = time.perf_counter()
        for _ in range(steps):
            opt.optimize()
        out.append((time.perf_counter() - t0) / steps * 1e6)
    print("grid %s n=%d G=%d %s: one launch %.1f us | large-grid path %.1f us per optimize()" %
          ("x".join(map(str, sides)), n, G, kind, out[0], out[1]), flush=True)
    opt.small_step = True
    return opt


if __name__ == "__main__":
    opt = run(1, [1000], 20)                 # BASELINE.json config 1
    run(1, [1000], 5)
    run(2, [100, 100], 12)
    run(2, [128, 128], 40, G=2, kind="Matern52")
    run(2, [32, 32], 5)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000):
        opt.optimize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print(s.getvalue()[:3000])
