import time, cProfile, pstats, io, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import safeopt_amd as sa, safeopt_amd.gpy as gpy
for N1, n in ((32, 5), (100, 12), (1000, 20)):
    grid = sa.linearly_spaced_combinations([(-5., 5.)] * 2, N1)
    rng = np.random.default_rng(0)
    X = rng.uniform(-2, 2, size=(n, 2)); Y = (1.0 + np.exp(-(X ** 2).sum(1)))[:, None]
    gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(2, 2., [1., 1.], ARD=True), noise_var=0.05 ** 2)
    opt = sa.SafeOpt(gp, grid, 0., threshold=0.2)
    for _ in range(50): opt.optimize()
    t0 = time.perf_counter()
    for _ in range(500): opt.optimize()
    dt = (time.perf_counter() - t0) / 500
    print("grid %dx%d n=%d: %.1f us per optimize()" % (N1, N1, n, dt * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): opt.optimize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
