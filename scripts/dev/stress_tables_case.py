"""Repeat one case of tests/test_gpu_posterior.py::test_tensor_grid_tables_match_generic many times
and compare every result with the first of its kind: any difference between two runs of the same
launch is a race.   python scripts/dev/stress_tables_case.py [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import safeopt_amd as sa, safeopt_amd.gpy as gpy
from safeopt_amd import _hip
from _gpu_common import smooth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CASES = [([37, 29], 0, [17, 255], (100, 1000)), ([23, 19, 7], 0, [600, 257], (50, 3000)),
         ([5, 4, 6, 3], 0, [60, 60], (7, 355)), ([37, 29], 0, [200], None)]
for sides, nc, ns, shard in CASES:
    dp = len(sides); d = dp + nc
    rng = np.random.default_rng(sum(ns) + 7 * d)
    full = sa.linearly_spaced_combinations([(-3., 3.)] * dp, sides)
    axes = _hip.tensor_grid_axes(full)
    lo, hi = shard or (0, full.shape[0])
    gps = []
    for i, n in enumerate(ns):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 3 + i) + 0.3
        k = gpy.kern.RBF(d, 1.7, list(rng.uniform(0.6, 1.5, size=d)), ARD=True)
        gps.append(gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2))
    devs = [g._fitted() for g in gps]
    ctx = devs[0].ctx
    G = len(ns)
    fmin = np.full(G, 0.1)
    grid = _hip.DeviceGrid(ctx, full[lo:hi], G, lo)
    assert grid.set_axes(axes)
    for which in (8, 0):
        old = ctx.set_sweep(which)
        first, bad, worst = None, 0, 0.0
        for r in range(reps):
            grid.confidence(devs, 2.0, fmin)
            out = [grid.download(a) for a in (_hip.MEAN, _hip.VAR)]
            if first is None:
                first = out
            else:
                dm = max(np.abs(out[0] - first[0]).max(), np.abs(out[1] - first[1]).max())
                if dm != 0.0:
                    bad += 1; worst = max(worst, dm)
        ctx.set_sweep(old)
        print("sides %s ns %s shard %s sweep %d (%s): %d of %d runs differ from the first, max %.3g"
              % (sides, ns, shard, which, ctx.last_sweep() if hasattr(ctx, "last_sweep") else "?", bad, reps - 1, worst), flush=True)
