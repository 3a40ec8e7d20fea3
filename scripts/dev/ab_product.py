#!/usr/bin/env python
"""A/B of the two posterior-sweep kernels on PRODUCT-kernel GPs (the shape of the
reference's context example: a kernel over the parameters times a kernel over the
context), where the 4-wave kernel's instances spill a few registers.

    python scripts/dev/ab_product.py [n ...]        (default 200 256)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip, linearly_spaced_combinations  # noqa: E402


def run(n, d, rows=1000000, reps=5, single=False):
    ctx = _hip.Context.default()
    ctx.set_share(False)
    rng = np.random.default_rng(n + d)
    X = rng.uniform(-2, 2, size=(n, d))
    Y = np.sin(X.sum(axis=1, keepdims=True)) + 0.05 * rng.normal(size=(n, 1))
    # first d - 1 dimensions: Matern-5/2, last one: RBF (a "context")
    k = (gpy.kern.Matern52(d - 1, variance=2.0, lengthscale=1.0, ARD=True,
                           active_dims=list(range(d - 1))) *
         gpy.kern.RBF(1, variance=1.0, lengthscale=1.5, active_dims=[d - 1]))
    if single:
        k = gpy.kern.Matern52(d, variance=2.0, lengthscale=1.5, ARD=True)
    gp = gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
    if d <= 4:
        side = int(round(rows ** (1.0 / d)))
        pts = linearly_spaced_combinations([(-3, 3)] * d, side)
    else:
        pts = rng.uniform(-3, 3, size=(rows, d))
    grid = _hip.DeviceGrid(ctx, pts, 1)
    devs = [gp._fitted()]
    fmin = np.zeros(1)
    out = {}
    for which in ("classic", "pair"):
        ctx.set_sweep(which)
        grid.confidence(devs, 2.0, fmin)
        Q = grid.download(_hip.Q)
        ctx.profile_enable(True)
        for _ in range(reps):
            grid.confidence(devs, 2.0, fmin)
        ctx.sync()
        ms, cnt, fl = ctx.profile_read()
        ctx.profile_enable(False)
        out[which] = (ms / cnt, fl / ms / 1e9, Q)
    ctx.set_sweep("auto")
    a, b = out["classic"], out["pair"]
    print(("single" if single else "product") + " kernel d=%d n=%d rows=%d: classic %.3f ms (%.1f TF) | pair %.3f ms (%.1f TF) | "
          "pair/classic %.3f | max |diff| %.2e" %
          (d, n, len(pts), a[0], a[1], b[0], b[1], b[0] / a[0],
           float(np.max(np.abs(a[2] - b[2])))), flush=True)


if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [200, 256]
    if "--high-d" in sys.argv:
        for n in ns:
            for d in (5, 6, 8):
                run(n, d, single=True)
                run(n, d)
        sys.exit(0)
    for n in ns:
        for d in (2, 3, 4):
            run(n, d)
