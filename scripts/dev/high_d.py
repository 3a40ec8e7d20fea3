#!/usr/bin/env python
"""The posterior sweep at input dimensions 5 .. 8 (SafeOptSwarm's regime,
safeopt/__init__.py:8-10): P = 1e5 random points, n = 200 (4-wave kernel) and n = 1000
(paired kernel), RBF.  Time per launch and fraction of the fp64 MFMA roof.

    python scripts/dev/high_d.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402

ctx = _hip.Context.default()
ctx.set_share(False)
P = 100000
print("%-3s %-6s %-10s %9s %9s %9s" % ("d", "n", "kernel", "ms", "TFLOP/s", "of 78.6"))
for d in (4, 5, 6, 8):
    for n in (200, 1000):
        rng = np.random.default_rng(10 * d + n)
        X = rng.uniform(-3, 3, size=(n, d))
        f = bench._bumps(X, 7)
        Y = (f - f.min() + 0.5)[:, None]
        for kind in ("RBF", "Matern52"):
            k = getattr(gpy.kern, kind)(d, variance=2.0, lengthscale=[1.0] * d, ARD=True)
            gp = gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
            dev = gp._fitted()
            pts = rng.uniform(-5, 5, size=(P, d))
            grid = _hip.DeviceGrid(ctx, pts, 1)
            fmin = np.zeros(1)
            for _ in range(20 if n == 200 else 8):
                grid.confidence([dev], 2.0, fmin)
            ctx.sync()
            ctx.profile_enable(True)
            for _ in range(20 if n == 200 else 8):
                grid.confidence([dev], 2.0, fmin)
            ctx.sync()
            ms, cnt, fl = ctx.profile_read()
            ctx.profile_enable(False)
            tf = fl / ms / 1e9
            print("%-3d %-6d %-10s %9.4f %9.2f %9.3f" % (d, n, kind, ms / cnt, tf, tf / 78.6),
                  flush=True)
