#!/usr/bin/env python
"""A few launches of the sweep at n observations on the config-2 grid (for rocprofv3):
    python scripts/dev/mid_run.py n [RBF|Matern52] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402

n = int(sys.argv[1])
kind = sys.argv[2] if len(sys.argv) > 2 else "RBF"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ctx = _hip.Context.default()
ctx.set_sweep(os.environ.get("MID_SWEEP", "auto"))
pts = bench.make_config(2)["grid"]
rng = np.random.default_rng(n)
X = rng.uniform(-2, 2, size=(n, 2))
Y = (bench._bumps(X, 3) - bench._bumps(X, 3).min() + 0.5)[:, None]
gp = gpy.models.GPRegression(X, Y, getattr(gpy.kern, kind)(2, variance=2.0, lengthscale=[1.0, 0.8], ARD=True), noise_var=0.05 ** 2)
dev = gp._fitted()
grid = _hip.DeviceGrid(ctx, pts, 1)
for _ in range(reps):
    grid.confidence([dev], 2.0, np.zeros(1))
ctx.sync()
print(ctx.last_sweep())
