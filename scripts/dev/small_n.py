#!/usr/bin/env python
"""The posterior sweep in the reference's own regime (n = 1 .. 256 observations: every
notebook and test of the reference runs n <= 20, BASELINE config 2 has n = 200) on the
1000 x 1000 grid of config 2: time per launch, fraction of the fp64 MFMA roof
(algorithmic (n^2 + 2n) N flops) AND of the fp64 VALU roof (covariance evaluation:
~22 instructions per RBF value, ~30 per Matern-5/2 value, n N values; 39.3 T lane-
operations/s = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz) -- below n ~ 60 the sweep is
bound by the evaluation, not by the matrix unit.

    python scripts/dev/small_n.py [n ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [8, 20, 32, 64, 96, 128, 200, 256]
ctx = _hip.Context.default()
ctx.set_share(False)
cfg = bench.make_config(2)
grid_pts = cfg["grid"]
N = grid_pts.shape[0]
axes = _hip.tensor_grid_axes(grid_pts)
print("%-5s %-34s %9s %9s %9s %9s" % ("n", "covariances [sweep kernel]", "ms", "TFLOP/s", "of 78.6", "VALU roof"))
for n in ns:
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, size=(n, 2))
    Y = (bench._bumps(X, 3) - bench._bumps(X, 3).min() + 0.5)[:, None]
    cases = [("RBF", True, "auto"), ("RBF", False, "auto"), ("Matern52", False, "auto")]
    if 48 < n <= 128:    # auto = the resident-factor kernel (sweep_mid.hip); the 4-wave kernel next to it
        cases = cases + [("RBF", True, "classic"), ("RBF", False, "classic"), ("Matern52", False, "classic")]
    if n > 128 and os.environ.get("PAIR_AB"):
        # evaluated covariances at n = 129 .. 256: the 4-wave kernel (auto) against the paired
        # kernel forced (auto takes it from n_pad > 256 on)
        cases = [("RBF", False, "auto"), ("RBF", False, "pair"), ("Matern52", False, "auto"),
                 ("Matern52", False, "pair")]
    if n <= 48:      # auto = the VALU kernel (sweep_tiny.hip); the matrix-core kernel next to it
        cases = [("RBF", False, "auto"), ("Matern52", False, "auto"), ("RBF", True, "classic"),
                 ("RBF", False, "classic"), ("Matern52", False, "classic")]
    for kind, tables, which in cases:
        ctx.set_sweep(which)
        k = getattr(gpy.kern, kind)(2, variance=2.0, lengthscale=[1.0, 1.0], ARD=True)
        gp = gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
        dev = gp._fitted()
        grid = _hip.DeviceGrid(ctx, grid_pts, 1)
        if tables:
            grid.set_axes(axes)
        fmin = np.zeros(1)
        for _ in range(60):
            grid.confidence([dev], 2.0, fmin)
        ctx.sync()
        ctx.profile_enable(True)
        for _ in range(40):
            grid.confidence([dev], 2.0, fmin)
        ctx.sync()
        ms, cnt, fl = ctx.profile_read()
        ctx.profile_enable(False)
        t = ms / cnt
        tf = fl / ms / 1e9
        per = 2 if tables else (22 if kind == "RBF" else 30)
        valu = per * n * N / (t * 1e-3) / 1e12 / 39.3
        # VALU kernel: n evaluations + n (n + 1) / 2 + n FMAs per row
        if ctx.last_sweep() == "tiny":
            valu = ((22 if kind == "RBF" else 30) * n + n * (n + 1) / 2 + n) * N / (t * 1e-3) / 1e12 / 39.3
        print("%-5d %-34s %9.4f %9.2f %9.3f %9.3f" %
              (n, kind + (" factor tables" if tables else " evaluated") + " [" + ctx.last_sweep() + "]",
               t, tf, tf / 78.6, valu), flush=True)
ctx.set_sweep("auto")
