"""Rows at which the VALU kernel (sweep_tiny.hip) overtakes the 4-wave kernel for GPs with
few observations: time per launch for N rows, both kernels, n = 8 / 20 / 40."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, safeopt_amd.gpy as gpy
from safeopt_amd import _hip
ctx = _hip.Context.default()
print("%-5s %-9s %10s %10s" % ("n", "rows", "tiny us", "classic us"))
for n in (8, 20, 40):
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, size=(n, 2)); Y = (1.0 + np.exp(-(X ** 2).sum(1)))[:, None]
    gp = gpy.models.GPRegression(X, Y, gpy.kern.Matern52(2, 2., [1., 1.], ARD=True), noise_var=0.05 ** 2)
    dev = gp._fitted()
    for N in (20, 1000, 4000, 16000, 32000, 64000, 128000, 500000):
        pts = rng.uniform(-3, 3, size=(N, 2))
        out = []
        for which in ("auto", "classic"):
            ctx.set_sweep(which)
            grid = _hip.DeviceGrid(ctx, pts, 1)
            for _ in range(30): grid.confidence([dev], 2.0, np.zeros(1))
            ctx.sync(); ctx.profile_enable(True)
            for _ in range(50): grid.confidence([dev], 2.0, np.zeros(1))
            ctx.sync(); ms, cnt, _ = ctx.profile_read(); ctx.profile_enable(False)
            out.append(ms / cnt * 1e3)
        ctx.set_sweep("auto")
        print("%-5d %-9d %10.1f %10.1f" % (n, N, out[0], out[1]), flush=True)
