#!/usr/bin/env python
"""sweep_mid.hip against the 4-wave kernel and the oracle: n = 50 .. 128, RBF / Matern-5/2,
one and two GPs, on the 1000 x 1000 grid of config 2 (time per launch of both kernels, largest
difference of the intervals, S identical, a slice of rows against the NumPy oracle).

    python scripts/dev/mid_check.py [n ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402
from oracle import gp_numpy as gpn  # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [50, 64, 80, 96, 112, 128]
ctx = _hip.Context.default()
ctx.set_share(False)
cfg = bench.make_config(2)
pts = cfg["grid"]
N = pts.shape[0]
sl = np.arange(0, N, 499)
AXES = _hip.tensor_grid_axes(pts)
for n in ns:
    for kind in ("RBF", "Matern52"):
        for G in (1, 2):
            rng = np.random.default_rng(n + G)
            X = rng.uniform(-2, 2, size=(n, 2))
            Ys = [(bench._bumps(X, 3 + g) - bench._bumps(X, 3 + g).min() + 0.5)[:, None] for g in range(G)]
            gps = [gpy.models.GPRegression(X, Y, getattr(gpy.kern, kind)(2, variance=2.0, lengthscale=[1.0, 0.8], ARD=True),
                                           noise_var=0.05 ** 2) for Y in Ys]
            devs = [g._fitted() for g in gps]
            fmin = np.full(G, 0.6)
            out = {}
            for which in ("classic", "auto") + (("auto+tables",) if kind == "RBF" else ()):
                ctx.set_sweep(which.split("+")[0])
                grid = _hip.DeviceGrid(ctx, pts, G)
                if which.endswith("tables"):
                    grid.set_axes(AXES)
                grid.confidence(devs, 2.0, fmin)
                name = ctx.last_sweep()
                Q = grid.download(_hip.Q).copy()
                S = grid.download(_hip.S).copy()
                for _ in range(20):
                    grid.confidence(devs, 2.0, fmin)
                ctx.sync()
                ctx.profile_enable(True)
                for _ in range(20):
                    grid.confidence(devs, 2.0, fmin)
                ctx.sync()
                ms, cnt, fl = ctx.profile_read()
                ctx.profile_enable(False)
                out[which] = (name, ms / cnt, fl / ms / 1e9 / 78.6, Q, S)
            ctx.set_sweep("auto")
            qo = np.empty((sl.size, 2 * G))
            for g in range(G):
                go = gpn.GPRegression(X, Ys[g], getattr(gpn, kind)(2, 2.0, [1.0, 0.8], ARD=True), noise_var=0.05 ** 2)
                mu, var = go.predict_noiseless(pts[sl])
                sd = np.sqrt(var[:, 0])
                qo[:, 2 * g] = mu[:, 0] - 2.0 * sd
                qo[:, 2 * g + 1] = mu[:, 0] + 2.0 * sd
            a, b = out["classic"], out["auto"]
            if "auto+tables" in out:
                c = out["auto+tables"]
                print("n %3d %-8s G %d | tables: %s %.4f ms (%.3f) | max |dQ| vs evaluated %.1e  S equal %s | vs oracle %.1e"
                      % (n, kind, G, c[0], c[1], c[2], np.max(np.abs(c[3] - b[3])), bool(np.array_equal(c[4], b[4])),
                         np.max(np.abs(c[3][sl] - qo))), flush=True)
            print("n %3d %-8s G %d | %s %.4f ms (%.3f) | %s %.4f ms (%.3f) | x %.2f | max |dQ| %.1e  S equal %s | vs oracle %.1e"
                  % (n, kind, G, a[0], a[1], a[2], b[0], b[1], b[2], a[1] / b[1],
                     np.max(np.abs(a[3] - b[3])), bool(np.array_equal(a[4], b[4])),
                     np.max(np.abs(b[3][sl] - qo))), flush=True)
