import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import safeopt_amd.gpy as gpy
from oracle import gp_numpy as gpn
from safeopt_amd import _hip
from _gpu_common import smooth
def kern(ns, d, spec):
    k = None
    for i, (kind, cols) in enumerate(spec):
        part = getattr(ns, kind)(len(cols), variance=1.0 + 0.3 * i, lengthscale=list(np.linspace(0.9, 1.4, len(cols))), ARD=True, active_dims=cols)
        k = part if k is None else k * part
    return k
for d in (5, 6, 7, 8):
  for n in (16, 70, 200):
    spec = [("RBF", list(range(d - 1))), ("Matern32", [d - 1])]
    rng = np.random.default_rng(31 * d + n)
    X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 5) + 0.3
    gp = gpy.models.GPRegression(X, Y, kern(gpy.kern, d, spec), noise_var=0.05 ** 2)
    go = gpn.GPRegression(X, Y, kern(gpn, d, spec), noise_var=0.05 ** 2)
    pts = rng.uniform(-3, 3, size=(256, d))
    ctx = gp._fitted().ctx
    old = ctx.set_sweep("classic")
    res = []
    for rep in range(2):
        grid = _hip.DeviceGrid(ctx, pts, 1)
        grid.confidence([gp._fitted()], 2.0, np.zeros(1))
        var = grid.download(_hip.VAR)[0]
        mo, vo = go.predict_noiseless(pts)
        ev = np.abs(var - vo[:, 0])
        res.append((float(ev.max()), np.flatnonzero(ev > 1e-8)))
    ctx.set_sweep(old)
    print("d", d, "n", n, "err %.2e / %.2e" % (res[0][0], res[1][0]), "bad", len(res[0][1]), len(res[1][1]),
          "same" if np.array_equal(res[0][1], res[1][1]) else "DIFFERENT", res[0][1][:16])
