"""Time sgp_grid_expander_check (single candidate) for several near_frac values."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, safeopt_amd, safeopt_amd.gpy as gpy
from safeopt_amd import _hip
cfg = bench.make_config(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
gps = bench.build_gps(cfg, gpy)
opt = safeopt_amd.SafeOpt(gps if cfg["G"] > 1 else gps[0], cfg["grid"],
                          cfg["fmin"] if cfg["G"] > 1 else 0., threshold=cfg["threshold"])
x = opt.optimize()
be = opt._backend
idx = int(np.flatnonzero((cfg["grid"] == x).all(axis=1))[0])
xc, mean, var, Q = be.gather_rows(np.array([idx]))
ctx = _hip.Context.default()
for nf in (10.0, 0.9, 0.5, 0.1, 0.0):
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        fl = be.expander_check(2.0, opt.fmin, xc, mean, Q[:, 1::2], nf)
        ctx.sync(); dt = time.perf_counter() - t0
    print("near_frac %.1f: %.1f us  flags %s" % (nf, dt * 1e6, fl.ravel()[:3]))
