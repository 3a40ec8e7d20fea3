"""Host-buffer boundary at config 3: what the drop-in pays when it DOES move the
big arrays (it normally does not: the grid is uploaded once per SafeOpt object, Q is
read back only when the user touches opt.Q)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, safeopt_amd, safeopt_amd.gpy as gpy
from safeopt_amd import _hip

ctx = _hip.Context.default()
cfg = bench.make_config(3)
gps = bench.build_gps(cfg, gpy)
t0 = time.perf_counter()
opt = safeopt_amd.SafeOpt(gps, cfg["grid"], cfg["fmin"], threshold=cfg["threshold"])
ctx.sync(); t_up = time.perf_counter() - t0
for _ in range(3):
    opt.optimize()
ctx.sync()
ts, tq = [], []
for _ in range(10):
    t0 = time.perf_counter(); opt.optimize(); ctx.sync(); t1 = time.perf_counter()
    q = opt.Q; t2 = time.perf_counter()
    ts.append(t1 - t0); tq.append(t2 - t1)
N = cfg["grid"].shape[0]
print("SafeOpt construction incl. grid upload (%d x %d doubles): %.2f ms" % (N, cfg["d"], t_up * 1e3))
print("optimize(): %.3f ms; reading opt.Q back (%.0f MB): %.3f ms -> %.3g candidates/s with the read-back in every step"
      % (np.median(ts) * 1e3, q.nbytes / 1e6, np.median(tq) * 1e3, N / (np.median(ts) + np.median(tq))))
