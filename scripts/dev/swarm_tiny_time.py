import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, safeopt_amd, safeopt_amd.gpy as gpy
from safeopt_amd import _hip
ctx = _hip.Context.default()
for n in (12, 20, 40):
    cfg = bench.make_config(5)
    cfg["X"], cfg["Y"], cfg["n"] = cfg["X"][:n], cfg["Y"][:n], n
    gps = bench.build_gps(cfg, gpy)
    opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * 4, threshold=0.2, pso="device")
    np.random.seed(0)
    opt.optimize()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(5):
        opt.optimize()
    ctx.sync(); dt = (time.perf_counter() - t0) / 5
    print("n=%d optimize %.2f ms (%s sweep)" % (n, dt * 1e3, ctx.last_sweep()))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): opt.optimize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2600])
