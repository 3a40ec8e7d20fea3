"""Wall time of SafeOptSwarm.optimize() with the reference's default swarm size
(20 particles, 100 iterations, 3 swarms) -- the small-P regime."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, safeopt_amd, safeopt_amd.gpy as gpy
from safeopt_amd import _hip

ctx = _hip.Context.default()
for n in (50, 200, 1000, 2000):
    cfg = bench.make_config(5)
    cfg["X"], cfg["Y"], cfg["n"] = cfg["X"][:n], cfg["Y"][:n], n
    gps = bench.build_gps(cfg, gpy)
    for pso in ("device", "host"):
        opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * 4, threshold=0.2, pso=pso)
        np.random.seed(0)
        opt.optimize()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            opt.optimize()
        ctx.sync(); dt = (time.perf_counter() - t0) / 3
        parts = np.random.rand(20, 4)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(20):
            opt._compute_particle_fitness("maximizers", parts)
        ctx.sync(); df = (time.perf_counter() - t0) / 20
        print("n=%4d pso=%-6s optimize %.1f ms   one fitness call (P=20) %.3f ms" % (n, pso, dt * 1e3, df * 1e3))
