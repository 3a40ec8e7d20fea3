#!/usr/bin/env python
"""cProfile of SafeOpt.optimize() on a tiny grid: where the HOST time goes."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safeopt_amd, safeopt_amd.gpy as gpy
from bench import make_config, build_gps
cfg = make_config(2, side=64)
gps = build_gps(cfg, gpy)
opt = safeopt_amd.SafeOpt(gps[0], cfg["grid"], 0., threshold=0.2)
for _ in range(20): opt.optimize()
t0 = time.perf_counter()
for _ in range(500): opt.optimize()
print("us per optimize (tiny grid):", (time.perf_counter() - t0) / 500 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): opt.optimize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
