"""Under SGP_POISON=1: which few-point predictions / sweeps come out wrong (reads of memory nobody wrote)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import safeopt_amd, safeopt_amd.gpy as gpy
from oracle import gp_numpy as gpn
from bench import make_config, build_gps
cfg0 = make_config(5)
for n in [int(a) for a in sys.argv[1:]] or (2000, 500):
    cfg = dict(cfg0); cfg["X"], cfg["Y"], cfg["n"] = cfg0["X"][:n], cfg0["Y"][:n], n
    gps, gos = build_gps(cfg, gpy), build_gps(cfg, gpn)
    for g in range(2):
        f = gps[g]._fitted()
        Li, al = f.factor()
        print("n %d GP %d: NaN in L^-1 lower %d upper %d, alpha %d" % (
            n, g, np.isnan(np.tril(Li)).sum(), np.isnan(np.triu(Li, 1)).sum(), np.isnan(al).sum()))
    for P in (64, 17, 1):
        parts = np.random.default_rng(n + P).uniform(-3, 3, size=(P, 4))
        res = []
        for g in range(2):
            m, v = gps[g].predict_noiseless(parts)
            mo, vo = gos[g].predict_noiseless(parts)
            res.append((int((np.abs(m - mo) > 1e-8).sum() + np.isnan(m).sum()), int((np.abs(v - vo) > 1e-8).sum() + np.isnan(v).sum())))
        print("n %d P %d: wrong (mean, var) per GP %s" % (n, P, res), flush=True)
