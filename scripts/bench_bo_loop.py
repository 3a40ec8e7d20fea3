#!/usr/bin/env python
"""Per-iteration cost of the sequential BO loop (add_new_data_point + optimize)
with and without the incremental path (bordered factor update + closed-form
rank-1 update of the resident posterior).

    python scripts/bench_bo_loop.py [--config 2|3] [--iters 12]

The headline metric of bench.py is NOT affected: it always times the full
O(n^2 N) sweep.  This reports what a user of the drop-in sees per iteration.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safeopt_amd, safeopt_amd.gpy as gpy
from safeopt_amd import _hip
from bench import make_config, build_gps, _bumps


def run(cfg, incremental, iters):
    gps = build_gps(cfg, gpy)
    for g in gps:
        g.incremental = incremental
    G = cfg["G"]
    opt = safeopt_amd.SafeOpt(gps if G > 1 else gps[0], cfg["grid"],
                              cfg["fmin"] if G > 1 else 0., threshold=cfg["threshold"])
    opt._backend.incremental = incremental
    opt._backend.refresh_every = 1 << 30
    ctx = _hip.Context.default()
    xs, t_add, t_opt = [], [], []
    x = opt.optimize()
    for it in range(iters):
        y = np.array([[_bumps(np.atleast_2d(x), 100 + cfg["k"] - 1 + g)[0] + 1.0 for g in range(G)]])
        ctx.sync(); t0 = time.perf_counter()
        opt.add_new_data_point(x, y)
        ctx.sync(); t1 = time.perf_counter()
        x = opt.optimize()
        ctx.sync(); t2 = time.perf_counter()
        t_add.append(t1 - t0); t_opt.append(t2 - t1); xs.append(np.array(x))
    return np.array(xs), np.median(t_add[2:]) * 1e3, np.median(t_opt[2:]) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    cfg = make_config(a.config)
    xf, addf, optf = run(cfg, False, a.iters)
    xi, addi, opti = run(cfg, True, a.iters)
    print(json.dumps({
        "workload": "config%d BO loop, %d iterations from n=%d" % (a.config, a.iters, cfg["n"]),
        "full_refit_ms": {"add_new_data_point": addf, "optimize": optf},
        "incremental_ms": {"add_new_data_point": addi, "optimize": opti},
        "speedup_per_iteration": (addf + optf) / (addi + opti),
        "same_query_points": bool(np.array_equal(xf, xi)),
    }))


if __name__ == "__main__":
    main()
