#!/usr/bin/env python
"""A/B of the two posterior-sweep kernels on the BASELINE.json configs (one
process, same box): k_sweep (4 waves, csrc/sweep.hip) vs k_sweep_pair
(csrc/sweep_pair.hip).  Time per launch from the library's hipEvent pairs.

    python scripts/dev/ab_sweep.py [configs ...]        (default 3 2 4 5)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402


def run(k, reps):
    ctx = _hip.Context.default()
    ctx.set_share(os.environ.get("AB_SHARE", "0") == "1")   # headline: every GP on its own
    cfg = bench.make_config(k)
    gps = bench.build_gps(cfg, gpy)
    G = int(os.environ.get("AB_G", cfg["G"]))     # AB_G=1: the first GP only
    devs = [g._fitted() for g in gps[:G]]
    fmin = np.zeros(G)
    out = {}
    if k == 5:
        parts = cfg["particles"]
        scaling = np.ones(G)
        for which in ((ONLY,) if ONLY else ("classic", "pair")):
            ctx.set_sweep(which)
            res = [_hip.swarm_fitness(ctx, devs, "maximizers", parts, 2.0, fmin, scaling, 0.4)]
            ctx.profile_enable(True)
            for _ in range(reps):
                _hip.swarm_fitness(ctx, devs, "maximizers", parts, 2.0, fmin, scaling, 0.4)
            ctx.sync()
            ms, n, fl = ctx.profile_read()
            ctx.profile_enable(False)
            out[which] = (ms / n, fl / ms / 1e9, res[0][0])
    else:
        pts = cfg["grid"]
        if k == 4:
            pts = pts[4000000:5000000]       # one rank's share of the 200^3 grid
        grid = _hip.DeviceGrid(ctx, pts, G)
        if os.environ.get("AB_SEP", "1") == "1":        # tensor grid: factor tables (RBF)
            lo = 4000000 if k == 4 else 0
            grid = _hip.DeviceGrid(ctx, pts, G, lo)
            grid.set_axes(_hip.tensor_grid_axes(cfg["grid"]))
        for which in ((ONLY,) if ONLY else ("classic", "pair")):
            ctx.set_sweep(which)
            grid.confidence(devs, 2.0, fmin)
            Q = grid.download(_hip.Q)
            # (the clocks ramp up over the first ~30 ms of load: profiles/r04/clock_ramp.txt)
            for _ in range(WARM.get(k, 3)):
                grid.confidence(devs, 2.0, fmin)
            ctx.sync()
            ctx.profile_enable(True)
            for _ in range(reps):
                grid.confidence(devs, 2.0, fmin)
            ctx.sync()
            ms, n, fl = ctx.profile_read()
            ctx.profile_enable(False)
            out[which] = (ms / n, fl / ms / 1e9, Q)
    ctx.set_sweep("auto")
    if ONLY:
        b = out[ONLY]
        import hashlib
        h = hashlib.sha1(np.ascontiguousarray(b[2]).tobytes()).hexdigest()[:10]
        kc = " kc" if hasattr(ctx, "cov_cache_used") and ctx.cov_cache_used() else ""
        print("cfg %d %s: %.3f ms (%.1f TF, %.3f of 78.6) bits %s%s%s" % (
            k, ONLY, b[0], b[1], b[1] / 78.6, h, kc, TAG), flush=True)
        return
    a, b = out["classic"], out["pair"]
    diff = float(np.max(np.abs(a[2] - b[2])))
    print("cfg %d: classic %.3f ms (%.1f TF, %.3f of 78.6) | pair %.3f ms (%.1f TF, %.3f of 78.6) | "
          "pair/classic %.3f | max |diff| %.2e" %
          (k, a[0], a[1], a[1] / 78.6, b[0], b[1], b[1] / 78.6, b[0] / a[0], diff), flush=True)


WARM = {2: 40, 3: 6, 4: 4}                # launches before the timed ones
ONLY = os.environ.get("AB_ONLY")          # "pair" | "classic": time one kernel only
TAG = "  [%s]" % os.environ["AB_TAG"] if os.environ.get("AB_TAG") else ""

if __name__ == "__main__":
    ks = [int(a) for a in sys.argv[1:]] or [3, 2, 4, 5]
    for k in ks:
        run(k, 3 if k == 4 else (20 if k == 2 else 5))
