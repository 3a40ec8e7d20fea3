#!/usr/bin/env python
"""Timing experiments on the posterior sweep (results are WRONG with any flag
set; this only answers "where does the time go").

    SGP_ABLATE=<mask> python scripts/ablate.py <config> [reps]

mask bits: 1 no stage barrier, 2 no LDS-DMA, 4 no covariance evaluation,
8 no MFMA.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ctx = _hip.Context.default()
    cfg = bench.make_config(k)
    gps = bench.build_gps(cfg, gpy)
    pts = cfg["grid"] if "grid" in cfg else cfg["particles"]
    if k == 4:                       # one rank's share of the 200^3 grid
        pts = pts[4000000:5000000]
    grid = _hip.DeviceGrid(ctx, pts, cfg["G"])
    devs = [g._fitted() for g in gps]
    fmin = np.zeros(cfg["G"])
    grid.confidence(devs, 2.0, fmin)
    ctx.profile_enable(True)
    for _ in range(reps):
        grid.confidence(devs, 2.0, fmin)
    ctx.sync()
    ms, n, fl = ctx.profile_read()
    print("cfg %d ablate %s: sweep %.3f ms  (%.1f TF algorithmic)" %
          (k, os.environ.get("SGP_ABLATE", "0"), ms / n, fl / ms / 1e9))


if __name__ == "__main__":
    main()
