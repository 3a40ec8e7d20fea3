#!/usr/bin/env python
"""hipEvent vs rocprofv3 kernel time of the 4-wave sweep (config 2 shape) at 1x and
Mx the rows: is the difference a fixed cost per dispatch or proportional?

    python scripts/dev/clock_reconcile.py MULT [reps]          (plain, and again under
    rocprofv3 --kernel-trace --stats: compare the printed hipEvent average with the
    k_sweep average of the stats file)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import safeopt_amd.gpy as gpy  # noqa: E402
from safeopt_amd import _hip  # noqa: E402

mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
k = int(os.environ.get("RECONCILE_CFG", "2"))
ctx = _hip.Context.default()
ctx.set_share(False)
cfg = bench.make_config(k, rows_y_mult=mult)
gps = bench.build_gps(cfg, gpy)
devs = [g._fitted() for g in gps]
grid = _hip.DeviceGrid(ctx, cfg["grid"], cfg["G"])
fmin = np.zeros(cfg["G"])
for _ in range(3):
    grid.confidence(devs, 2.0, fmin)
ctx.sync()
ctx.profile_enable(True)
for _ in range(reps):
    grid.confidence(devs, 2.0, fmin)
ctx.sync()
ms, n, fl = ctx.profile_read()
print("cfg %d x%d rows: hipEvent %.4f ms per launch (%d launches, %.3f of 78.6)" %
      (k, mult, ms / n, n, fl / ms / 1e9 / 78.6), flush=True)
