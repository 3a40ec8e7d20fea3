import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, safeopt_amd.gpy as gpy
from safeopt_amd import _hip
ctx = _hip.Context.default()
for n in (50, 200, 256):
    cfg = bench.make_config(3)
    cfg["X"], cfg["Y"] = cfg["X"][:n], cfg["Y"][:n]
    gps = bench.build_gps(cfg, gpy)
    devs = [g._fitted() for g in gps]
    grid = _hip.DeviceGrid(ctx, cfg["grid"], 3)
    out = {}
    for share in (False, True):
        ctx.set_share(share)
        grid.confidence(devs, 2.0, np.zeros(3))
        Q = grid.download(_hip.Q)
        ctx.profile_enable(True)
        for _ in range(10):
            grid.confidence(devs, 2.0, np.zeros(3))
        ctx.sync()
        ms, cnt, fl = ctx.profile_read()
        ctx.profile_enable(False)
        out[share] = (ms / cnt, Q)
    print("config-3 shape at n=%d: every GP on its own %.3f ms | riders %.3f ms | x%.2f | same bits %s"
          % (n, out[False][0], out[True][0], out[False][0] / out[True][0], np.array_equal(out[False][1], out[True][1])))
