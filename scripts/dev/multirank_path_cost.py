"""Cost of the multi-rank control flow (4 device round trips + 4 RCCL
collectives) measured on ONE GPU: a one-rank RCCL communicator that claims
world = 2 towards the host driver, so SafeOpt takes the N-rank branches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SAFEOPT_FORCE_RCCL"] = "1"
import bench, safeopt_amd, safeopt_amd.gpy as gpy
from safeopt_amd import dist

ctx, comm = dist.init_from_env()


class Pretend(object):
    rank = 0

    def __init__(self, c, world, in_stream):
        self.c, self.world, self.in_stream = c, world, in_stream

    def allreduce_max(self, a):
        return self.c.allreduce_max(a)

    def allgather(self, a):
        return self.c.allgather(a)

    def barrier(self):
        self.c.barrier()


cfg = bench.make_config(2)
for world, in_stream in ((1, False), (2, False), (2, True)):
    gps = bench.build_gps(cfg, gpy)
    opt = safeopt_amd.SafeOpt(gps[0], cfg["grid"], 0., threshold=cfg["threshold"],
                              comm=Pretend(comm, world, in_stream))
    for _ in range(3):
        x = opt.optimize()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(20):
        x = opt.optimize()
    ctx.sync(); dt = (time.perf_counter() - t0) / 20
    print("control flow of world=%d (in-stream scalars: %s): %.3f ms/step, x=%s"
          % (world, in_stream, dt * 1e3, x))
