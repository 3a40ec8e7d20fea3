"""Cost of the multi-rank control flow (2 device round trips + 2 RCCL collectives
per certified step, the scalar all-reduces in stream) measured on ONE GPU: a
one-rank RCCL communicator that claims world = 2 towards the host driver, so
SafeOpt takes the N-rank branches on its half of the grid; the same half run as a
single-rank problem is the reference point."""
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


def timed(opt):
    for _ in range(3):
        x = opt.optimize()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(40):
        x = opt.optimize()
    ctx.sync()
    return (time.perf_counter() - t0) / 40 * 1e3, x


for k in (2, 3):
    cfg = bench.make_config(k)
    G = cfg["G"]
    half = cfg["grid"][:cfg["grid"].shape[0] // 2]
    fmin = cfg["fmin"] if G > 1 else 0.
    gps = bench.build_gps(cfg, gpy)
    t1, _ = timed(safeopt_amd.SafeOpt(gps if G > 1 else gps[0], half, fmin, threshold=cfg["threshold"]))
    print("config %d, %d rows as a single-rank problem: %.3f ms/step" % (k, half.shape[0], t1))
    for in_stream in (False, True):
        gps = bench.build_gps(cfg, gpy)
        opt = safeopt_amd.SafeOpt(gps if G > 1 else gps[0], cfg["grid"], fmin, threshold=cfg["threshold"],
                                  comm=Pretend(comm, 2, in_stream))
        t2, _ = timed(opt)
        print("config %d, the same rows as rank 0 of 2 (in-stream scalars: %s): %.3f ms/step  (+%.3f ms)"
              % (k, in_stream, t2, t2 - t1))
