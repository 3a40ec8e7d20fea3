#!/usr/bin/env python
"""Benchmark of the GP-posterior + safe-set sweep (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

A "step" is one ``SafeOpt.optimize()`` = ``update_confidence_intervals`` +
``compute_sets`` + ``get_new_query_point`` over the whole resident candidate
grid (config 5: one ``SafeOptSwarm._compute_particle_fitness`` call for every
swarm type's GP set).  Inputs are synthetic and already
resident in HBM when the timed region starts.  For N > 1 the driver launches
one process per GPU (torchrun env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*);
each rank holds 1e6 rows (weak scaling), the scalar reductions go over RCCL.

Rank 0 prints ONE JSON line.  No torch anywhere: device memory, streams,
events and RCCL all come from libsafeopt_hip.so.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor figure (fp64 matrix = vector)
HBM_PEAK_GBS = 8000.0


def _bumps(x, seed):
    """Smooth synthetic objective: sum of RBF bumps (deterministic)."""
    rng = np.random.default_rng(seed)
    d = x.shape[1]
    c = rng.uniform(-3, 3, size=(16, d))
    w = rng.normal(size=16)
    out = np.zeros(x.shape[0])
    for ci, wi in zip(c, w):
        out += wi * np.exp(-0.5 * ((x - ci) ** 2).sum(1) / 1.5)
    return out


def make_config(k, side=None, rows_y_mult=1):
    """Synthetic inputs of BASELINE.json configs[k-1].

    Returns kernels (fixture-style specs), X, Y (n, G), grid (F-ordered, as
    linearly_spaced_combinations gives it), noise_var, threshold, fmin.
    """
    from safeopt_amd import linearly_spaced_combinations
    spec = {
        1: dict(d=1, kind="RBF", G=1, n=20, side=1000, box=10., xr=4.0, seed=0),
        2: dict(d=2, kind="RBF", G=1, n=200, side=1000, box=5., xr=2.0, seed=1),
        3: dict(d=2, kind="Matern52", G=3, n=500, side=1000, box=5., xr=2.0, seed=2),
        4: dict(d=3, kind="RBF", G=1, n=1000, side=100, box=5., xr=2.5, seed=5),
        5: dict(d=4, kind="RBF", G=2, n=2000, side=None, box=5., xr=3.0, seed=6),
    }[k]
    d, G, n = spec["d"], spec["G"], spec["n"]
    rng = np.random.default_rng(spec["seed"])
    X = rng.uniform(-spec["xr"], spec["xr"], size=(n, d))
    Y = np.empty((n, G))
    for g in range(G):
        f = _bumps(X, 100 + spec["seed"] + g)
        Y[:, g] = f - f.min() + 0.5          # all observations >= 0.5: S != {}
    Y += 0.05 * rng.normal(size=Y.shape)
    kern = [[dict(kind=spec["kind"], variance=2.0, lengthscale=[1.0] * d,
                  ARD=True, input_dim=d, active_dims=list(range(d)))]
            for _ in range(G)]
    cfg = dict(k=k, d=d, G=G, n=n, kernels=kern, X=X, Y=Y, noise_var=0.05 ** 2,
               threshold=0.2, fmin=[0.0] * G, beta=2.0, box=spec["box"])
    if spec["side"] is not None:
        s = side or spec["side"]
        sides = [s] * d
        sides[-1] = s * rows_y_mult          # weak scaling: more rows, same box
        cfg["grid"] = linearly_spaced_combinations(
            [(-spec["box"], spec["box"])] * d, sides)
        cfg["sides"] = sides
    else:
        P = side or 100000
        cfg["particles"] = np.random.default_rng(7).uniform(
            -spec["box"], spec["box"], size=(P, d))
    return cfg


def _kernels(cfg, ns):
    out = []
    for spec in cfg["kernels"]:
        p = spec[0]
        out.append(getattr(ns, p["kind"])(p["input_dim"], variance=p["variance"],
                                          lengthscale=p["lengthscale"], ARD=True))
    return out


def build_gps(cfg, ns, **kw):
    ks = _kernels(cfg, ns)
    return [ns.GPRegression(cfg["X"], cfg["Y"][:, [g]], ks[g],
                            noise_var=cfg["noise_var"], **kw)
            for g in range(cfg["G"])]


CPU_BASELINE_SECONDS = 12.0      # target CPU work of the default sample


def cpu_baseline(cfg, sample_rows, dev_Q=None):
    """The oracle (NumPy restatement of the reference path, all host cores via
    the BLAS thread pool) on a bounded sample of the same workload.
    ``sample_rows=None``: as many rows (a centred block of the grid) as a 20k-row
    pilot predicts for ~CPU_BASELINE_SECONDS of work."""
    from oracle import gp_numpy as gpn
    from oracle import safeopt_numpy as son
    gps = build_gps(cfg, gpn)
    if sample_rows is None:
        N = cfg["grid"].shape[0]
        pilot = np.ascontiguousarray(cfg["grid"][(N - 20000) // 2:(N + 20000) // 2])
        son.confidence_intervals(gps, pilot[:8192], cfg["beta"])
        t0 = time.perf_counter()
        son.confidence_intervals(gps, pilot, cfg["beta"])
        rate = pilot.shape[0] / (time.perf_counter() - t0)
        sample_rows = int(min(N, max(20000, rate * CPU_BASELINE_SECONDS)))
    # a contiguous block of rows from the middle of the grid (the block around
    # the training data, so the sample contains safe, maximiser and unsafe rows)
    N = cfg["grid"].shape[0]
    start = max(0, (N - sample_rows) // 2)
    grid = np.ascontiguousarray(cfg["grid"][start:start + sample_rows])
    scaling = np.sqrt([2.0] * cfg["G"])
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    son.confidence_intervals(gps, grid[:8192], cfg["beta"])    # warm-up
    t0 = time.perf_counter()
    try:
        idx, Q, S, M, G = son.optimize_grid(gps, grid, cfg["fmin"], scaling,
                                            cfg["threshold"], cfg["beta"])
    except EnvironmentError:          # sample without a safe row: sweep only
        Q = son.confidence_intervals(gps, grid, cfg["beta"])
    dt = time.perf_counter() - t0
    out = dict(value=sample_rows / dt, unit="candidates/s", cores=int(cores),
               kind="port",
               sample="oracle optimize_grid (NumPy/OpenBLAS restatement of "
                      "gp_opt.py:453-649 + GPy predict) on rows [%d, %d) of "
                      "the same grid, %.1f s" % (start, start + sample_rows, dt))
    parity = None
    if dev_Q is not None:
        dq = dev_Q[start:start + sample_rows]
        lo, up = Q[:, ::2], Q[:, 1::2]
        mean_o, mean_d = 0.5 * (lo + up), 0.5 * (dq[:, ::2] + dq[:, 1::2])
        var_o = ((up - lo) / (2 * cfg["beta"])) ** 2
        var_d = ((dq[:, 1::2] - dq[:, ::2]) / (2 * cfg["beta"])) ** 2
        parity = dict(
            mean_linf_rel=float(np.max(np.abs(mean_d - mean_o)) /
                                np.max(np.abs(mean_o))),
            var_linf_over_prior=float(np.max(np.abs(var_d - var_o)) / 2.0),
            rows=int(sample_rows))
    return out, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--side", type=int, default=None,
                    help="grid points per dimension (default: the config's)")
    ap.add_argument("--cpu-rows", type=int, default=None,
                    help="rows of the CPU-baseline sample (default: ~12 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import safeopt_amd
    import safeopt_amd.gpy as gpy
    from safeopt_amd import _hip, dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run "
                         "(one process per GPU)")
    ctx, comm = dist.init_from_env()

    cfg = make_config(args.config, side=args.side, rows_y_mult=world)
    gps = build_gps(cfg, gpy)
    ctx.sync()

    if args.config == 5:
        parts = cfg["particles"]
        opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * cfg["d"],
                                       threshold=cfg["threshold"])
        units = parts.shape[0]

        def step():
            for st in ("greedy", "maximizers", "expanders"):
                opt._compute_particle_fitness(st, parts)
        workload = ("config5: SafeOptSwarm fitness, 4-D RBF, G=2, n=2000, "
                    "P=%d particles, greedy+maximizers+expanders" % units)
    else:
        grid = cfg["grid"]
        opt = safeopt_amd.SafeOpt(gps if cfg["G"] > 1 else gps[0], grid,
                                  cfg["fmin"] if cfg["G"] > 1 else 0.0,
                                  threshold=cfg["threshold"], comm=comm)
        units = grid.shape[0]
        last = {}

        def step():
            last["x"] = opt.optimize()
        workload = ("config%d: %d-D %s, G=%d, n=%d, grid %s = %d rows "
                    "(%d per GPU), one SafeOpt.optimize()" %
                    (args.config, cfg["d"], cfg["kernels"][0][0]["kind"],
                     cfg["G"], cfg["n"], "x".join(map(str, cfg["sides"])),
                     units, units // world))

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)
    comm.barrier()
    ctx.sync()
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    comm.barrier()
    dt = time.perf_counter() - t0
    ev_ms = ctx.timer_stop()
    prof_ms, launches, flops = ctx.profile_read()
    ctx.profile_enable(False)
    dt = float(comm.allreduce_max(np.array([dt]))[0])       # MAX over ranks

    if rank != 0:
        return
    ms_per_step = dt * 1e3 / args.steps
    achieved = flops / (prof_ms * 1e-3) / 1e12 if prof_ms > 0 else 0.0
    G, d = cfg["G"], cfg["d"]
    res = {
        "metric": "candidate-points/s (posterior+safe-set sweep)",
        "value": units / (dt / args.steps),
        "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "n_train": cfg["n"], "G": G, "d": d,
                   "rows": int(units)},
        "roofline": {
            "bound": "mfma", "kernel": "k_sweep (posterior_sweep)",
            "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": None,
            "kernel_ms_avg": prof_ms / max(launches, 1),
            "launches": int(launches),
            "algorithmic_flops_per_launch": flops / max(launches, 1),
            "algorithmic_hbm_frac": ((8 * d + 16 * G + 3) * (units / world) /
                                     (prof_ms / max(launches, 1) * 1e-3) /
                                     1e9 / HBM_PEAK_GBS) if prof_ms > 0 else 0.0,
        },
        "hip_event_ms_per_step": ev_ms / args.steps,
    }
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc
    # passes (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process)
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            tr = json.load(f).get("config%d" % args.config)
        if tr and args.side is None and world == 1:
            res["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            res["roofline"]["traffic_source"] = tr["note"]
    except (OSError, ValueError):
        pass
    if args.config != 5:
        res["chosen_x"] = [float(v) for v in np.atleast_1d(last["x"])]
    if world == 1:
        res["mfma_f64_microbench_tflops"] = ctx.microbench_mfma_f64(20000)
    if world == 1 and not args.no_cpu_baseline and args.config != 5:
        rows = None if args.cpu_rows is None else min(args.cpu_rows, units)
        base, parity = cpu_baseline(cfg, rows, dev_Q=opt.Q)
        res["cpu_baseline"] = base
        res["parity"] = parity
        res["speedup_vs_cpu"] = res["value"] / base["value"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
