#!/usr/bin/env python
"""Benchmark of the GP-posterior + safe-set sweep (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

A "step" is one ``SafeOpt.optimize()`` = ``update_confidence_intervals`` +
``compute_sets`` + ``get_new_query_point`` over the whole resident candidate
grid (config 5: one ``SafeOptSwarm._compute_particle_fitness`` call for every
swarm type's GP set).  Inputs are synthetic and already
resident in HBM when the timed region starts.

Default workload = BASELINE.json configs[2] ("config 3", the north-star target):
2-D Matern-5/2, 3 GPs, 500 training observations, 1e6-point grid.

Multi-GPU: one process per GPU.  Under ``torch.distributed.run`` the ranks come
from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; a bare
``python bench.py --gpus N`` spawns the N ranks itself (same environment
variables, rank 0's JSON line is passed through).  Configs 2 and 3 scale weakly
(1e6 rows per rank: the grid gets ``N`` times as many rows in its last
dimension); config 4 is BASELINE.json's fixed 200^3 grid, row-sharded in
contiguous blocks of the flat index (strong scaling).  The only cross-rank
traffic is a handful of scalars per step over RCCL.

Rank 0 prints ONE JSON line.  No torch anywhere: device memory, streams,
events and RCCL all come from libsafeopt_hip.so.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
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
        4: dict(d=3, kind="RBF", G=1, n=1000, side=200, box=5., xr=2.5, seed=5),
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


CPU_BASELINE_SECONDS = 2.5       # target CPU work of ONE timed run of the sample
CPU_BASELINE_REPEATS = 5         # median of this many runs (after one warm-up)
CPU_BASELINE_REPEATS_DEFAULT = 5


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_PAR = {}


def _par_init(cfg):
    """Worker of cpu_baseline's process pool: its own GPs, one BLAS thread."""
    from threadpoolctl import threadpool_limits
    from oracle import gp_numpy as gpn
    _PAR["limit"] = threadpool_limits(limits=1)
    _PAR["gps"] = build_gps(cfg, gpn)
    _PAR["cfg"] = cfg


def _par_run(block):
    from oracle import safeopt_numpy as son
    cfg = _PAR["cfg"]
    scaling = np.sqrt([2.0] * cfg["G"])
    try:
        son.optimize_grid(_PAR["gps"], block, cfg["fmin"], scaling, cfg["threshold"],
                          cfg["beta"])
    except EnvironmentError:          # block without a safe row: sweep only
        son.confidence_intervals(_PAR["gps"], block, cfg["beta"])
    return block.shape[0]


def cpu_baseline_parallel(cfg, rate_one, repeats=3):
    """The same oracle on EVERY host core: candidate rows are independent, so the
    block is cut into nproc row blocks, one single-threaded process each (what a
    user of the reference would do with multiprocessing); warm-up + median."""
    import multiprocessing as mp
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    N = cfg["grid"].shape[0]
    # (the oracle materialises n x rows temporaries: with every hardware thread busy
    # it is bound by memory traffic, a process then runs far below its solo rate --
    # small blocks keep a round at a few seconds)
    per = int(max(500, min(N // nproc, 2000, rate_one * CPU_BASELINE_SECONDS * 0.6)))
    rows = per * nproc
    start = max(0, (N - rows) // 2)
    blocks = [np.ascontiguousarray(cfg["grid"][start + i * per:start + (i + 1) * per])
              for i in range(nproc)]
    small = {k: v for k, v in cfg.items() if k not in ("grid", "sides")}
    ctxmp = mp.get_context("spawn")     # (the parent holds a HIP context: no fork)
    with ctxmp.Pool(nproc, initializer=_par_init, initargs=(small,)) as pool:
        # (bounded waits: a pool that cannot start must not hang the bench line)
        pool.map_async(_par_run, blocks, chunksize=1).get(timeout=300)      # warm-up
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            pool.map_async(_par_run, blocks, chunksize=1).get(timeout=300)
            times.append(time.perf_counter() - t0)
    times.sort()
    t = times[len(times) // 2]
    return dict(value=rows / t, unit="candidates/s", processes=int(nproc), rows=int(rows),
                spread=[rows / max(times), rows / min(times)],
                sample="%d single-threaded processes, %d rows each of the same grid "
                       "(rows [%d, %d)), warm-up + median of %d rounds, %.2f s each"
                       % (nproc, per, start, start + rows, repeats, t))


def _time_oracle(son, gps, grid, cfg, scaling, repeats):
    """Median wall time of the oracle's optimize() over `grid` (one warm-up)."""
    def once():
        t0 = time.perf_counter()
        try:
            out = son.optimize_grid(gps, grid, cfg["fmin"], scaling,
                                    cfg["threshold"], cfg["beta"])
        except EnvironmentError:      # sample without a safe row: sweep only
            out = (None, son.confidence_intervals(gps, grid, cfg["beta"]))
        return time.perf_counter() - t0, out
    once()
    runs = [once() for _ in range(repeats)]
    times = sorted(r[0] for r in runs)
    return times[len(times) // 2], times, runs[-1][1]


def cpu_baseline(cfg, sample_rows, dev_Q=None, row_offset=0):
    """The oracle (NumPy restatement of the reference path) on the host cores,
    on a bounded sample of the same workload: a centred block of the grid.

    Protocol: buffers pre-faulted by a warm-up run,
    median of CPU_BASELINE_REPEATS runs with every BLAS thread, and the same on
    ONE thread (smaller block).  ``sample_rows=None`` sizes the blocks from a
    pilot so that one run is ~CPU_BASELINE_SECONDS of host work."""
    from oracle import gp_numpy as gpn
    from oracle import safeopt_numpy as son
    from threadpoolctl import threadpool_info, threadpool_limits
    gps = build_gps(cfg, gpn)
    N = cfg["grid"].shape[0]
    scaling = np.sqrt([2.0] * cfg["G"])
    cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])

    def block(rows):
        start = max(0, (N - rows) // 2)
        return start, np.ascontiguousarray(cfg["grid"][start:start + rows])

    def pilot_rate():
        _, pg = block(min(N, 20000))
        son.confidence_intervals(gps, pg[:8192], cfg["beta"])
        t0 = time.perf_counter()
        son.confidence_intervals(gps, pg, cfg["beta"])
        return pg.shape[0] / (time.perf_counter() - t0)

    rows_all = sample_rows or int(min(N, max(20000, pilot_rate() * CPU_BASELINE_SECONDS)))
    start, grid = block(rows_all)
    t_all, times_all, (idx, *rest) = _time_oracle(son, gps, grid, cfg, scaling,
                                                  CPU_BASELINE_REPEATS)
    Q = rest[0]
    with threadpool_limits(limits=1):
        rows_one = sample_rows or int(min(N, max(5000, pilot_rate() * CPU_BASELINE_SECONDS * 0.6)))
        _, grid1 = block(rows_one)
        t_one, times_one, _ = _time_oracle(son, gps, grid1, cfg, scaling,
                                           CPU_BASELINE_REPEATS)
    out = dict(value=rows_all / t_all, unit="candidates/s", cores=int(cores),
               kind="port", cpu=_cpu_model(),
               runs=CPU_BASELINE_REPEATS, spread=[rows_all / max(times_all), rows_all / min(times_all)],
               single_thread=dict(value=rows_one / t_one, rows=int(rows_one),
                                  spread=[rows_one / max(times_one), rows_one / min(times_one)]),
               sample="oracle optimize_grid (NumPy/OpenBLAS restatement of "
                      "gp_opt.py:453-649 + GPy predict) on rows [%d, %d) of the "
                      "same grid: warm-up + median of %d runs, %.2f s each on %d "
                      "threads; single thread: %d rows, %.2f s each"
                      % (start, start + rows_all, CPU_BASELINE_REPEATS, t_all,
                         cores, rows_one, t_one))
    # every core at work: nproc independent row blocks.  THIS is the
    # baseline's value -- `cores` = the processes that really ran; the BLAS-threaded run
    # above is elementwise-bound NumPy (faithful to GPy), in effect ONE core whatever the
    # thread count, and is kept as `blas_threads` next to `single_thread`.
    out["blas_threads"] = dict(value=out["value"], threads=int(cores), spread=out.pop("spread"),
                               note="OpenBLAS threads of ONE process: the NumPy restatement "
                                    "is elementwise-bound, in effect a single core")
    try:
        par = cpu_baseline_parallel(cfg, rows_one / t_one)
        out["parallel"] = par
        out["value"], out["cores"] = par["value"], par["processes"]
        out["sample"] = par["sample"] + " | one process: " + out["sample"]
    except Exception as e:        # noqa -- the baseline must never break the bench line
        out["parallel"] = {"error": repr(e)}
        out["cores"] = 1
        out["sample"] = "(parallel run failed: one process, in effect one core) " + out["sample"]
    parity = None
    if dev_Q is not None:
        dq = dev_Q[start - row_offset:start - row_offset + rows_all]
        lo, up = Q[:, ::2], Q[:, 1::2]
        mean_o, mean_d = 0.5 * (lo + up), 0.5 * (dq[:, ::2] + dq[:, 1::2])
        var_o = ((up - lo) / (2 * cfg["beta"])) ** 2
        var_d = ((dq[:, 1::2] - dq[:, ::2]) / (2 * cfg["beta"])) ** 2
        parity = dict(
            mean_linf_rel=float(np.max(np.abs(mean_d - mean_o)) /
                                np.max(np.abs(mean_o))),
            var_linf_over_prior=float(np.max(np.abs(var_d - var_o)) / 2.0),
            rows=int(rows_all))
    return out, parity


def full_grid_check(cfg, opt, chosen_index):
    """One oracle run over the WHOLE grid (checker only, after the timed region):
    are the safe set, maximisers, expanders and the chosen row identical?"""
    from oracle import gp_numpy as gpn
    from oracle import safeopt_numpy as son
    gps = build_gps(cfg, gpn)
    scaling = np.sqrt([2.0] * cfg["G"])
    t0 = time.perf_counter()
    idx, Q, S, M, G = son.optimize_grid(gps, cfg["grid"], cfg["fmin"], scaling,
                                        cfg["threshold"], cfg["beta"])
    return dict(full_grid_oracle_s=time.perf_counter() - t0,
                chosen_index_identical=bool(chosen_index == int(idx)),
                S_identical=bool(np.array_equal(opt.S, S)),
                M_identical=bool(np.array_equal(opt.M, M)),
                G_identical=bool(np.array_equal(opt.G, G)),
                q_linf=float(np.max(np.abs(opt.Q - Q))))


def bo_iteration(cfg, gpy, safeopt_amd, ctx, iters=12):
    """What a user of the drop-in pays per sequential BO iteration with the product
    defaults (shared factor, bordered factor update + closed-form rank-1 refresh of the
    resident posterior, safeopt/gp_opt.py:230-255 then :651-675): median ms of
    ``add_new_data_point`` + ``optimize`` over ``iters`` iterations, candidates/s, and
    whether the query points equal those of the full-refit path.  Also the hipEvent
    time of the rank-1 refresh alone (``rank1_roofline``)."""
    out = {}
    old_share = ctx.set_share(True)
    try:
        runs = {}
        for incremental in (False, True):
            gps = build_gps(cfg, gpy)
            for g in gps:
                g.incremental = incremental
            G = cfg["G"]
            opt = safeopt_amd.SafeOpt(gps if G > 1 else gps[0], cfg["grid"],
                                      cfg["fmin"] if G > 1 else 0., threshold=cfg["threshold"])
            opt._backend.incremental = incremental
            opt._backend.refresh_every = 1 << 30
            xs, t_add, t_opt, t_r1 = [], [], [], []
            x = opt.optimize()
            for it in range(iters):
                y = np.array([[_bumps(np.atleast_2d(x), 100 + cfg["k"] - 1 + g)[0] + 1.0
                               for g in range(G)]])
                ctx.sync(); t0 = time.perf_counter()
                opt.add_new_data_point(x, y)
                ctx.sync(); t1 = time.perf_counter()
                x = opt.optimize()
                ctx.sync(); t2 = time.perf_counter()
                t_add.append(t1 - t0); t_opt.append(t2 - t1); xs.append(np.array(x))
            if incremental:
                # the refresh of the resident posterior alone (k_rank1 + Q / S), between
                # two events on the library's stream (its own iterations: the explicit
                # update takes the pending append, optimize() then sweeps in full)
                for it in range(6):
                    y = np.array([[_bumps(np.atleast_2d(x), 100 + cfg["k"] - 1 + g)[0] + 1.0
                                   for g in range(G)]])
                    opt.add_new_data_point(x, y)
                    ctx.sync()
                    ctx.timer_start()
                    opt.update_confidence_intervals()
                    t_r1.append(ctx.timer_stop())
                    x = opt.optimize()
            runs[incremental] = (np.array(xs), float(np.median(t_add[2:]) * 1e3),
                                 float(np.median(t_opt[2:]) * 1e3),
                                 float(np.median(t_r1[1:])) if t_r1 else None)
        xf, addf, optf, _ = runs[False]
        xi, addi, opti, r1 = runs[True]
        rows = cfg["grid"].shape[0]
        out = {
            "note": "product defaults: shared factor + one-row factor update + rank-1 refresh "
                    "of the resident posterior (sgp_gp_append, sgp_grid_rank1_update); "
                    "median of iterations 3..%d from n = %d" % (iters, cfg["n"]),
            "ms": addi + opti, "add_new_data_point_ms": addi, "optimize_ms": opti,
            "value": rows / ((addi + opti) * 1e-3), "unit": "candidates/s",
            "full_refit_ms": addf + optf,
            "same_query_points": bool(np.array_equal(xf, xi)),
        }
        if r1:
            # VALU-bound: n covariance evaluations + 1 FMA per row and updated GP;
            # roof = the chip's fp64 vector rate in lane-operations (256 CUs x 4 SIMDs
            # x 16 lanes at 2.4 GHz); ~22 instructions per RBF / 30 per Matern value
            per = 22 if cfg["kernels"][0][0]["kind"] == "RBF" else 30
            # (GPs that share the factor of the GP in front of them -- same inputs, kernel,
            # noise: the outputs of a multi-output GP -- reuse its c(x): evaluations are
            # counted once per group)
            groups = 1 + sum(1 for g in range(1, cfg["G"])
                             if cfg["kernels"][g] != cfg["kernels"][g - 1])
            ops = float(rows) * groups * (cfg["n"] + iters // 2) * per
            out["rank1_roofline"] = {
                "bound": "fp64 valu", "kernel": "k_rank1 (+ Q / S epilogue)",
                "ms": r1, "achieved": ops / (r1 * 1e-3) / 1e12,
                "peak": 39.3, "unit": "T lane-ops/s",
                "frac": ops / (r1 * 1e-3) / 1e12 / 39.3,
                "ops_note": "%d instructions per covariance value x n x %d group(s) of GPs with "
                            "one factor x rows" % (per, groups)}
    finally:
        ctx.set_share(old_share)
    return out


def sets_roofline(opt, ctx, cfg, rows, steps=20):
    """The set passes of one step (compute_sets + get_new_query_point,
    gp_opt.py:483-649) between two events on the library's stream, after a fresh
    confidence pass: HBM-bound chain of masked passes over Q / masks / widths."""
    G, d = cfg["G"], cfg["d"]
    ms = []
    for _ in range(steps):
        opt.update_confidence_intervals()
        ctx.sync()
        ctx.timer_start()
        opt.compute_sets()
        opt.get_new_query_point()
        ms.append(ctx.timer_stop())
    ms.sort()
    t = ms[len(ms) // 2]
    # bytes a row costs: maximisers (l0, u0, S -> M), candidates (Q, S, M -> mask, width),
    # expander pre-filter of one candidate (S, the row, mean / var of the active GPs),
    # arg-max (Q, M, G)
    per_row = (16 + 2) + (16 * G + 2 + 9) + (1 + 8 * d + 16 * G) + (16 * G + 2)
    gbs = per_row * rows / (t * 1e-3) / 1e9
    return {"bound": "hbm", "kernels": "set passes of one step (sets.hip + expander scan)",
            "ms": t, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_row": per_row,
            "note": "event-to-event on the library's stream: kernels + launch gaps + the one "
                    "host round trip of the step"}


def reference_regime(gpy, safeopt_amd, ctx, n=20, steps=200):
    """The regime of the reference's own examples and tests -- a GP with ~20 observations
    (2-D RBF here, on the 1000 x 1000 grid of config 2): ms per SafeOpt.optimize(), the
    sweep launch alone (sweep_tiny.hip: one thread per row on the fp64 VALU), its
    fraction of the fp64-VALU roof (22 instructions per covariance value, n (n + 1) / 2
    + n FMAs per row) -- and the oracle on a block of the same rows."""
    cfg = make_config(2)
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, size=(n, 2))
    Y = (_bumps(X, 3) - _bumps(X, 3).min() + 0.5)[:, None]
    gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(2, variance=2.0, lengthscale=[1.0, 1.0], ARD=True),
                                 noise_var=0.05 ** 2)
    opt = safeopt_amd.SafeOpt(gp, cfg["grid"], 0.0, threshold=cfg["threshold"])
    for _ in range(20):
        opt.optimize()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.optimize()
    ctx.sync()
    step_ms = (time.perf_counter() - t0) * 1e3 / steps
    ctx.profile_enable(True)
    for _ in range(20):
        opt.optimize()
    ctx.sync()
    ms, launches, _ = ctx.profile_read()
    ctx.profile_enable(False)
    t = ms / max(launches, 1)
    N = cfg["grid"].shape[0]
    ops = (22.0 * n + n * (n + 1) / 2.0 + n) * N
    return {"workload": "2-D RBF, G=1, n=%d, grid 1000x1000, one SafeOpt.optimize()" % n,
            "ms_per_step": step_ms, "value": N / (step_ms * 1e-3), "unit": "candidates/s",
            "sweep_kernel": ctx.last_sweep(), "sweep_ms": t,
            "roofline": {"bound": "fp64 valu", "achieved": ops / (t * 1e-3) / 1e12, "peak": 39.3,
                         "unit": "T lane-ops/s", "frac": ops / (t * 1e-3) / 1e12 / 39.3}}


def no_expander_state(gpy, safeopt_amd, ctx):
    """The expander loop of gp_opt.py:557-612 where it has to visit EVERY candidate: a converged
    run (tests/_scenarios.py: a safe disk whose rim is observed densely, a coarsely observed
    plateau inside -- thousands of candidates wider than every maximiser -- and no unsafe row
    within reach of any of them; 2-D RBF), and full_sets = True (:553-555: every safe row is
    visited).  Two states: config-2 scale (1e6 rows, 217 observations) and a denser problem on
    a 1e5-row grid (637 observations).  ms per SafeOpt.optimize() / compute_sets(full_sets=True)
    with the big passes of round 6 (sgp_grid_expander_pass: hundreds to thousands of candidates
    per device pass, two synchronisations each) and with the 16-candidates-per-round-trip loop
    they replace.  Parity: tests/test_gpu_expander_passes.py."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import _scenarios as sc
    out = {"note": no_expander_state.__doc__.split("  ms per")[0].strip().replace("\n   ", ""),
           "pass_sizes": "SafeOpt._pass_size: by the number of observations"}
    states = (("config2_scale_1e6_rows", 1000,
               dict(ls=0.7, rings=5, dring=0.3, dmid=0.8, dtop=0.4, r0=2.0, dout=1.4, plateau=0.6)),
              ("grid_320x320", 320, dict(r0=2.0, rings=8, ls=0.4, dmid=0.45, plateau=0.6)))
    for name, side, kw in states:
        gp, grid = sc.converged_state(side, 0.05, ns=gpy, **kw)
        row = {"rows": int(len(grid)), "n_train": int(gp.X.shape[0])}
        for big in (True, False):
            opt = safeopt_amd.SafeOpt(gp, grid, 0.0, threshold=0.1)
            opt.big_passes = big
            passes = []
            attr = "expander_pass" if big else "expander_batch"
            orig = getattr(opt._backend, attr)
            setattr(opt._backend, attr, lambda *a, _o=orig, _i=(7 if big else -1): (passes.append(a[_i]), _o(*a))[1])
            opt.optimize()
            times = []
            for _ in range(5 if big else 1):          # (median of five)
                del passes[:]
                ctx.sync()
                t0 = time.perf_counter()
                opt.optimize()
                ctx.sync()
                times.append((time.perf_counter() - t0) * 1e3)
            ms = float(np.median(times))
            key = "big_passes" if big else "sixteen_per_round_trip"
            S = np.asarray(opt.S, dtype=bool)
            row.setdefault("unsafe_rows", int((~S).sum()))
            row.setdefault("safe_rows", int(S.sum()))
            if big:
                row["pass_sizes"] = list(passes)
            row[key] = {"optimize_ms": ms, "device_passes": len(passes),
                        "expanders_found": int(np.asarray(opt.G).sum())}
            if big or side <= 400:
                if big:
                    opt.compute_sets(full_sets=True)        # (first call: the scratch buffers grow)
                ctx.sync()
                t0 = time.perf_counter()
                opt.compute_sets(full_sets=True)
                ctx.sync()
                row[key]["full_sets_ms"] = (time.perf_counter() - t0) * 1e3
                row[key]["full_sets_expanders"] = int(np.asarray(opt.G).sum())
            if big:
                # candidates of the state: S & ~M & wider than every maximiser & above the threshold
                opt.optimize()
                Q = np.asarray(opt.Q)
                w = Q[:, 1] - Q[:, 0]
                M = np.asarray(opt.M, dtype=bool)
                row["candidates"] = int((S & ~M & (w > w[M].max()) & (w > 0.1 * 2.0)).sum())
        if side >= 1000:
            # the same state with Lipschitz certificates (gp_opt.py:558-576; L = 1: no candidate
            # reaches an unsafe row): sgp_grid_lipschitz_pass against the 16-candidate loop
            lip = {"L": 1.0}
            for big in (True, False):
                opt = safeopt_amd.SafeOpt(gp, grid, 0.0, lipschitz=1.0, threshold=0.1)
                opt.big_passes = big
                opt.optimize()
                times = []
                for _ in range(5 if big else 1):
                    ctx.sync()
                    t0 = time.perf_counter()
                    opt.optimize()
                    ctx.sync()
                    times.append((time.perf_counter() - t0) * 1e3)
                lip["big_passes_ms" if big else "sixteen_per_round_trip_ms"] = float(np.median(times))
                lip["expanders_found"] = int(np.asarray(opt.G).sum())
            row["lipschitz_certificates"] = lip
        out[name] = row
    return out


def config1(gpy, safeopt_amd, ctx, steps=2000):
    """BASELINE.json configs[0] -- the reference's own problem size, examples/1d_example.ipynb
    plumbing: 1-D RBF (variance 2, lengthscale 1, noise 0.05^2), 1 GP that is objective and
    constraint, the 1000-point grid on [-10, 10], 20 observations gathered by running
    SafeOpt from x0 = 0 on a sampled GP function (utilities.sample_gp_function), threshold
    0.2 -- microseconds per SafeOpt.optimize().  Here the whole step is ONE launch of one
    workgroup (sgp_grid_step_small); the large-grid path on the same object (sweep + nine
    set passes) is timed next to it, and the oracle once on the same intervals."""
    from oracle import gp_numpy as gpn
    from oracle import safeopt_numpy as son
    rng_state = np.random.get_state()
    np.random.seed(0)
    try:
        bounds = [(-10., 10.)]
        kern = gpy.kern.RBF(1, variance=2.0, lengthscale=1.0)
        grid = safeopt_amd.linearly_spaced_combinations(bounds, 1000)
        fun = safeopt_amd.sample_gp_function(kern, bounds, 0.05 ** 2, 100)
        x0 = np.zeros((1, 1))
        gp = gpy.models.GPRegression(x0, fun(x0), kern, noise_var=0.05 ** 2)
        # (fmin one unit below the start value: the start is safe, the set can grow)
        opt = safeopt_amd.SafeOpt(gp, grid, float(fun(x0, noise=False)[0, 0]) - 1.0, threshold=0.2)

        def timed(small, reps):
            opt.small_step = small
            opt._backend.incremental = False
            for _ in range(50):
                xx = opt.optimize()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                xx = opt.optimize()
            ctx.sync()
            return (time.perf_counter() - t0) / reps * 1e6, xx

        # along the run: what a step costs while the safe set still grows (the first
        # candidate in visiting order is an expander: the whole step is the one launch) and
        # once no expander is left (the loop then visits every candidate, 16 per round trip)
        along = []
        for it in range(19):                   # 20 observations, as 1d_example gathers them
            if it in (2, 5, 9, 14):
                us1, _ = timed(True, 300)
                us2, _ = timed(False, 300)
                opt.small_step = True
                opt.optimize()
                along.append({"n": int(opt.gp.X.shape[0]), "one_launch_us": us1,
                              "large_grid_path_us": us2, "expander_found": bool(opt.G.any()),
                              "safe_rows": int(opt.S.sum())})
            opt.small_step = True
            x = opt.optimize()
            opt.add_new_data_point(x, fun(x))
    finally:
        np.random.set_state(rng_state)
    n = opt.gp.X.shape[0]
    out = {"workload": "config1: 1-D RBF, G=1, n=%d (19 SafeOpt iterations from x0 = 0 on a "
                       "sampled GP function), grid 1000 points on [-10, 10], one "
                       "SafeOpt.optimize()" % n,
           "along_the_run": along}
    for name, small in (("one_launch", True), ("large_grid_path", False)):
        us, x = timed(small, steps)
        out[name] = {"us_per_optimize": us, "candidates_per_s": 1000 / (us * 1e-6),
                     "sweep_kernel": ctx.last_sweep(), "chosen_x": float(x[0])}
    opt.small_step = True
    x = opt.optimize()
    go = gpn.GPRegression(opt.gp.X, opt.gp.Y, gpn.RBF(1, variance=2.0, lengthscale=1.0),
                          noise_var=0.05 ** 2)
    t0 = time.perf_counter()
    idx, Qo, So, Mo, Go = son.optimize_grid([go], grid, opt.fmin, opt.scaling, 0.2, 2.0)
    out["oracle"] = {"us_per_optimize": (time.perf_counter() - t0) * 1e6,
                     "same_chosen_x": bool(np.array_equal(x, grid[idx])),
                     "S_M_G_identical": bool(np.array_equal(opt.S, So) and np.array_equal(opt.M, Mo)
                                             and np.array_equal(opt.G, Go)),
                     "q_linf": float(np.max(np.abs(opt.Q - Qo)))}
    out["oracle"]["expander_found"] = bool(Go.any())
    out["us_per_optimize"] = out["one_launch"]["us_per_optimize"]
    return out


def config4_strong(gpy, safeopt_amd, dist, ctx, comm, rank, world, steps=None, warmup=2):
    """BASELINE.json's 8-GPU config (3-D RBF, n = 1000, the fixed 200^3 grid,
    row-sharded in contiguous blocks of the flat index): ms per SafeOpt.optimize() and
    candidates/s at THIS number of ranks -- strong scaling, whatever --config the line
    itself is for."""
    if steps is None:
        steps = 4 if world == 1 else 8 * min(world, 4)     # (137 ms / ranks per step)
    cfg = make_config(4)
    gps = build_gps(cfg, gpy)
    opt = safeopt_amd.SafeOpt(gps[0], cfg["grid"], 0.0, threshold=cfg["threshold"], comm=comm)
    for _ in range(warmup):
        x = opt.optimize()
    comm.barrier(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        x = opt.optimize()
    ctx.sync(); comm.barrier()
    dt = float(comm.allreduce_max(np.array([time.perf_counter() - t0]))[0])
    rows = cfg["grid"].shape[0]
    lo, hi = dist.shard_range(rows, rank, world)
    return {"workload": "config4: 3-D RBF, G=1, n=1000, grid 200x200x200 = 8000000 rows, "
                        "%d per GPU (contiguous blocks of the flat index)" % (hi - lo),
            "scaling": "strong", "n_gpus": world, "steps": steps,
            "ms_per_step": dt * 1e3 / steps, "value": rows / (dt / steps),
            "unit": "candidates/s", "chosen_x": [float(v) for v in np.atleast_1d(x)]}


def _probe_transport(kind):
    """``bench.py --probe-transport KIND`` (a child of every rank, under a watchdog): bring
    the transport up and take ONE certified N-rank step on a small problem.  Exit code 0 =
    this rank got through."""
    import safeopt_amd
    import safeopt_amd.gpy as gpy
    from safeopt_amd import dist
    if kind == "rccl-host":
        os.environ["SAFEOPT_RCCL_IN_STREAM"] = "0"
    ctx, comm = dist.init_from_env(timeout=60.0)
    rng = np.random.default_rng(3)
    X = rng.uniform(-1.5, 1.5, size=(40, 2))
    Y = _bumps(X, 7)[:, None]
    Y = Y - Y.min() + 0.5
    gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(2, 2.0, [1.0, 1.0], ARD=True),
                                 noise_var=0.05 ** 2)
    grid = safeopt_amd.linearly_spaced_combinations([(-4., 4.)] * 2, [96, 64])
    opt = safeopt_amd.SafeOpt(gp, grid, 0.0, threshold=0.2, comm=comm)
    x = opt.optimize()
    ctx.sync()
    comm.barrier()
    assert np.all(np.isfinite(x))


def choose_transport(rank, world, timeout=None):
    """Which transport the N ranks use, decided under a watchdog so that a hang in
    ncclCommInitRank or in an in-stream collective yields a JSON line, not a timeout.

    Chain: RCCL with the certified step in stream (the product default) -> RCCL with the
    collectives on the host side of the step (SAFEOPT_RCCL_IN_STREAM=0) -> TCP
    (SAFEOPT_COMM=socket, the collectives staged through the host).  Every rank starts the
    same probe in a child process (its own rendezvous tag), the ranks agree on the outcome
    over a TCP control channel, the first variant that every rank got through is taken.
    ``SAFEOPT_COMM`` / ``SAFEOPT_RCCL_IN_STREAM`` set by the operator skip the probes."""
    from safeopt_amd import dist
    if timeout is None:
        timeout = float(os.environ.get("SAFEOPT_BENCH_PROBE_TIMEOUT", "150"))
    report = {"chain": [], "chosen": None}
    if os.environ.get("SAFEOPT_COMM") == "socket":
        report["chosen"] = "socket (SAFEOPT_COMM)"
        return report
    if os.environ.get("SAFEOPT_BENCH_PROBE", "1") == "0":
        report["chosen"] = "rccl (not probed: SAFEOPT_BENCH_PROBE=0)"
        return report
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = int(os.environ.get("MASTER_PORT", "29500"))
    ctl = dist.SocketComm(rank, world, addr, mport + 23, timeout=120.0)
    kinds = ["rccl-in-stream", "rccl-host"]
    if os.environ.get("SAFEOPT_RCCL_IN_STREAM", "1") == "0":
        kinds = ["rccl-host"]
    for k, kind in enumerate(kinds):
        env = dict(os.environ, MASTER_PORT=str(mport + 40 + 3 * k),
                   SAFEOPT_RDZV_NONCE="probe%d-%s" % (k, os.environ.get("SAFEOPT_RDZV_NONCE", "")))
        env.pop("SAFEOPT_RDZV_PORT", None)
        env["SAFEOPT_RDZV"] = "tcp"      # (the file rendezvous names a launch by its parent pid)
        t0 = time.perf_counter()
        cmd = [sys.executable, os.path.abspath(__file__), "--probe-transport", kind]
        if os.environ.get("SAFEOPT_BENCH_PROBE_CMD"):      # (tests: a probe that hangs / fails)
            cmd = os.environ["SAFEOPT_BENCH_PROBE_CMD"].split("\x1f")
        child = subprocess.Popen(cmd, env=env,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        try:
            _, err = child.communicate(timeout=timeout)
            rc = child.returncode
        except subprocess.TimeoutExpired:
            child.kill()                      # (this very process, nothing by pattern)
            _, err = child.communicate()
            rc = -9
        bad = float(ctl.allreduce_max(np.array([0.0 if rc == 0 else 1.0]))[0])
        report["chain"].append({"transport": kind, "ok": bad == 0.0, "rc_this_rank": rc,
                                "s": round(time.perf_counter() - t0, 1),
                                "stderr_tail": None if rc == 0 else (err or "")[-300:]})
        if bad == 0.0:
            report["chosen"] = kind
            break
    ctl.close()
    if report["chosen"] is None:
        report["chosen"] = "socket (fallback)"
    if report["chosen"] == "rccl-host":
        os.environ["SAFEOPT_RCCL_IN_STREAM"] = "0"
    if report["chosen"].startswith("socket"):
        os.environ["SAFEOPT_COMM"] = "socket"
    return report


def nrank_selfcheck(opt, comm):
    """One step through BOTH variants of the N-rank step on the same intervals -- the
    one-round-trip step with the merges on the device behind the collectives
    (sgp_grid_sets_fused_comm) and the host-side variant (sets_front / sets_back, the
    merges in NumPy) -- must give the same chosen row and the same set sizes."""
    from safeopt_amd import _hip

    def once():
        x = opt.optimize()
        be = opt._backend
        cnt = np.array([float(be.download(w).sum()) for w in (_hip.S, _hip.M, _hip.G)])
        return x, comm.allgather(cnt).sum(axis=0)
    had = getattr(comm, "in_stream", False)
    xa, ca = once()
    try:
        comm.in_stream = False
        xb, cb = once()
    finally:
        comm.in_stream = had
    return {"in_stream_variant_ran": bool(had),
            "same_chosen_x": bool(np.array_equal(xa, xb)),
            "S_M_G_counts": [int(v) for v in ca], "S_M_G_counts_host_variant": [int(v) for v in cb],
            "ok": bool(np.array_equal(xa, xb) and np.array_equal(ca, cb))}


def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` from a bare shell: start the N ranks (one
    process per GPU, torchrun-style environment) and pass rank 0's line on."""
    import socket
    with socket.socket() as sk:                  # a free port for the launch
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    nonce = "%d-%x" % (os.getpid(), int(time.time() * 1e6))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SAFEOPT_RDZV_NONCE=nonce)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + argv, env=env,
            stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("rank exit codes: %s" % rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: enough for a timed region >= 0.5 s: "
                         "40 at configs 3 and 5, 500 at config 2, 6 at config 4)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=None, choices=[2, 3, 4, 5],
                    help="BASELINE.json config (default: 3, the north-star config, plus "
                         "the key config4_strong: the 8-GPU config at this number of ranks)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip bo_iteration / sets_roofline / config4_strong")
    ap.add_argument("--side", type=int, default=None,
                    help="grid points per dimension (default: the config's)")
    ap.add_argument("--cpu-rows", type=int, default=None,
                    help="rows of the CPU-baseline sample (default: ~2.5 s per run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-chosen", dest="check_chosen", action="store_true", default=None,
                    help="after the timed region, run the oracle ONCE on the whole grid and "
                         "report whether S / M / G and the chosen index are identical "
                         "(default: on at configs 2 and 3 -- 7 s / 28 s of host work)")
    ap.add_argument("--no-check-chosen", dest="check_chosen", action="store_false")
    ap.add_argument("--no-shared-pass", action="store_true",
                    help="skip the second timing pass with the factor shared between identical "
                         "GPs (profiles: every k_sweep launch of the run is then the headline path)")
    ap.add_argument("--profile-steps", type=int, default=10,
                    help="steps of the separate (untimed) per-launch hipEvent pass")
    ap.add_argument("--launch-check", action="store_true",
                    help="only rendezvous the ranks (no device): launcher self-test")
    ap.add_argument("--probe-transport", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.probe_transport:
        return _probe_transport(args.probe_transport)
    default_run = args.config is None
    if default_run:
        args.config = 3
    if args.steps is None:
        args.steps = {2: 500, 3: 40, 4: 6, 5: 40}[args.config]
    if args.check_chosen is None:
        args.check_chosen = args.config in (2, 3) and args.side is None

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])
    if args.launch_check:
        from safeopt_amd import dist
        rank, world, ok = dist.launch_check()
        if not ok:
            raise SystemExit("rank %d: rendezvous token mismatch" % rank)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world,
                              "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}))
        return

    import safeopt_amd
    import safeopt_amd.gpy as gpy
    from safeopt_amd import dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    transport = choose_transport(rank, world) if world > 1 else None
    if (world > 1 and os.environ.get("SAFEOPT_REQUIRE_RCCL") == "1"
            and not transport["chosen"].startswith(("rccl-in-stream", "rccl (not probed"))):
        # the operator asked for the product path or nothing: no silent TCP number
        if rank == 0:
            print(json.dumps({"metric": "candidate-points/s (posterior+safe-set sweep)",
                              "value": None, "n_gpus": world,
                              "error": "SAFEOPT_REQUIRE_RCCL=1: RCCL with the step in stream did "
                                       "not come up on every rank",
                              "transport": transport}), flush=True)
        sys.exit(3)
    ctx, comm = dist.init_from_env()

    # configs 2/3: weak scaling (1e6 rows per rank); config 4: the fixed 200^3
    # grid of BASELINE.json, row-sharded by SafeOpt (strong scaling)
    weak = args.config != 4
    cfg = make_config(args.config, side=args.side,
                      rows_y_mult=world if weak else 1)
    gps = build_gps(cfg, gpy)
    ctx.sync()
    # The headline is the reference's arithmetic: every GP swept on its own
    # (SURVEY 8d counts G (n^2 + 2n) flops per row).  The product shares the factor
    # between GPs with identical inputs (config 3 is such a multi-output GP):
    # measured separately below, reported under "shared_factor".
    ctx.set_share(False)

    last = {}
    if args.config == 5:
        parts = cfg["particles"]
        opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * cfg["d"],
                                       threshold=cfg["threshold"])
        units = parts.shape[0]
        rows_rank = units

        # P / t of ONE _compute_particle_fitness call over all G GPs
        def step():
            opt._compute_particle_fitness("maximizers", parts)

        def step3():
            for st in ("greedy", "maximizers", "expanders"):
                opt._compute_particle_fitness(st, parts)
        workload = ("config5: ONE SafeOptSwarm._compute_particle_fitness call "
                    "('maximizers': both GPs), 4-D RBF, G=2, n=2000, P=%d particles "
                    "[unit since round 4: particles / ONE fitness call (SURVEY 8d); rounds 1-3 "
                    "timed the greedy + maximizers + expanders step -- see three_call_step]" % units)
    else:
        grid = cfg["grid"]
        opt = safeopt_amd.SafeOpt(gps if cfg["G"] > 1 else gps[0], grid,
                                  cfg["fmin"] if cfg["G"] > 1 else 0.0,
                                  threshold=cfg["threshold"], comm=comm)
        units = grid.shape[0]
        lo, hi = dist.shard_range(units, rank, world)
        rows_rank = hi - lo

        def step():
            last["x"] = opt.optimize()
        workload = ("config%d: %d-D %s, G=%d, n=%d, grid %s = %d rows "
                    "(%d per GPU, contiguous blocks of the flat index), one "
                    "SafeOpt.optimize()" %
                    (args.config, cfg["d"], cfg["kernels"][0][0]["kind"],
                     cfg["G"], cfg["n"], "x".join(map(str, cfg["sides"])),
                     units, rows_rank))

    selfcheck = None
    if world > 1 and args.config != 5:
        try:
            selfcheck = nrank_selfcheck(opt, comm)
        except Exception as e:      # noqa -- reported, never fatal
            selfcheck = {"ok": False, "error": repr(e)}

    # (the clocks ramp up over the first ~30 ms of load -- profiles/r04/clock_ramp.txt:
    # a kernel is ~10 % slower in the first launches of a process -- so a short run of
    # untimed steps comes in front of the W warm-up steps)
    # (N ranks: every step is a sequence of collectives, so the ranks must take the SAME
    # number of ramp steps -- the slowest clock decides for all)
    t_ramp = time.perf_counter()
    while True:
        go = 1.0 if time.perf_counter() - t_ramp < 0.1 else 0.0
        if world > 1:
            go = float(comm.allreduce_max(np.array([go]))[0])
        if go <= 0.0:
            break
        step()
    for _ in range(args.warmup):
        step()
    # ---- the timed region: K steps, nothing but the path
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
    dt = float(comm.allreduce_max(np.array([dt]))[0])       # MAX over ranks

    # ---- separate, untimed pass: a hipEvent pair around every sweep launch on
    # the library's stream (roofline.achieved); rocprofv3 --kernel-trace of the
    # same command must agree (profiles/)
    ctx.profile_enable(True)
    for _ in range(max(1, args.profile_steps)):
        step()
    ctx.sync()
    prof_ms, launches, flops = ctx.profile_read()
    ctx.profile_enable(False)
    sweep_kernel = {"pair": "k_sweep_pair (posterior sweep, paired waves: n > 256)",
                    "classic": "k_sweep (posterior sweep, 4 waves)",
                    "mid": "k_sweep_mid (posterior sweep with the factor resident in LDS: 49 .. 128 "
                           "observations; on tensor grids with factor tables up to 256, in passes "
                           "of row blocks -- one timed launch = all passes of all GPs)",
                    "tiny": "k_sweep_tiny (one thread per row, fp64 VALU: up to 48 observations)",
                    }.get(ctx.last_sweep(), str(ctx.last_sweep()))

    # ---- the product's default: consecutive GPs with identical (X, kernel, noise)
    # share the variance contraction -- own timing, own flop count
    shared = None
    if cfg["G"] > 1 and not args.no_shared_pass:
        ctx.set_share(True)
        for _ in range(args.warmup):
            step()
        comm.barrier()
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.sync()
        comm.barrier()
        dts = float(comm.allreduce_max(np.array([time.perf_counter() - t1]))[0])
        ctx.profile_enable(True)
        for _ in range(max(1, args.profile_steps)):
            step()
        ctx.sync()
        s_ms, s_launches, _ = ctx.profile_read()
        ctx.profile_enable(False)
        ctx.set_share(False)
        shared = (dts, s_ms, s_launches)

    # ---- config 5: the step of three calls a swarm iteration makes (extra key)
    three = None
    if args.config == 5:
        for _ in range(2):
            step3()
        ctx.sync(); t3 = time.perf_counter()
        for _ in range(max(4, args.steps // 4)):
            step3()
        ctx.sync()
        three = (time.perf_counter() - t3) / max(4, args.steps // 4)

    # ---- extra keys (never inside the timed region above)
    extras = {}
    if not args.no_extras and args.config != 5 and args.side is None:
        if world == 1:
            try:
                extras["sets_roofline"] = sets_roofline(opt, ctx, cfg, rows_rank)
                extras["bo_iteration"] = bo_iteration(cfg, gpy, safeopt_amd, ctx)
                # (also at the top level, next to sets_roofline)
                if "rank1_roofline" in extras["bo_iteration"]:
                    extras["rank1_roofline"] = extras["bo_iteration"]["rank1_roofline"]
            except Exception as e:      # noqa -- an extra must never break the line
                extras["extras_error"] = repr(e)
        if default_run and world == 1:
            try:
                extras["reference_regime"] = reference_regime(gpy, safeopt_amd, ctx)
            except Exception as e:      # noqa
                extras["reference_regime"] = {"error": repr(e)}
            try:
                extras["config1"] = config1(gpy, safeopt_amd, ctx)
            except Exception as e:      # noqa
                extras["config1"] = {"error": repr(e)}
            try:
                extras["no_expander_state"] = no_expander_state(gpy, safeopt_amd, ctx)
            except Exception as e:      # noqa
                extras["no_expander_state"] = {"error": repr(e)}
        if default_run:
            try:
                extras["config4_strong"] = config4_strong(gpy, safeopt_amd, dist, ctx, comm,
                                                          rank, world)
            except Exception as e:      # noqa
                extras["config4_strong"] = {"error": repr(e)}

    # ---- what the ranks did, for the N-rank line
    per_rank_ms = comm.allgather(np.array([prof_ms / max(launches, 1)]))[:, 0]
    t2 = time.perf_counter()
    for _ in range(50):
        comm.allreduce_max(np.zeros(2))
    coll_us = (time.perf_counter() - t2) / 50 * 1e6
    rccl_ranks = ctx.comm_count() if world > 1 or os.environ.get("SAFEOPT_FORCE_RCCL") == "1" else 1

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
        "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "n_train": cfg["n"], "G": G, "d": d,
                   "rows": int(units), "rows_per_gpu": int(rows_rank),
                   "sharding": "contiguous row blocks (dist.shard_range)",
                   "scaling_note": ("weak: %d rows per rank, the grid's last dimension grows "
                                    "with the ranks" % rows_rank) if weak else
                                   "strong: BASELINE.json's fixed 200^3 grid, 8e6 / ranks rows each",
                   "share_factors": False},
        "rccl_ranks": int(rccl_ranks),
        # at a glance (N > 1): anything but "rccl-in-stream" here means the RCCL path FAILED
        # its probe and the number below was measured over a fallback transport
        "transport_chosen": None if transport is None else transport["chosen"],
        "transport": transport, "nrank_selfcheck": selfcheck,
        "nrank_step": (None if world == 1 else
                       "in stream (sgp_grid_sets_fused_comm: one round trip, merges on the device)"
                       if getattr(comm, "in_stream", False) else
                       "host side (sets_front / sets_back: three round trips)"),
        "per_rank_sweep_ms": [float(v) for v in per_rank_ms],
        "scalar_allreduce_us": coll_us if world > 1 else None,
        "roofline": {
            "bound": "mfma",
            "kernel": sweep_kernel,
            "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": None,
            "kernel_ms_avg": prof_ms / max(launches, 1),
            "launches": int(launches),
            "algorithmic_flops_per_launch": flops / max(launches, 1),
            "algorithmic_hbm_bytes_per_launch": (8 * d + 16 * G + 3) * rows_rank,
            "algorithmic_hbm_frac": ((8 * d + 16 * G + 3) * rows_rank /
                                     (prof_ms / max(launches, 1) * 1e-3) /
                                     1e9 / HBM_PEAK_GBS) if prof_ms > 0 else 0.0,
        },
        "hip_event_ms_per_step": ev_ms / args.steps,
        "timed_region_s": dt,
    }
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc
    # passes (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process)
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            tr = json.load(f).get("config%d" % args.config)
        if tr and args.side is None and world == 1:
            res["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            res["roofline"]["traffic_source"] = tr["note"]
            if "mfma_pipe_busy" in tr:       # SQ_VALU_MFMA_BUSY_CYCLES, same PMC passes
                res["roofline"]["mfma_pipe_busy"] = tr["mfma_pipe_busy"]
                res["roofline"]["clock_ghz_under_profiler"] = tr["clock_ghz_under_profiler"]
    except (OSError, ValueError):
        pass
    if shared:
        dts, s_ms, s_launches = shared
        sfl = (cfg["n"] ** 2 + 2.0 * G * cfg["n"]) * rows_rank
        res["shared_factor"] = {
            "note": "the product default (sgp_ctx_set_share): the GPs of this config have "
                    "identical inputs, kernel and noise, so |L^-1 k|^2 is formed once and "
                    "alpha . k per GP -- the followers ride in their leader's stages; same "
                    "bits (tests/test_gpu_posterior.py)",
            "value": units / (dts / args.steps), "unit": "candidates/s",
            "ms_per_step": dts * 1e3 / args.steps,
            "kernel_ms_avg": s_ms / max(s_launches, 1),
            "algorithmic_flops_per_launch": sfl,
            "achieved": sfl / (s_ms / max(s_launches, 1) * 1e-3) / 1e12 if s_ms > 0 else 0.0,
        }
        res["shared_factor"]["frac"] = res["shared_factor"]["achieved"] / FP64_MFMA_PEAK_TFLOPS
    res.update(extras)
    if three is not None:
        res["three_call_step"] = {
            "note": "greedy + maximizers + expanders: the three fitness calls of one swarm "
                    "iteration of SafeOptSwarm.optimize (gp_opt.py:1136-1177)",
            "ms": three * 1e3, "value": units / three, "unit": "particles/s"}
    if args.config != 5:
        x = np.atleast_1d(last["x"])
        res["chosen_x"] = [float(v) for v in x]
        hit = np.flatnonzero(np.all(cfg["grid"] == x, axis=1))
        res["chosen_index"] = int(hit[0]) if hit.size else None
    if world == 1 and not args.no_cpu_baseline and args.config != 5:
        rows = None if args.cpu_rows is None else min(args.cpu_rows, units)
        base, parity = cpu_baseline(cfg, rows, dev_Q=opt.Q)
        res["cpu_baseline"] = base
        res["parity"] = parity
        res["speedup_vs_cpu"] = res["value"] / base["value"]      # (all host cores at work)
        res["speedup_vs_cpu_one_process"] = res["value"] / base["blas_threads"]["value"]
        if args.check_chosen:
            res["parity"].update(full_grid_check(cfg, opt, res.get("chosen_index")))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
