"""The N-rank product on REAL kernels in REAL processes, on the one GPU of the box.

N operating-system processes, each with its own HIP context on device 0 and its
TRUE contiguous shard of the grid, exchange their merged values through
``safeopt_amd.dist.SocketComm`` (the interface of ``RcclComm``, host side: RCCL wants
one GPU per rank).  Every rank must arrive at what the unsharded run gives -- the
reference's golden vectors where there are any (``gp_opt.py:453-649``), a one-process
run of the same product otherwise: ``Q / S / M / G`` gathered over the ranks, the
chosen parameter, ``get_maximum``.

What this executes for the first time with genuinely foreign data: ``shard_range`` /
the global offset of every kernel, ``sets_front`` / ``sets_back`` merged over ranks
(first candidate of ANOTHER rank as the operand of the expander scan, counts and
flags summed / or-ed over ranks), ``_settle_ties`` over gathered widths, the uneven
all-gather of ``_mirror``, a shard without any safe row, the factor tables of a
tensor grid on shards that start in the middle of a grid row.
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _smooth(x, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-3, 3, size=(10, x.shape[1]))
    w = rng.normal(size=10)
    r2 = ((x[:, None, :] - c[None]) ** 2).sum(-1)
    return (np.exp(-0.25 * r2) * w).sum(1)[:, None]


def _golden_cases(sa, gpy, comm, report):
    """Recorded iterations of the reference (tests/golden) through the sharded driver."""
    from _golden import load, make_kernel
    names = ["safeopt_1d_rbf", "safeopt_2d_rbf", "safeopt_1d_multi", "safeopt_2d_mat52_g3",
             "safeopt_1d_lipschitz", "safeopt_context", "safeopt_2d_ucb"]
    for name in names:
        z, meta = load(name)
        for t in meta["recorded"]:
            gps = [gpy.models.GPRegression(z["it%d_X%d" % (t, i)], z["it%d_Y%d" % (t, i)],
                                           make_kernel(gpy.kern, spec),
                                           noise_var=meta["noise_vars"][i])
                   for i, spec in enumerate(meta["kernels"])]
            lip = meta["lipschitz"]
            if lip is not None and len(lip) == 1:
                lip = lip[0]
            opt = sa.SafeOpt(gps if len(gps) > 1 else gps[0], z["parameter_set"],
                             meta["fmin"] if len(gps) > 1 else meta["fmin"][0],
                             lipschitz=lip, beta=float(z["beta_all"][t]),
                             threshold=meta["threshold"], num_contexts=meta["num_contexts"],
                             comm=comm)
            lo, hi = opt._shard
            assert hi - lo < z["parameter_set"].shape[0]        # really sharded
            ctx = z["it%d_context" % t] if meta["num_contexts"] else None
            x = opt.optimize(context=ctx, ucb=meta["ucb"])
            ok = (np.array_equal(x, z["it%d_x_next" % t]) and
                  np.array_equal(opt.S, z["it%d_S" % t]) and
                  np.allclose(opt.Q, z["it%d_Q" % t], atol=1e-8, rtol=0))
            if not meta["ucb"]:
                ok = ok and np.array_equal(opt.M, z["it%d_M" % t]) and \
                    np.array_equal(opt.G, z["it%d_G" % t])
            mx, ml = opt.get_maximum(context=ctx)
            ok = ok and np.array_equal(mx, z["it%d_max_x" % t]) and \
                abs(ml - z["it%d_max_l" % t]) < 1e-8
            # a shard without a safe row takes part like any other
            empty = not bool(np.any(z["it%d_S" % t][lo:hi]))
            report.append(("golden %s it%d%s" % (name, t, " (no safe row here)" if empty else ""),
                           bool(ok)))


def _tie_cases(sa, gpy, comm, report):
    """Exact ties in the candidate widths (gp_opt.py:542-552) across REAL ranks."""
    from _golden import load, make_kernel
    from oracle import gp_numpy as gpn, safeopt_numpy as son
    for seed in range(6):
        z, meta = load("ties_1d_seed%d" % seed)
        gp = gpy.models.GPRegression(z["X0"], z["Y0"], make_kernel(gpy.kern, meta["kernels"][0]),
                                     noise_var=meta["noise_vars"][0])
        go = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                              noise_var=meta["noise_vars"][0])
        opt = sa.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"], comm=comm)
        opt.Q = z["Q"]
        opt.compute_sets()
        x = opt.get_new_query_point()
        So, Mo, Go = son.compute_sets([go], z["parameter_set"], z["Q"], meta["fmin"],
                                      meta["scaling"], meta["threshold"], meta["beta"])
        ok = (np.array_equal(opt.S, So) and np.array_equal(opt.M, Mo) and
              np.array_equal(opt.G, Go) and
              np.array_equal(x, z["parameter_set"][son.query_index(z["Q"], So, Mo, Go,
                                                                  meta["scaling"])]))
        same_numpy = meta.get("numpy_version") == np.__version__
        if same_numpy or np.array_equal(Go, z["G"]):
            ok = ok and np.array_equal(opt.G, z["G"]) and np.array_equal(x, z["x_next"])
            branch = "reference G"
        else:
            branch = "local oracle only (NumPy %s, fixture %s)" % (np.__version__,
                                                                  meta.get("numpy_version"))
        report.append(("ties seed %d [%s]" % (seed, branch), bool(ok)))


def _loop_cases(sa, gpy, comm, report):
    """Whole BO loops on twins of BASELINE.json's configs 3 and 4 against the SAME
    product on one rank: every iteration's query point, then the final sets."""
    from safeopt_amd import dist
    cases = [
        # name, kind, d, G, n, sides, iterations
        ("config-3 twin (Matern52, 3 GPs, n = 120, 61 x 47)", "Matern52", 2, 3, 120, [61, 47], 4),
        ("config-3 twin with n = 300 (paired-wave sweep)", "Matern52", 2, 2, 300, [40, 33], 3),
        ("config-4 twin (RBF, 30^3, n = 200: factor tables)", "RBF", 3, 1, 200, [30, 30, 30], 4),
        ("config-4 twin (RBF, 43 x 37 x 41, n = 400)", "RBF", 3, 1, 400, [43, 37, 41], 3),
    ]
    for name, kind, d, G, n, sides, iters in cases:
        rng = np.random.default_rng(n + 13 * d)
        grid = sa.linearly_spaced_combinations([(-4., 4.)] * d, sides)
        X = rng.uniform(-1.5, 1.5, size=(n, d))
        Ys = [_smooth(X, 50 + g) - _smooth(X, 50 + g).min() + 0.5 for g in range(G)]

        def build(cm):
            gps = [gpy.models.GPRegression(X, Ys[g], getattr(gpy.kern, kind)(
                d, 2., list(np.linspace(0.9, 1.3, d)), ARD=True), noise_var=0.05 ** 2)
                for g in range(G)]
            return sa.SafeOpt(gps if G > 1 else gps[0], grid, [0.] * G if G > 1 else 0.,
                              threshold=0.2, comm=cm)
        a, b = build(comm), build(dist.LocalComm())
        ok = True
        for it in range(iters):
            xa, xb = a.optimize(), b.optimize()
            ok = ok and np.array_equal(xa, xb)
            y = np.array([float(_smooth(xb[None, :], 50 + g)[0, 0]) + 0.4 for g in range(G)])
            a.add_new_data_point(xa, y); b.add_new_data_point(xb, y)
        xa, xb = a.optimize(), b.optimize()
        ok = (ok and np.array_equal(xa, xb) and np.array_equal(a.S, b.S) and
              np.array_equal(a.M, b.M) and np.array_equal(a.G, b.G) and
              np.allclose(a.Q, b.Q, rtol=0, atol=1e-9))
        ma, mb = a.get_maximum(), b.get_maximum()
        ok = ok and np.array_equal(ma[0], mb[0])
        report.append((name, bool(ok)))


def _big_pass_cases(sa, gpy, comm, report):
    """The expander loop where it goes far (tests/test_gpu_expander_passes.py has the one-rank
    form against the oracle): a converged run -- every candidate visited, none marked --, a state
    whose first expander sits far down the visiting order, and ``full_sets``, on true shards
    against the SAME product on one rank.  The ranks agree on a threshold from their summed
    histograms, gather the candidates, test all of them against their own unsafe rows and or
    the flags (``SafeOpt._visit_in_big_passes_nrank``)."""
    import _scenarios as sc
    from safeopt_amd import dist, gp_opt
    calls = {"n": 0}
    orig = gp_opt._HipGridBackend.pass_test

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)
    gp_opt._HipGridBackend.pass_test = counted
    try:
        state = dict(r0=2.0, rings=8, ls=0.4, dmid=0.45, plateau=0.6)
        for name, kw, margin in (("converged run (no expander)", state, 0.05),
                                 ("first expander far down the order", dict(state, plateau=0.45), None)):
            data = sc.rim_data(160, **kw)
            grid = data["grid"]
            if margin is not None:
                gp0 = sc.make_gp(gpy, data)
                grid = np.ascontiguousarray(grid[sc.converged_rows(gp0, grid, margin)])

            def build(cm):
                o = sa.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, threshold=0.1, comm=cm)
                o.pass_sizes = (64, 512)
                return o
            a, b = build(comm), build(dist.LocalComm())
            before = calls["n"]
            xa, xb = a.optimize(), b.optimize()
            ok = (np.array_equal(xa, xb) and np.array_equal(a.S, b.S) and np.array_equal(a.M, b.M)
                  and np.array_equal(a.G, b.G))
            report.append(("big passes, %s: |G| = %d, %d N-rank passes" % (
                name, int(np.sum(b.G)), calls["n"] - before), bool(ok) and calls["n"] > before))
        # full_sets: every safe row is a candidate, every expander is marked
        data = sc.rim_data(100, **dict(state, plateau=0.45))

        def build(cm):
            o = sa.SafeOpt(sc.make_gp(gpy, data), data["grid"], 0.0, threshold=0.1, comm=cm)
            o.pass_sizes = (200, 1000)
            o.update_confidence_intervals()
            o.compute_sets(full_sets=True)
            return o
        before = calls["n"]
        a, b = build(comm), build(dist.LocalComm())
        report.append(("big passes, full_sets: |G| = %d, %d N-rank passes" % (
            int(np.sum(b.G)), calls["n"] - before),
            bool(np.array_equal(a.G, b.G)) and int(np.sum(b.G)) > 0 and calls["n"] > before))
        # Lipschitz certificates (gp_opt.py:558-576) on true shards: two constants (the candidates
        # of this state need L <= 0.6 .. 0.75 to reach an unsafe row) and full_sets -- against the
        # same product on one rank
        lcalls = {"n": 0}
        lorig = gp_opt._HipGridBackend.pass_lipschitz_test

        def lcounted(self, *a, **k):
            lcalls["n"] += 1
            return lorig(self, *a, **k)
        gp_opt._HipGridBackend.pass_lipschitz_test = lcounted
        try:
            data = sc.rim_data(160, **state)
            # (L = 0.72: the first expander is behind the first 16 candidates on this grid)
            for name, L, full in (("L = 1", 1.0, False), ("L = 0.72", 0.72, False),
                                  ("L = 0.7", 0.7, False), ("L = 1, full_sets", 1.0, True)):
                def build(cm):
                    o = sa.SafeOpt(sc.make_gp(gpy, data), data["grid"], 0.0, lipschitz=L,
                                   threshold=0.1, comm=cm)
                    o.pass_sizes = (64, 512)
                    if full:
                        o.update_confidence_intervals()
                        o.compute_sets(full_sets=True)
                    else:
                        o.optimize()
                    return o
                before = lcalls["n"]
                a, b = build(comm), build(dist.LocalComm())
                ok = (np.array_equal(a.S, b.S) and np.array_equal(a.M, b.M)
                      and np.array_equal(a.G, b.G))
                report.append(("big passes, Lipschitz certificates, %s: |G| = %d, %d N-rank passes" % (
                    name, int(np.sum(b.G)), lcalls["n"] - before),
                    bool(ok) and (lcalls["n"] > before or L == 0.7)))
        finally:
            gp_opt._HipGridBackend.pass_lipschitz_test = lorig
    finally:
        gp_opt._HipGridBackend.pass_test = orig


def _worker(rank, world, port, q, variant):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          SAFEOPT_COMM="socket", SAFEOPT_HIP_DEVICE="0",
                          SAFEOPT_SOCKET_IN_STREAM="1" if variant == "fused_comm" else "0")
        sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
        import safeopt_amd as sa
        import safeopt_amd.gpy as gpy
        from safeopt_amd import dist, gp_opt
        ctx, comm = dist.init_from_env()
        assert isinstance(comm, dist.SocketComm) and comm.world == world
        assert comm.in_stream == (variant == "fused_comm")
        # which variant of the N-rank step runs: count the entry points
        calls = {"sets_fused_comm": 0, "sets_front_comm": 0, "sets_back": 0}
        for name in calls:
            def counted(self, *a, _f=getattr(gp_opt._HipGridBackend, name), _n=name, **k):
                calls[_n] += 1
                return _f(self, *a, **k)
            setattr(gp_opt._HipGridBackend, name, counted)
        report = []
        _golden_cases(sa, gpy, comm, report)
        _tie_cases(sa, gpy, comm, report)
        _loop_cases(sa, gpy, comm, report)
        _big_pass_cases(sa, gpy, comm, report)
        if variant == "fused_comm":
            # the one-round-trip step (k_merge_front, flag all-reduce, k_merge_argmax behind
            # the staged collectives) is what ran, on every certified step
            # (sets_back remains for steps on hand-assigned intervals: the tie fixtures)
            report.append(("in-stream step taken (%d x sgp_grid_sets_fused_comm, %d x "
                           "sets_back)" % (calls["sets_fused_comm"], calls["sets_back"]),
                           calls["sets_fused_comm"] >= 20 and calls["sets_back"] <= 6))
        else:
            report.append(("host-side step taken (%d x sets_back, %d x fused_comm)"
                           % (calls["sets_back"], calls["sets_fused_comm"]),
                           calls["sets_fused_comm"] == 0 and calls["sets_back"] > 0))
        comm.barrier()
        comm.close()
        q.put((rank, report, None))
    except Exception:
        import traceback
        q.put((rank, [], traceback.format_exc()))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("variant", ["host", "fused_comm"])
@pytest.mark.parametrize("world", [2, 4])
def test_real_processes_true_shards_one_gpu(hip_device, world, variant):
    """``variant``: which N-rank step the ranks take -- ``host``: three round trips with
    the packed collectives on the host side (``sets_front`` / ``sets_back``);
    ``fused_comm``: the product default on RCCL, ``sgp_grid_sets_fused_comm`` -- one round
    trip, the merges on the device (``k_merge_front``, the int32 flag all-reduce,
    ``k_merge_argmax``) behind the collectives, here staged through the host over TCP
    (``sgp_comm_init_host``) because RCCL wants one GPU per rank."""
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    port = _free_port()
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q, variant)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=850) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    print("NumPy on this box:", np.__version__)
    for rank, report, err in sorted(results):
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        for what, ok in report:
            print("rank %d  %-72s %s" % (rank, what, "ok" if ok else "MISMATCH"))
        assert report and all(ok for _, ok in report), [w for w, ok in report if not ok]
    # every rank ran the same cases
    assert len({tuple(w for w, _ in rep) for _, rep, _ in
                [(r, [(w.split(" (no safe")[0].split(" (")[0], o) for w, o in rep], e)
                 for r, rep, e in results]}) == 1
