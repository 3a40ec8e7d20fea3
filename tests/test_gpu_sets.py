"""GPU parity, SURVEY.md 8(a) rows A3-A6 and next rows f1 / f4: confidence intervals, safe set, ``compute_sets``
(maximisers, candidates, the expander loop, ties), the arg-max, the incremental BO loop, the one-launch
step of small grids -- against reference-generated fixtures and the oracle."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal
from _golden import load, make_kernel

from _gpu_common import (  # noqa: F401
    MEAN_TOL, VAR_TOL, mods, smooth, kernels, check_posterior, product_kernel, GOLD, build_opt, _swarm_problem, _grow_reference, kernels_from, _PretendWorld, _PretendWorldPadded, _dev_script)

pytestmark = pytest.mark.gpu


def test_q_written_in_place_reaches_the_device(mods):
    """``opt.Q[...] = ...`` (the reference mutates ``Q`` in place) is uploaded before
    the next pass on the HIP backend and equals ``opt.Q = array``."""
    safeopt_amd, gpy, _, _ = mods
    rng = np.random.default_rng(11)
    X = rng.uniform(-2, 2, (25, 2)); Y = smooth(X, 3) + 0.4
    grid = safeopt_amd.linearly_spaced_combinations([(-3, 3)] * 2, 50)

    def make():
        k = gpy.kern.RBF(2, variance=1.5, lengthscale=[1.0, 1.3], ARD=True)
        opt = safeopt_amd.SafeOpt(gpy.models.GPRegression(X, Y, k, noise_var=0.01), grid, 0.0,
                                  threshold=0.1)
        opt.update_confidence_intervals()
        return opt
    a, b = make(), make()
    target = np.array(a.Q)
    target[100:400, 0] -= 0.3
    target[[7, 9], 1] += 2.0
    a.Q[100:400, 0] -= 0.3
    a.Q[[7, 9], 1] += 2.0
    b.Q = target
    a.compute_sets(); b.compute_sets()
    assert_array_equal(a.Q, target)
    for name in ("S", "M", "G"):
        assert_array_equal(getattr(a, name), getattr(b, name))
    assert_array_equal(a.get_new_query_point(), b.get_new_query_point())
    # (S / M / G are writable too since round 5: test_mask_writes_reach_the_device)


@pytest.mark.parametrize("n,layout", [(60, "aaa"), (300, "aab"), (200, "abba"), (25, "aa")])
def test_rank1_refresh_reuses_the_shared_factor_same_bits(mods, n, layout):
    """The outputs of a multi-output GP get their new observation at the same x*: c(x) of
    the closed-form rank-1 refresh (k_rank1) is the same for all of them and computed once
    when the factor is shared -- mean, var, Q and S must be the same bits as with every GP
    refreshed on its own, and agree with a sweep of the refitted GPs."""
    _, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(n + len(layout))
    d = 2
    Xa = rng.uniform(-2, 2, size=(n, d)); Xb = rng.uniform(-2, 2, size=(n, d))
    pts = rng.uniform(-3, 3, size=(3000, d))
    xn = rng.uniform(-1, 1, size=d)
    G = len(layout)
    fmin = np.full(G, 0.1)
    out = {}
    for on in (True, False):
        gps = []
        for i, c in enumerate(layout):
            X = Xa if c == "a" else Xb
            gps.append(gpy.models.GPRegression(X, smooth(X, 7 + i) + 0.3, kernels(gpy.kern, "Matern52", d),
                                               noise_var=0.05 ** 2))
        devs = [g._fitted() for g in gps]
        ctx = devs[0].ctx
        old = ctx.set_share(on)
        try:
            grid = _hip.DeviceGrid(ctx, pts, G)
            grid.confidence(devs, 2.0, fmin)
            for i, dv in enumerate(devs):
                assert dv.append(xn, 0.4 + 0.1 * i)
            ml = grid.rank1_update(devs, [1] * G, 2.0, fmin)
            out[on] = (ml, grid.download(_hip.Q), grid.download(_hip.S), grid.download(_hip.MEAN),
                       grid.download(_hip.VAR))
            if on:      # ... and against the sweep of the grown GPs
                ref = _hip.DeviceGrid(ctx, pts, G)
                ref.confidence(devs, 2.0, fmin)
                assert_allclose(out[on][3], ref.download(_hip.MEAN), rtol=0, atol=1e-9)
                assert_allclose(out[on][4], ref.download(_hip.VAR), rtol=0, atol=1e-9 * 1.7)
        finally:
            ctx.set_share(old)
    assert out[True][0] == out[False][0]
    for x, y in zip(out[True][1:], out[False][1:]):
        assert_array_equal(x, y)


@pytest.mark.parametrize("name", GOLD)
def test_replay_reference_golden(mods, name):
    """The product's SafeOpt reproduces what the reference's gp_opt.py did."""
    z, meta = load(name)
    for t in meta["recorded"]:
        opt = build_opt(mods, z, meta, t)
        ctx = z["it%d_context" % t] if meta["num_contexts"] else None
        x = opt.optimize(context=ctx, ucb=meta["ucb"])
        assert_allclose(opt.Q, z["it%d_Q" % t], rtol=0, atol=1e-8)
        assert_array_equal(opt.S, z["it%d_S" % t])
        if not meta["ucb"]:
            assert_array_equal(opt.M, z["it%d_M" % t])
            assert_array_equal(opt.G, z["it%d_G" % t])
        assert_array_equal(x, z["it%d_x_next" % t])
        mx, ml = opt.get_maximum(context=ctx)
        assert_array_equal(mx, z["it%d_max_x" % t])
        assert_allclose(ml, z["it%d_max_l" % t], atol=1e-8)


@pytest.mark.parametrize("name", ["sets_1d_seed0", "sets_1d_seed7", "sets_1d_g2_seed0",
                                  "sets_1d_g2_seed7", "sets_2d_seed3"])
def test_expander_loop_golden(mods, name):
    """Rank-1 expander test == the reference's add-point / re-predict loop,
    including a case where the 22nd candidate in width order is the first
    expander."""
    safeopt_amd, gpy, _, _ = mods
    z, meta = load(name)
    gps = [gpy.models.GPRegression(z["X%d" % i], z["Y%d" % i], make_kernel(gpy.kern, spec),
                                   noise_var=meta["noise_vars"][i])
           for i, spec in enumerate(meta["kernels"])]
    opt = safeopt_amd.SafeOpt(gps if len(gps) > 1 else gps[0], z["parameter_set"],
                              meta["fmin"] if len(gps) > 1 else meta["fmin"][0],
                              threshold=meta["threshold"])
    opt.update_confidence_intervals()
    opt.compute_sets()
    assert_allclose(opt.Q, z["Q"], rtol=0, atol=1e-8)
    assert_array_equal(opt.S, z["S"]); assert_array_equal(opt.M, z["M"])
    assert_array_equal(opt.G, z["G"])
    assert_array_equal(opt.get_new_query_point(), z["x_next"])


def test_full_sets_golden(mods):
    safeopt_amd, gpy, _, _ = mods
    z, meta = load("safeopt_full_sets")
    gp = gpy.models.GPRegression(z["X0"], z["Y0"], make_kernel(gpy.kern, meta["kernels"][0]),
                                 noise_var=meta["noise_vars"][0])
    opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"])
    opt.update_confidence_intervals()
    opt.compute_sets(full_sets=True)
    assert_array_equal(opt.S, z["S"]); assert_array_equal(opt.M, z["M"])
    assert_array_equal(opt.G, z["G"])


def test_sets_bit_exact_on_random_intervals(mods):
    """Set logic alone: upload arbitrary Q, compare S / M / candidate flow /
    arg-max with the NumPy restatement bit for bit (Lipschitz certifies the
    expanders so no GP arithmetic is involved)."""
    safeopt_amd, gpy, gpn, son = mods
    rng = np.random.default_rng(11)
    for trial in range(6):
        N = [1000, 4097, 300, 12345, 128, 77][trial]
        G = [1, 2, 3, 1, 2, 3][trial]
        grid = np.sort(rng.uniform(-5, 5, size=(N, 2)), axis=0)
        gps = [gpy.models.GPRegression(np.zeros((1, 2)), np.ones((1, 1)), gpy.kern.RBF(2),
                                       noise_var=0.01) for _ in range(G)]
        lo = rng.normal(0.2, 1.0, size=(N, G))
        wd = np.abs(rng.normal(0.5, 0.4, size=(N, G))) + 1e-3
        if trial % 2 == 0:            # force exact ties in values and widths
            lo = np.round(lo, 1); wd = np.round(wd, 1) + 0.1
        Q = np.empty((N, 2 * G)); Q[:, ::2] = lo; Q[:, 1::2] = lo + wd
        fmin = [0.0, -np.inf, 0.3][:G] if G > 1 else [0.0]
        scaling = [1.0, 2.0, 0.5][:G]
        lips = [0.8, 0.5, 1.1][:G]
        thr = 0.15
        opt = safeopt_amd.SafeOpt(gps if G > 1 else gps[0], grid, fmin if G > 1 else 0.0,
                                  lipschitz=lips if G > 1 else lips[0], threshold=thr,
                                  scaling=scaling)
        opt.Q = Q
        opt.compute_sets()
        S = son.safe_set(Q, fmin)
        assert_array_equal(opt.S, S)
        if not S.any():
            with pytest.raises(EnvironmentError):
                opt.get_new_query_point()
            continue
        So, Mo, Go, trace = son.compute_sets([None] * G, grid, Q, fmin, scaling, thr, 2.,
                                             lipschitz=np.asarray(lips), return_trace=True)
        assert_array_equal(opt.M, Mo)
        # exact ties included: the visiting order among equal widths is the one
        # of the reference's own argsort()[::-1] (run here by the oracle)
        assert_array_equal(opt.G, Go)
        idx = son.query_index(Q, So, Mo, Go, scaling)
        assert_array_equal(opt.get_new_query_point(), grid[idx])
        assert_array_equal(opt.get_new_query_point(ucb=True),
                           grid[son.query_index(Q, So, Mo, Go, scaling, ucb=True)])
        assert opt.G.sum() <= 1


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_tied_widths_golden(mods, seed):
    """Exact ties in the candidate widths (gp_opt.py:542-552): intervals assigned
    by hand (quantised), GP expander test.  The reference fixture pins which of
    the tied candidates ends up in G -- NumPy's argsort()[::-1] order, which the
    product reproduces by running that very expression when (and only when) a tie
    can matter.  (If this host's NumPy sorts ties differently from the one that
    wrote the fixture, the oracle run on THIS host is the reference.)"""
    safeopt_amd, gpy, gpn, son = mods
    z, meta = load("ties_1d_seed%d" % seed)
    assert int(z["n_tied_top"]) > 1
    gp = gpy.models.GPRegression(z["X0"], z["Y0"], make_kernel(gpy.kern, meta["kernels"][0]),
                                 noise_var=meta["noise_vars"][0])
    go = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                          noise_var=meta["noise_vars"][0])
    opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"])
    opt.Q = z["Q"]
    opt.compute_sets()
    x = opt.get_new_query_point()
    So, Mo, Go = son.compute_sets([go], z["parameter_set"], z["Q"], meta["fmin"],
                                  meta["scaling"], meta["threshold"], meta["beta"])
    assert_array_equal(opt.S, So); assert_array_equal(opt.M, Mo)
    assert_array_equal(opt.G, Go)
    assert_array_equal(x, z["parameter_set"][son.query_index(z["Q"], So, Mo, Go, meta["scaling"])])
    # which tied candidate argsort()[::-1] visits first is NumPy's choice: with the
    # NumPy that wrote the fixture the REFERENCE's own G / x_next must come out
    same_numpy = meta.get("numpy_version") == np.__version__
    if same_numpy or np.array_equal(Go, z["G"]):
        assert_array_equal(opt.G, z["G"]); assert_array_equal(x, z["x_next"])
        print("ties seed %d: asserted the REFERENCE's G / x_next (NumPy here %s, fixture %s)"
              % (seed, np.__version__, meta.get("numpy_version")))
    else:
        print("ties seed %d: NumPy here %s sorts ties unlike the fixture's %s: asserted the "
              "LOCAL oracle only" % (seed, np.__version__, meta.get("numpy_version")))
    # the whole step in one call gives the same sets (Q is recomputed: no ties then,
    # but the path through sets_fused with its tie count must still agree)
    assert_array_equal(opt.S, z["S"]); assert_array_equal(opt.M, z["M"])
    # the tie count the front half reports (it travels with the first candidate on N
    # ranks): candidates whose width equals the first one's bit for bit
    # ... counted by the front half ITSELF: the fused pass above (k_front_final)
    # leaves its own count in the same scratch word, so that word is overwritten
    # first (a top-16 query uses the slot) and the upload resets the sets
    be = opt._backend
    opt.Q = z["Q"]
    thr_beta = np.atleast_1d(np.asarray(meta["threshold"], dtype=float) * meta["beta"])
    be.maximizers(opt._max_l)
    be.candidates(0.0, opt.scaling, thr_beta, True)
    be.topk(0, np.inf, np.iinfo(np.int64).max, 16)
    opt.Q = z["Q"]
    out5, _x, _m, _q = be.sets_front(opt._max_l, None, opt.scaling, thr_beta)
    cand, width = be.candidate_widths()
    assert int(out5[5]) == int(np.sum(cand & (width == out5[3]))) == int(z["n_tied_top"])
    # the N-rank front half (in-stream scalars; here without a communicator)
    be.topk(0, np.inf, np.iinfo(np.int64).max, 16)
    opt.Q = z["Q"]
    out5c, _x, _m, _q, _ml = be.sets_front_comm(opt.scaling, thr_beta)
    assert_array_equal(out5c, out5)


def test_topk_order_and_ties(mods):
    """Visiting order: width descending, ties -> higher index first."""
    from safeopt_amd import _hip
    safeopt_amd, gpy, _, _ = mods
    N = 10000
    rng = np.random.default_rng(3)
    grid = rng.uniform(-1, 1, size=(N, 1))
    gp = gpy.models.GPRegression(np.zeros((1, 1)), np.ones((1, 1)), gpy.kern.RBF(1), noise_var=0.01)
    w = np.round(rng.uniform(0.1, 1.0, N), 2)          # many exact ties
    Q = np.stack([np.ones(N), 1.0 + w], axis=1)
    Q[0] = [5.0, 5.01]                                 # the single maximiser
    opt = safeopt_amd.SafeOpt(gp, grid, 0., threshold=0., scaling=[1.0])
    opt.Q = Q
    be = opt._backend
    be.maximizers(5.0)
    n_cand, _ = be.candidates(0.01, [1.0], [0.0], False)
    ref = np.lexsort((-np.arange(N), -w))              # w desc, index desc
    ref = ref[ref != 0]
    assert n_cand == ref.size
    cut = (np.inf, np.iinfo(np.int64).max)
    got = []
    for _ in range(5):
        ww, ii = be.topk(0, cut[0], cut[1], 16)
        got.extend(ii.tolist()); cut = (ww[-1], ii[-1])
    assert got == ref[:80].tolist()


def test_optimize_one_round_trip_paths(mods):
    """SafeOpt.optimize() enqueues sweep + set passes + probe + arg-max with one
    read-back; same answer as the step-by-step methods, and the reference's
    EnvironmentError when nothing is safe (gp_opt.py:631-632)."""
    safeopt_amd, gpy, _, _ = mods
    rng = np.random.default_rng(4)
    X = rng.uniform(-1, 1, size=(12, 2))
    Y = 1.0 + 0.3 * np.sin(3 * X[:, :1]) + 0.2 * X[:, 1:]
    grid = safeopt_amd.linearly_spaced_combinations([(-3, 3)] * 2, 60)

    def make(y):
        gp = gpy.models.GPRegression(X, y, gpy.kern.RBF(2, variance=2., lengthscale=1., ARD=True),
                                     noise_var=0.05 ** 2)
        return safeopt_amd.SafeOpt(gp, grid, 0., threshold=0.2)
    a, b = make(Y), make(Y)
    xa = a.optimize()
    b.update_confidence_intervals()
    b.compute_sets()
    xb = b.get_new_query_point()
    assert_array_equal(xa, xb)
    for name in "QSMG":
        assert_array_equal(getattr(a, name), getattr(b, name))
    assert a.S.any() and a.M.any()
    # nothing safe: every observation far below fmin
    c = make(Y - 5.0)
    with pytest.raises(EnvironmentError):
        c.optimize()
    assert not c.S.any() and not c.M.any() and not c.G.any()
    assert c.get_maximum() is None


def test_sample_gp_function_device_interpolant(mods, monkeypatch):
    """with the package's kernels the RKHS interpolant of
    sample_gp_function is the posterior mean of a device GP handle.  The prior
    draw is pinned to the reference's (the covariance bits differ between kernel
    implementations, and the SVD behind multivariate_normal amplifies that), the
    evaluations are compared with the reference run."""
    safeopt_amd, gpy, _, _ = mods
    z, meta = load("sample_gp_function")
    for tag in ("rbf1", "m52_2"):
        m = meta[tag]
        k = make_kernel(gpy.kern, m["kernel"])
        bounds = [tuple(b) for b in m["bounds"]]
        xq = z[tag + "_xq"]
        seen = {}

        def draw(mean, cov, _v=z[tag + "_output"], _s=seen):
            _s["cov"] = np.array(cov)
            return _v.copy()
        monkeypatch.setattr(np.random, "multivariate_normal", draw)
        for mean in (None, "mean"):
            mf = None if mean is None else (lambda x: 0.3 * x[:, :1] - 0.1)
            np.random.seed(m["seed"])
            f = safeopt_amd.sample_gp_function(k, bounds, m["noise_var"], m["num_samples"],
                                               interpolation="kernel", mean_function=mf)
            key = "%s_kernel_%s" % (tag, "mean" if mean else "nomean")
            assert_allclose(f.nodes, z[key + "_nodes"], rtol=0, atol=0)
            # the prior covariance handed to multivariate_normal (kernel.K(nodes) +
            # 1e-6 I, utilities.py:89-93), from the device kernel matrix, against
            # what the reference run handed over
            assert_allclose(seen["cov"], z[tag + "_cov"], rtol=1e-12, atol=1e-13)
            # jitter 1e-6 on a smooth prior: the interpolation weights are ~1e5, so
            # 1e-6 absolute is the conditioning, not the kernels
            assert_allclose(f(xq, noise=False), z[key + "_clean"], rtol=0, atol=2e-6)
            np.random.seed(5)
            noisy = f(xq)                      # one randn(25, 1) call, as in the reference
            np.random.seed(5)
            assert_allclose(noisy - f(xq, noise=False),
                            np.sqrt(m["noise_var"]) * np.random.randn(xq.shape[0], 1),
                            rtol=0, atol=1e-12)


@pytest.mark.parametrize("name,last", [("safeopt_1d_rbf", 19), ("safeopt_2d_rbf", 11),
                                       ("safeopt_1d_multi", 9), ("safeopt_2d_mat52_g3", 7)])
def test_bo_loop_with_rank1_updates_matches_reference(mods, name, last):
    """The whole sequential BO loop of the reference run (optimize -> measure ->
    add_new_data_point), with every posterior after the first obtained by the
    closed-form rank-1 update: same chosen parameter at every iteration."""
    safeopt_amd, gpy, _, _ = mods
    z, meta = load(name)
    G = len(meta["kernels"])
    gps = [gpy.models.GPRegression(z["it0_X%d" % i], z["it0_Y%d" % i], make_kernel(gpy.kern, spec),
                                   noise_var=meta["noise_vars"][i])
           for i, spec in enumerate(meta["kernels"])]
    opt = safeopt_amd.SafeOpt(gps if G > 1 else gps[0], z["parameter_set"],
                              meta["fmin"] if G > 1 else meta["fmin"][0], threshold=meta["threshold"])
    opt.small_step = False      # (grids this small take a full step in one launch otherwise)
    n0 = z["it0_X0"].shape[0]
    Yall = np.hstack([z["it%d_Y%d" % (last, i)] for i in range(G)])
    for t in range(last + 1):
        x = opt.optimize()
        assert_array_equal(x, z["x_next_all"][t]), t
        if t in meta["recorded"]:
            assert_allclose(opt.Q, z["it%d_Q" % t], rtol=0, atol=1e-8)
            assert_array_equal(opt.S, z["it%d_S" % t]); assert_array_equal(opt.M, z["it%d_M" % t])
            assert_array_equal(opt.G, z["it%d_G" % t])
        if t < last:
            opt.add_new_data_point(x, Yall[n0 + t][None, :])
    assert opt._backend._rank1_streak > 0          # the incremental path really ran
    # remove_last_data_point -> pop -> full sweep again
    opt.remove_last_data_point()
    assert_array_equal(opt.optimize(), z["x_next_all"][last - 1])


def test_warm_path_makes_no_device_allocations(mods):
    """Buffers grow on demand (hipMalloc + stream sync).  A steady-state loop --
    same data, same grid -- must not allocate at all, and a BO loop that appends
    one observation per iteration only when a capacity is exhausted (the factor
    is sized for 64+ appends, scratch grows geometrically)."""
    safeopt_amd, gpy, _, _ = mods
    from safeopt_amd import _hip
    ctx = _hip.Context.default()
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, size=(40, 2)); Y = smooth(X, 4) + 1.0
    gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(2, 2., 0.5, ARD=True), noise_var=1e-4)
    grid = safeopt_amd.linearly_spaced_combinations([(-1.5, 1.5)] * 2, 150)
    opt = safeopt_amd.SafeOpt(gp, grid, 0., threshold=0.1)
    for _ in range(2):
        opt.optimize()
    opt.get_maximum()
    base = ctx.alloc_count()
    for _ in range(4):
        opt.optimize()
        opt.get_maximum()
    assert ctx.alloc_count() == base
    grown = 0
    for t in range(30):                       # n = 40 -> 70: crosses 48 and 64
        x = opt.optimize()
        before = ctx.alloc_count()
        opt.add_new_data_point(x, float(smooth(x[None, :], 4)[0, 0]) + 1.0)
        opt.optimize()
        grown += ctx.alloc_count() - before
    assert grown <= 6, grown


@pytest.mark.parametrize("seed", range(24))
def test_one_launch_step_of_small_grids(mods, seed):
    """The reference's own regime -- grids of 1e3-1e4 rows, a few observations
    (gp_opt.py:651-675 as examples/1d_example.ipynb runs it; BASELINE.json config 1) --
    takes a whole ``optimize()`` in ONE launch (``sgp_grid_step_small``): against the
    large-grid path on the same data (``small_step = False``: sweep + nine set passes) the
    chosen parameter and ``S / M / G`` must be identical and ``Q`` the same bits (one copy
    of the posterior arithmetic, tiny_row.h), over several BO iterations, and against the
    oracle."""
    safeopt_amd, gpy, gpn, son = mods
    rng = np.random.default_rng(9100 + seed)
    d = int(rng.integers(1, 4))
    G = int(rng.integers(1, 4))
    n = int(rng.integers(1, 49))
    sides = {1: [int(rng.integers(50, 4000))], 2: [int(rng.integers(8, 120)), int(rng.integers(8, 120))],
             3: [int(rng.integers(5, 26)) for _ in range(3)]}[d]
    kind = ["RBF", "Matern32", "Matern52"][int(rng.integers(0, 3))]
    grid = safeopt_amd.linearly_spaced_combinations([(-4., 4.)] * d, sides)
    X = rng.uniform(-1.5, 1.5, size=(n, d))
    Ys = [smooth(X, 70 + g) - smooth(X, 70 + g).min() + 0.3 for g in range(G)]
    ls = list(rng.uniform(0.6, 1.6, size=d))
    fmin = [0.0 if (g == 0 or rng.random() < 0.7) else -np.inf for g in range(G)]

    def build(ns):
        reg = ns.models.GPRegression if hasattr(ns, "models") else ns.GPRegression
        kns = ns.kern if hasattr(ns, "kern") else ns
        return [reg(X, Ys[g], getattr(kns, kind)(d, 2.0, ls, ARD=True), noise_var=0.05 ** 2)
                for g in range(G)]
    # (an INTEGER fmin in half of the single-GP cases: SafeOpt(gp, grid, 0) is how the
    # reference's examples pass it)
    f1 = int(fmin[0]) if seed % 2 == 0 else fmin[0]
    a = safeopt_amd.SafeOpt(build(gpy) if G > 1 else build(gpy)[0], grid,
                            fmin if G > 1 else f1, threshold=0.2)
    b = safeopt_amd.SafeOpt(build(gpy) if G > 1 else build(gpy)[0], grid,
                            fmin if G > 1 else f1, threshold=0.2)
    a._backend.SMALL_STEP_BUDGET = 10 ** 9   # (the one-launch step whatever it costs)
    b.small_step = False
    b._backend.incremental = False     # (a full sweep every step, like the one-launch step)
    ctx = a._backend.ctx
    for it in range(4):
        try:
            xa = a.optimize()
        except EnvironmentError:
            # no safe row (gp_opt.py:632): the large-grid path must say the same
            assert ctx.last_sweep() == "step-small"
            with pytest.raises(EnvironmentError):
                b.optimize()
            assert not a.S.any() and not a.M.any() and not a.G.any()
            assert_array_equal(a.Q, b.Q)
            break
        assert ctx.last_sweep() == "step-small"
        xb = b.optimize()
        assert ctx.last_sweep() != "step-small"
        assert_array_equal(xa, xb)
        assert_array_equal(a.S, b.S)
        assert_array_equal(a.M, b.M)
        assert_array_equal(a.G, b.G)
        assert_array_equal(a.Q, b.Q)           # the same bits
        if it == 0:
            go = build(gpn)
            scaling = np.array([np.sqrt(g.kern.Kdiag(np.zeros((1, d)))[0]) for g in go])
            idx, Qo, So, Mo, Go = son.optimize_grid(go, grid, np.asarray(fmin, dtype=float),
                                                    scaling, 0.2, 2.0)
            assert_array_equal(a.S, So)
            assert_array_equal(a.M, Mo)
            assert_array_equal(a.G, Go)
            assert_array_equal(xa, grid[idx])
            assert np.max(np.abs(a.Q - Qo)) < 1e-8
        if a.gps[0].X.shape[0] >= 48:
            break
        y = np.array([float(smooth(xa[None, :], 70 + g)[0, 0]) + 0.3 for g in range(G)])
        a.add_new_data_point(xa, y)
        b.add_new_data_point(xb, y)


def test_mask_writes_reach_the_device(mods):
    """``opt.S / M / G`` are live arrays in the reference (gp_opt.py:481, 505-506, 511, 615);
    here an element-wise write into the host mirror is uploaded (``sgp_grid_upload_mask``)
    before the next ``get_new_query_point``, whose arg-max runs over the EDITED ``M | G``
    (gp_opt.py:635-649); ``compute_sets`` recomputes all three, ``S`` from the intervals."""
    safeopt_amd, gpy, gpn, son = mods
    rng = np.random.default_rng(12)
    X = rng.uniform(-2, 2, size=(60, 2))
    Y = smooth(X, 9) - smooth(X, 9).min() + 0.5
    grid = safeopt_amd.linearly_spaced_combinations([(-4., 4.)] * 2, [150, 140])
    gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(2, 2.0, [1.0, 1.2], ARD=True), noise_var=0.05 ** 2)
    opt = safeopt_amd.SafeOpt(gp, grid, 0.0, threshold=0.2)
    x0 = opt.optimize()
    S0, M0, G0 = np.array(opt.S), np.array(opt.M), np.array(opt.G)
    Q = np.array(opt.Q)
    rows = np.flatnonzero(M0 | G0)
    keep = rows[~np.all(grid[rows] == x0, axis=1)][::3]
    opt.G[:] = False
    opt.M[:] = False
    opt.M[keep] = True
    x1 = opt.get_new_query_point()
    val = (Q[:, 1] - Q[:, 0]) / opt.scaling[0]
    assert_array_equal(x1, grid[keep[np.argmax(val[keep])]])
    assert_array_equal(opt._backend.download(safeopt_amd._hip.M).astype(bool), np.isin(np.arange(len(grid)), keep))
    opt.S[:] = False
    with pytest.raises(EnvironmentError):
        opt.get_new_query_point()
    opt.compute_sets()
    assert_array_equal(opt.S, S0); assert_array_equal(opt.M, M0); assert_array_equal(opt.G, G0)
    assert_array_equal(opt.get_new_query_point(), x0)


def test_mask_edits_are_dropped_by_the_one_launch_step(mods):
    """``optimize()`` recomputes all three sets (gp_opt.py:651-675 -> 478-481, 505-615): an
    element-wise edit of ``opt.M / G / S`` made before it must not survive it -- on the small
    grids of the reference's own examples the whole step is ONE launch (``k_step_small``),
    whose driver has to drop the pending edits just as ``compute_sets`` does on large grids."""
    safeopt_amd, gpy, gpn, son = mods
    rng = np.random.default_rng(5)
    X = rng.uniform(-3, 3, size=(12, 1))
    Y = smooth(X, 3) - smooth(X, 3).min() + 0.6

    def make():
        gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(1, 2.0, 1.0), noise_var=0.05 ** 2)
        return safeopt_amd.SafeOpt(gp, safeopt_amd.linearly_spaced_combinations([(-5., 5.)], 1000),
                                   0.0, threshold=0.2)
    fresh = make()
    x_ref = fresh.optimize()
    assert fresh._backend.ctx.last_sweep() == "step-small"
    for field in ("M", "G", "S"):
        opt = make()
        opt.optimize()
        getattr(opt, field)[:] = False
        x = opt.optimize()
        assert opt._backend.ctx.last_sweep() == "step-small"
        assert_array_equal(x, x_ref)
        for f in ("S", "M", "G"):
            assert_array_equal(getattr(opt, f), getattr(fresh, f))
        assert_array_equal(opt.get_new_query_point(), x_ref)
