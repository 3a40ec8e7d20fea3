"""GPU parity, SURVEY.md 8(a) rows A0-A2: kernel matrices, the factorisation, ``predict_noiseless`` through
every posterior-sweep kernel -- HIP path against the oracle on the same seeded inputs (split out of
test_gpu_parity.py in round 6: one failure under ``-x`` hides only its own row)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal
from _golden import load, make_kernel

from _gpu_common import (  # noqa: F401
    MEAN_TOL, VAR_TOL, mods, smooth, kernels, check_posterior, product_kernel, GOLD, build_opt, _swarm_problem, _grow_reference, kernels_from, _PretendWorld, _PretendWorldPadded, _dev_script)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["RBF", "Matern32", "Matern52"])
def test_kern_K(mods, kind):
    _, gpy, gpn, _ = mods
    rng = np.random.default_rng(0)
    for d in (1, 2, 3, 5, 8):
        X = rng.normal(size=(37, d)); X2 = rng.normal(size=(53, d))
        k, ko = kernels(gpy.kern, kind, d), kernels(gpn, kind, d)
        assert_allclose(k.K(X, X2), ko.K(X, X2), rtol=1e-12, atol=1e-14)
        assert_allclose(k.K(X), ko.K(X), rtol=1e-12, atol=1e-14)
    # non-ARD, product on disjoint columns, Kdiag
    k = gpy.kern.Matern52(2, 3., 0.7); ko = gpn.Matern52(2, 3., 0.7)
    assert_allclose(k.K(X[:, :2], X2[:, :2]), ko.K(X[:, :2], X2[:, :2]), rtol=1e-12)
    kp = gpy.kern.RBF(1, 2., 1., active_dims=[0]) * \
        gpy.kern.Matern32(1, 1.5, 0.6, active_dims=[1], name='context')
    kpo = gpn.RBF(1, 2., 1., active_dims=[0]) * \
        gpn.Matern32(1, 1.5, 0.6, active_dims=[1], name='context')
    assert_allclose(kp.K(X[:, :2], X2[:, :2]), kpo.K(X[:, :2], X2[:, :2]), rtol=1e-12)
    assert_allclose(kp.Kdiag(X[:, :2]), kpo.Kdiag(X[:, :2]))
    assert kp.context.variance[0] == 1.5


@pytest.mark.parametrize("n", [1, 5, 16, 31, 33, 64, 200, 500])
def test_factor_matches_lapack(mods, n):
    _, gpy, gpn, _ = mods
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, size=(n, 2)); Y = smooth(X, 1)
    k, ko = kernels(gpy.kern, "RBF", 2), kernels(gpn, "RBF", 2)
    gp = gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
    go = gpn.GPRegression(X, Y, ko, noise_var=0.05 ** 2)
    Linv, alpha = gp._fitted().factor()
    Linv_ref = np.linalg.inv(go.L)
    assert np.max(np.abs(Linv - Linv_ref)) / np.max(np.abs(Linv_ref)) < 1e-9
    assert np.max(np.abs(alpha - go.woodbury_vector.ravel())) / \
        np.max(np.abs(go.woodbury_vector)) < 1e-8
    assert np.all(np.triu(Linv, 1) == 0)


@pytest.mark.parametrize("kind", ["RBF", "Matern32", "Matern52"])
@pytest.mark.parametrize("n,d", [(1, 1), (7, 1), (16, 2), (17, 2), (200, 2),
                                 (300, 3), (520, 4), (40, 6),
                                 # accumulator-chunk boundaries (256 rows), many
                                 # chunks, and every input dimension up to 8
                                 (256, 2), (257, 2), (1040, 3), (100, 5),
                                 (100, 7), (33, 8),
                                 # last row block with 5..8 rows, also right behind
                                 # a chunk boundary and with 16 k row blocks
                                 (5, 1), (24, 2), (248, 2), (264, 2), (277, 3)])
def test_predict_noiseless(mods, kind, n, d):
    _, gpy, gpn, _ = mods
    rng = np.random.default_rng(100 * n + d)
    X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 2)
    gp = gpy.models.GPRegression(X, Y, kernels(gpy.kern, kind, d), noise_var=0.05 ** 2)
    go = gpn.GPRegression(X, Y, kernels(gpn, kind, d), noise_var=0.05 ** 2)
    for N in (1, 129, 1000):
        Xs = rng.uniform(-3, 3, size=(N, d))
        m, v = gp.predict_noiseless(Xs)
        mo, vo = go.predict_noiseless(Xs)
        assert m.shape == (N, 1) and v.shape == (N, 1)
        check_posterior(m, v, mo, vo, 1.7)
        # F-ordered input (what linearly_spaced_combinations returns)
        m2, v2 = gp.predict_noiseless(np.asfortranarray(Xs))
        assert_array_equal(m, m2); assert_array_equal(v, v2)
    assert v.min() >= 1e-15


# The whole-grid sweep (SafeOpt.update_confidence_intervals, gp_opt.py:453-481)
# through both sweep kernels: the 4-wave kernel (csrc/sweep.hip) and the
# paired-wave kernel (csrc/sweep_pair.hip), each FORCED on every shape -- one
# j-block, ragged tiles, accumulator-chunk boundaries of both (256 / 512 rows),
# narrow last row blocks (n = 16 k + 1..12: one to three 4-row groups), up to 8 GPs (more than the 6 whose Q
# rows are staged in LDS), GPs of different sizes in one launch, d up to 8.


SWEEP_CASES = [
    # kind, d, [n per GP], N
    ("RBF", 1, [1], 70), ("Matern52", 2, [17], 64), ("RBF", 2, [16, 3], 129),
    ("Matern32", 2, [200], 1000), ("RBF", 2, [255, 257], 777),
    ("Matern52", 2, [500, 500, 500], 2000), ("RBF", 3, [512], 640),
    ("RBF", 3, [513, 40], 999), ("Matern52", 2, [529], 1111),
    ("RBF", 3, [1000], 1500), ("Matern32", 4, [1040, 100], 700),
    ("RBF", 4, [2000, 2000], 300), ("RBF", 2, [33] * 8, 500),
    ("Matern52", 5, [300] * 7, 321), ("RBF", 8, [130, 290], 450),
    ("Matern32", 6, [600], 200), ("RBF", 7, [64, 1, 270], 260),
    # last row blocks of 6 / 11 / 9 rows: two and three narrow groups (sweep.hip)
    ("Matern32", 2, [22, 43, 201], 333),
]


@pytest.mark.parametrize("which", ["classic", "pair"])
@pytest.mark.parametrize("kind,d,ns,N", SWEEP_CASES)
def test_grid_sweep_both_kernels(mods, which, kind, d, ns, N):
    _, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(sum(ns) + 17 * d + N)
    gps, gos = [], []
    for i, n in enumerate(ns):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 5 + i) + 0.3
        gps.append(gpy.models.GPRegression(X, Y, kernels(gpy.kern, kind, d), noise_var=0.05 ** 2))
        gos.append(gpn.GPRegression(X, Y, kernels(gpn, kind, d), noise_var=0.05 ** 2))
    pts = rng.uniform(-3, 3, size=(N, d))
    G = len(ns)
    fmin = np.where(np.arange(G) % 3 == 2, -np.inf, 0.1)
    ctx = gps[0]._fitted().ctx
    old = ctx.set_sweep(which)
    try:
        grid = _hip.DeviceGrid(ctx, pts, G)
        max_l, any_safe = grid.confidence([g._fitted() for g in gps], 2.0, fmin)
        Q = grid.download(_hip.Q); S = grid.download(_hip.S)
        mean = grid.download(_hip.MEAN); var = grid.download(_hip.VAR)
    finally:
        ctx.set_sweep(old)
    Qo = np.empty((N, 2 * G))
    for i, go in enumerate(gos):
        mo, vo = go.predict_noiseless(pts)
        check_posterior(mean[i][:, None], var[i][:, None], mo, vo, 1.7)
        sd = np.sqrt(vo[:, 0])
        Qo[:, 2 * i] = mo[:, 0] - 2.0 * sd; Qo[:, 2 * i + 1] = mo[:, 0] + 2.0 * sd
    assert_allclose(Q, Qo, rtol=0, atol=2e-8)
    # Q is exactly what mean / var give (same arithmetic as the reference line)
    for i in range(G):
        sd = np.sqrt(var[i])
        assert_array_equal(Q[:, 2 * i], mean[i] - 2.0 * sd)
        assert_array_equal(Q[:, 2 * i + 1], mean[i] + 2.0 * sd)
    So = np.all(Q[:, ::2] > fmin, axis=1)
    assert_array_equal(S, So)
    assert any_safe == bool(So.any())
    if So.any():
        assert max_l == Q[So, 0].max()


@pytest.mark.parametrize("which", ["RBF", "Matern32", "Matern52", "RBF*Matern52", "Matern32*RBF*Matern52"])
def test_device_gp_against_sklearn(mods, which):
    """The product's GP handle (safeopt_amd/gpy.py -> C ABI -> device) against
    scikit-learn's GaussianProcessRegressor -- a third implementation that shares
    nothing with oracle/gp_numpy.py (whose kernel-parameter plumbing resembles
    gpy.py's): ARD lengthscales, every kind, products of parts on the same columns."""
    _, gpy, _, _ = mods
    skgp = pytest.importorskip("sklearn.gaussian_process")
    from sklearn.gaussian_process import kernels as skk
    rng = np.random.default_rng(len(which))
    n, d = 70, 3
    X = rng.uniform(-2, 2, (n, d))
    Y = np.sin(X.sum(1))[:, None] + 0.1 * rng.normal(size=(n, 1))
    Xs = rng.uniform(-3, 3, (400, d))
    noise = 0.04
    k, ks, vtot = None, None, 1.0
    for i, kind in enumerate(which.split("*")):
        ls = rng.uniform(0.6, 2.0, size=d)
        var = float(rng.uniform(0.7, 1.6))
        vtot *= var
        part = getattr(gpy.kern, kind)(d, variance=var, lengthscale=ls, ARD=True)
        spart = (skk.RBF(ls, "fixed") if kind == "RBF" else
                 skk.Matern(ls, "fixed", nu=1.5 if kind == "Matern32" else 2.5))
        k = part if k is None else k * part
        ks = spart if ks is None else ks * spart
    gpr = skgp.GaussianProcessRegressor(skk.ConstantKernel(vtot, "fixed") * ks,
                                        alpha=noise + 1e-8, optimizer=None).fit(X, Y)
    mu, std = gpr.predict(Xs, return_std=True)
    gp = gpy.models.GPRegression(X, Y, k, noise_var=noise)
    m, v = gp.predict_noiseless(Xs)
    assert_allclose(m.ravel(), mu.ravel(), rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(v.ravel() - std ** 2)) / vtot < 1e-8
    # ... and through the grid sweep (both kernels)
    from safeopt_amd import _hip
    dev = gp._fitted()
    for name in ("classic", "pair"):
        old = dev.ctx.set_sweep(name)
        try:
            grid = _hip.DeviceGrid(dev.ctx, Xs, 1)
            grid.confidence([dev], 2.0, np.zeros(1))
            mean = grid.download(_hip.MEAN)[0]; var = grid.download(_hip.VAR)[0]
        finally:
            dev.ctx.set_sweep(old)
        assert_allclose(mean, mu.ravel(), rtol=1e-8, atol=1e-10)
        assert np.max(np.abs(var - std ** 2)) / vtot < 1e-8


def test_hyperparameter_edits_take_effect_like_gpy(mods):
    """GPy refits when a kernel parameter is assigned or edited in place; the handle
    notices at the next use (no ``parameters_changed()`` call): predictions and a whole
    ``SafeOpt.optimize()`` after the edit equal a freshly built model's."""
    safeopt_amd, gpy, _, _ = mods
    rng = np.random.default_rng(5)
    X = rng.uniform(-2, 2, (60, 2)); Y = smooth(X, 3) + 0.4
    Xs = rng.uniform(-3, 3, (300, 2))

    def fresh(var, ls, noise):
        k = gpy.kern.Matern52(2, variance=var, lengthscale=ls, ARD=True)
        return gpy.models.GPRegression(X, Y, k, noise_var=noise)
    gp = fresh(1.5, [1.0, 1.3], 0.01)
    m0, v0 = gp.predict_noiseless(Xs)
    gp.kern.lengthscale[0] = 0.7                    # in place
    m1, v1 = gp.predict_noiseless(Xs)
    mf, vf = fresh(1.5, [0.7, 1.3], 0.01).predict_noiseless(Xs)
    assert_array_equal(m1, mf); assert_array_equal(v1, vf)
    assert np.max(np.abs(m1 - m0)) > 1e-3
    gp.kern.variance = 2.5                          # assignment
    gp.noise_var = 0.04
    m2, v2 = gp.predict_noiseless(Xs)
    mf, vf = fresh(2.5, [0.7, 1.3], 0.04).predict_noiseless(Xs)
    assert_array_equal(m2, mf); assert_array_equal(v2, vf)
    # inside a BO loop: the resident posterior of the grid follows
    grid = safeopt_amd.linearly_spaced_combinations([(-3, 3)] * 2, 60)
    a = safeopt_amd.SafeOpt(fresh(1.5, [1.0, 1.3], 0.01), grid, 0.0, threshold=0.1)
    a.optimize()
    a.gp.kern.lengthscale[:] = [0.8, 0.9]
    xa = a.optimize()
    b = safeopt_amd.SafeOpt(fresh(1.5, [0.8, 0.9], 0.01), grid, 0.0, threshold=0.1)
    xb = b.optimize()
    assert_array_equal(xa, xb)
    assert_array_equal(a.Q, b.Q)
    for name in ("S", "M", "G"):
        assert_array_equal(getattr(a, name), getattr(b, name))


PRODUCT_CASES = [
    # d, [(n, spec or kind)], N
    (2, [(60, [("RBF", [0]), ("RBF", [1])])], 500),
    (2, [(200, [("Matern52", [0]), ("RBF", [1])]), (90, "Matern32")], 1000),
    (3, [(300, [("Matern32", [0, 1]), ("Matern52", [1, 2])])], 640),
    (3, [(257, [("RBF", [0, 1, 2]), ("Matern52", [0, 1, 2])]), (256, "RBF")], 777),
    (4, [(530, [("RBF", [0]), ("Matern32", [1]), ("Matern52", [2, 3])])], 333),
    (4, [(100, [("RBF", [0, 1]), ("RBF", [2]), ("Matern52", [3]), ("Matern32", [0, 3])]),
         (1040, [("Matern52", [0, 1, 2]), ("RBF", [3])])], 450),
    (6, [(150, [("Matern52", [0, 1, 2, 3]), ("RBF", [4, 5])])], 260),
    (8, [(70, [("RBF", list(range(7))), ("Matern32", [7])]), (300, "Matern52")], 200),
]


@pytest.mark.parametrize("which", ["classic", "pair"])
@pytest.mark.parametrize("d,gpspec,N", PRODUCT_CASES)
def test_grid_sweep_product_kernels(mods, which, d, gpspec, N):
    _, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(N + 31 * d)
    gps, gos = [], []
    for i, (n, spec) in enumerate(gpspec):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 5 + i) + 0.3
        if isinstance(spec, str):
            k, ko = kernels(gpy.kern, spec, d), kernels(gpn, spec, d)
        else:
            k, ko = product_kernel(gpy.kern, d, spec, n), product_kernel(gpn, d, spec, n)
        gps.append(gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2))
        gos.append(gpn.GPRegression(X, Y, ko, noise_var=0.05 ** 2))
    pts = rng.uniform(-3, 3, size=(N, d))
    G = len(gps)
    fmin = np.full(G, 0.1)
    ctx = gps[0]._fitted().ctx
    old = ctx.set_sweep(which)
    try:
        grid = _hip.DeviceGrid(ctx, pts, G)
        grid.confidence([g._fitted() for g in gps], 2.0, fmin)
        Q = grid.download(_hip.Q)
        mean = grid.download(_hip.MEAN); var = grid.download(_hip.VAR)
    finally:
        ctx.set_sweep(old)
    for i, go in enumerate(gos):
        mo, vo = go.predict_noiseless(pts)
        kdiag = float(go.kern.Kdiag(pts[:1])[0])
        check_posterior(mean[i][:, None], var[i][:, None], mo, vo, kdiag)
        sd = np.sqrt(vo[:, 0])
        assert_allclose(Q[:, 2 * i], mo[:, 0] - 2.0 * sd, rtol=0, atol=2e-8)
        assert_allclose(Q[:, 2 * i + 1], mo[:, 0] + 2.0 * sd, rtol=0, atol=2e-8)


# Tensor grids (what linearly_spaced_combinations builds, utilities.py:21-54) with RBF
# kernels are swept through per-axis factor tables (sgp_grid_set_axes): same posterior
# as the generic evaluation within rounding, for 1..4 axes, constant context columns,
# products of RBF parts, shards of the grid, GPs of different sizes; anything else
# (other kernels, rows that are no tensor grid) silently takes the generic path.


SEP_CASES = [
    # sides, context columns, [n per GP], kernel spec, (lo, hi) shard or None
    ([70], 0, [1], "RBF", None), ([37, 29], 0, [200], "RBF", None),
    ([37, 29], 0, [17, 255], "RBF", (100, 1000)), ([9, 8, 11], 0, [130], "RBF", None),
    ([5, 4, 6, 3], 0, [60, 60], "RBF", (7, 355)), ([31, 17], 1, [90], "RBF*RBF", None),
    ([40, 25], 2, [33], "RBF", None), ([64, 3], 0, [256], "RBF", None),
    # more than 256 rows: the paired-wave sweep (whole tiles, cut remainder tiles, a
    # shard that starts inside a grid row, several chunks of 512 rows)
    ([37, 29], 0, [300], "RBF", None), ([23, 19, 7], 0, [600, 257], "RBF", (50, 3000)),
    ([31, 17], 1, [400], "RBF*RBF", None), ([200, 40], 0, [520], "RBF", (1000, 7900)),
    ([200, 100], 0, [300], "RBF", None), ([13, 11, 9], 1, [1100], "RBF", None),
]


@pytest.mark.parametrize("sides,nc,ns,spec,shard", SEP_CASES)
def test_tensor_grid_tables_match_generic(mods, sides, nc, ns, spec, shard):
    sa, gpy, gpn, son = mods
    from safeopt_amd import _hip
    dp = len(sides)
    d = dp + nc
    rng = np.random.default_rng(sum(ns) + 7 * d)
    full = sa.linearly_spaced_combinations([(-3., 3.)] * dp, sides)
    if nc:
        full = np.hstack([full, np.tile(rng.uniform(-1, 1, size=nc), (full.shape[0], 1))])
    axes = _hip.tensor_grid_axes(full)
    assert axes is not None
    lo, hi = shard or (0, full.shape[0])
    gps = []
    for i, n in enumerate(ns):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 3 + i) + 0.3
        if spec == "RBF":
            k = gpy.kern.RBF(d, 1.7, list(rng.uniform(0.6, 1.5, size=d)), ARD=True)
        else:
            k = (gpy.kern.RBF(dp, 1.3, list(rng.uniform(0.6, 1.5, size=dp)), ARD=True,
                              active_dims=list(range(dp))) *
                 gpy.kern.RBF(nc, 0.9, 0.8, active_dims=list(range(dp, d))))
        gps.append(gpy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2))
    devs = [g._fitted() for g in gps]
    ctx = devs[0].ctx
    G = len(ns)
    fmin = np.full(G, 0.1)
    grid = _hip.DeviceGrid(ctx, full[lo:hi], G, lo)
    assert grid.set_axes(axes)
    out = {}
    for which in (8, 0):                 # 8: no factor tables
        old = ctx.set_sweep(which)
        try:
            grid.confidence(devs, 2.0, fmin)
            out[which] = [grid.download(a) for a in (_hip.Q, _hip.MEAN, _hip.VAR, _hip.S)]
        finally:
            ctx.set_sweep(old)
    kd = max(float(g.kern.Kdiag(np.zeros((1, d)))[0]) for g in gps)
    assert_allclose(out[0][1], out[8][1], rtol=0, atol=1e-11 * max(1.0, np.abs(out[8][1]).max()))
    assert_allclose(out[0][2], out[8][2], rtol=0, atol=1e-11 * kd)
    assert_allclose(out[0][0], out[8][0], rtol=0, atol=2e-9)
    assert np.mean(out[0][3] != out[8][3]) < 1e-3
    if nc:
        # a new context: new axis values, new tables
        c = rng.uniform(-1, 1, size=nc)
        grid.set_context(c)
        full[:, dp:] = c
        grid.confidence(devs, 2.0, fmin)
        m1 = grid.download(_hip.MEAN)
        ref = _hip.DeviceGrid(ctx, full[lo:hi], G, lo)
        old = ctx.set_sweep(8)
        try:
            ref.confidence(devs, 2.0, fmin)
        finally:
            ctx.set_sweep(old)
        assert_allclose(m1, ref.download(_hip.MEAN), rtol=0,
                        atol=1e-11 * max(1.0, np.abs(m1).max()))
    # rows that are no tensor grid are refused (and swept as before)
    perm = full[lo:hi].copy()
    if perm.shape[0] > 3:
        perm[[1, 2]] = perm[[2, 1]]
        g2 = _hip.DeviceGrid(ctx, perm, G, lo)
        assert not g2.set_axes(axes)


@pytest.mark.parametrize("kind,d,ns,N", [("Matern52", 2, [500, 500, 500], 2000 + 64 * 256),
                                        ("RBF", 3, [1000], 3000),
                                        ("RBF", 4, [2000, 1500], 64 * 300 + 5),
                                        ("Matern32", 2, [300, 20, 600], 777)])
def test_split_remainder_tiles_same_bits(mods, kind, d, ns, N):
    """A remainder of tiles that would occupy a few workgroups for a whole round is
    cut into runs of accumulator chunks (sweep_pair.hip: pair_plan) whose per-lane
    sums k_pair_split_finish adds in the order of the unsplit loop: mean, var, Q and
    S must be the SAME BITS with and without the cut (a row's posterior must not
    depend on where its tile lands -- also what keeps 1/2/4/8-rank runs identical)."""
    _, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(N + d)
    gps = []
    for i, n in enumerate(ns):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 5 + i) + 0.3
        gps.append(gpy.models.GPRegression(X, Y, kernels(gpy.kern, kind, d), noise_var=0.05 ** 2))
    pts = rng.uniform(-3, 3, size=(N, d))
    G = len(ns)
    fmin = np.full(G, 0.1)
    ctx = gps[0]._fitted().ctx
    out = {}
    old = ctx.set_sweep("pair")
    try:
        for which in ("pair", "pair-nosplit"):
            ctx.set_sweep(which)
            grid = _hip.DeviceGrid(ctx, pts, G)
            ml = grid.confidence([g._fitted() for g in gps], 2.0, fmin)
            out[which] = (ml, grid.download(_hip.Q), grid.download(_hip.S),
                          grid.download(_hip.MEAN), grid.download(_hip.VAR))
    finally:
        ctx.set_sweep(old)
    a, b = out["pair"], out["pair-nosplit"]
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert_array_equal(x, y)


TINY_CASES = [
    # kernel, d, [n per GP], rows
    ("RBF", 1, [1], 300), ("RBF", 2, [5], 1000), ("Matern52", 2, [8], 5000), ("Matern32", 3, [9], 777),
    ("RBF", 2, [16], 4096), ("Matern52", 1, [17], 1000), ("RBF", 2, [20], 50000), ("Matern32", 4, [32], 3000),
    ("RBF", 8, [31], 900), ("Matern52", 5, [13, 2, 32], 2000), ("RBF", 2, [3] * 8, 1500),
    ("RBF*RBF", 4, [24], 2500), ("RBF*RBF", 4, [7, 30], 700), ("Matern52", 7, [19], 256), ("RBF", 6, [32, 32], 257),
    ("Matern52", 2, [33], 3000), ("RBF", 3, [48, 40], 1200), ("RBF*RBF", 4, [47], 600), ("Matern32", 8, [48], 300),
]


@pytest.mark.parametrize("kind,d,ns,N", TINY_CASES)
def test_few_observations_valu_kernel(mods, kind, d, ns, N):
    """Every GP of the launch has <= 48 observations (all examples of the reference): the
    sweep runs on the fp64 VALU, one thread per row (csrc/sweep_tiny.hip).  Against the
    oracle, against the 4-wave matrix-core kernel on the same rows, and the structural
    properties: Q = mean -+ beta sqrt(var) bit for bit, S from Q, max l0 over S."""
    sa, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(sum(ns) + 31 * d + N)

    def kern(ns_):
        if kind == "RBF*RBF":
            return (ns_.RBF(2, 1.3, [0.8, 1.1], ARD=True, active_dims=[0, 1]) *
                    ns_.RBF(2, 0.9, [1.2, 0.7], ARD=True, active_dims=[2, 3]))
        return kernels(ns_, kind, d)
    gps, gos = [], []
    for i, n in enumerate(ns):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 5 + i) + 0.3
        gps.append(gpy.models.GPRegression(X, Y, kern(gpy.kern), noise_var=0.05 ** 2))
        gos.append(gpn.GPRegression(X, Y, kern(gpn), noise_var=0.05 ** 2))
    pts = rng.uniform(-3, 3, size=(N, d))
    G = len(ns)
    fmin = np.where(np.arange(G) % 3 == 2, -np.inf, 0.1)
    ctx = gps[0]._fitted().ctx
    out = {}
    old = ctx.set_sweep("auto")
    try:
        for which in ("auto", "classic"):
            ctx.set_sweep(which)
            grid = _hip.DeviceGrid(ctx, pts, G)
            ml = grid.confidence([g._fitted() for g in gps], 2.0, fmin)
            assert ctx.last_sweep() == ("tiny" if which == "auto" else "classic")
            out[which] = (ml, grid.download(_hip.Q), grid.download(_hip.S),
                          grid.download(_hip.MEAN), grid.download(_hip.VAR))
    finally:
        ctx.set_sweep(old)
    (max_l, any_safe), Q, S, mean, var = out["auto"]
    kd = float(gps[0].kern.Kdiag(np.zeros((1, d)))[0])
    for i, go in enumerate(gos):
        mo, vo = go.predict_noiseless(pts)
        check_posterior(mean[i][:, None], var[i][:, None], mo, vo, kd)
        sd = np.sqrt(var[i])
        assert_array_equal(Q[:, 2 * i], mean[i] - 2.0 * sd)
        assert_array_equal(Q[:, 2 * i + 1], mean[i] + 2.0 * sd)
        # the matrix-core kernel on the same rows: another summation order, same posterior
        assert_allclose(mean[i], out["classic"][3][i], rtol=0, atol=1e-12 * max(1.0, np.abs(mo).max()))
        assert_allclose(var[i], out["classic"][4][i], rtol=0, atol=1e-12 * kd)
    assert_array_equal(S, np.all(Q[:, ::2] > fmin, axis=1))
    assert any_safe == bool(S.any())
    if S.any():
        assert max_l == Q[S, 0].max()
    # points handed over per call (predict, swarm particles): the thread-per-row kernel only
    # with enough rows to hide its chains behind, or when the GP is tiny; same posterior
    few = pts[:50]
    m_few, v_few = gps[0].predict_noiseless(few)
    assert ctx.last_sweep() == ("tiny" if ns[0] <= 10 else "classic")
    assert_allclose(m_few[:, 0], mean[0][:50], rtol=0, atol=1e-12 * max(1.0, np.abs(mean[0]).max()))
    assert_allclose(v_few[:, 0], var[0][:50], rtol=0, atol=1e-12 * kd)
    if ns[0] <= 12:
        many = rng.uniform(-3, 3, size=(1536 * ns[0], d))
        gps[0].predict_noiseless(many)
        assert ctx.last_sweep() == "tiny"
    # from 49 observations on the matrix-core kernels take over: the resident-factor kernel
    # (sweep_mid.hip) for single-part kernels up to d = 4, the 4-wave kernel otherwise
    X = rng.uniform(-2, 2, size=(49, d)); Y = smooth(X, 3) + 0.3
    big = gpy.models.GPRegression(X, Y, kern(gpy.kern), noise_var=0.05 ** 2)
    g1 = _hip.DeviceGrid(ctx, pts, 1)
    g1.confidence([big._fitted()], 2.0, np.zeros(1))
    assert ctx.last_sweep() == ("mid" if (d <= 4 and kind != "RBF*RBF") else "classic")


@pytest.mark.parametrize("kind,d,ns,N,layout,grid", [
    ("RBF", 2, [64], 64 * 520 + 3, "a", True), ("Matern52", 2, [49], 5000, "a", False),
    ("Matern32", 3, [80, 64, 50], 9000, "abc", False), ("RBF", 1, [100], 3000, "a", True),
    ("RBF", 4, [128], 4000, "a", False), ("Matern52", 2, [96], 20001, "aa", False),
    ("RBF", 2, [72], 5, "aab", False), ("RBF", 3, [112], 17 * 19 * 23, "a", True),
    ("Matern52", 1, [128], 777, "a", False), ("RBF", 2, [60], 16 * 12 * 256 + 16, "aaa", True),
    # 129 .. 256 observations: in passes of row blocks, with factor tables only
    ("RBF", 2, [200], 40000, "a", True), ("RBF", 3, [144, 130], 17 * 19 * 23, "ab", True),
    ("RBF", 2, [256], 20000, "aa", True), ("RBF", 1, [230, 160, 129], 5000, "abc", True),
    ("Matern52", 2, [150], 9000, "a", True)])
def test_resident_factor_kernel_49_to_128(mods, kind, d, ns, N, layout, grid):
    """49 .. 128 observations, single-part kernels, d <= 4: the resident-factor kernel
    (csrc/sweep_mid.hip; the whole L^-1 of every GP in LDS, straight-line j-block / row-block
    nest, three waves per SIMD).  Posterior against the oracle and against the 4-wave kernel
    (another summation order: 1e-12), Q = mean -+ beta sd exactly, S and max l0 from Q; GPs
    with a shared factor (same bits as swept on their own), factor tables on tensor grids,
    ragged row counts, and the posterior of a prefix of the rows = the prefix of the
    posterior (the kernel is chosen by the GPs alone)."""
    sa, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(N + 7 * d + sum(ns))
    groups = {}
    gps, gos = [], []
    for i, c in enumerate(layout):
        if c not in groups:
            n = ns[len(groups) % len(ns)]
            groups[c] = rng.uniform(-2, 2, size=(n, d))
        X = groups[c]
        Y = smooth(X, 5 + i) + 0.3
        gps.append(gpy.models.GPRegression(X, Y, kernels(gpy.kern, kind, d), noise_var=0.05 ** 2))
        gos.append(gpn.GPRegression(X, Y, kernels(gpn, kind, d), noise_var=0.05 ** 2))
    if grid:
        side = max(2, int(round(N ** (1.0 / d))))
        pts = sa.linearly_spaced_combinations([(-3., 3.)] * d, [side + k for k in range(d)])
    else:
        pts = rng.uniform(-3, 3, size=(N, d))
    G = len(layout)
    fmin = np.where(np.arange(G) % 3 == 2, -np.inf, 0.1)
    ctx = gps[0]._fitted().ctx
    out = {}
    old = ctx.set_sweep("auto")
    old_share = ctx.set_share(True)
    nmax = max(int(c.shape[0]) for c in groups.values())
    followers = len(set(layout)) < len(layout)

    def expected(which, share, tables):
        # up to 128 observations everything resident; beyond, passes -- with factor tables
        # (tensor grid, RBF) and without followers of a shared factor only
        if which == "classic":
            return "classic"
        if nmax <= 128:
            return "mid"
        return "mid" if (tables and kind == "RBF" and not (share and followers)) else "classic"
    variants = [("auto", True, grid), ("auto", False, False), ("classic", False, False)]
    if grid:
        variants.append(("auto", False, True))
    try:
        for which, share, tables in variants:
            ctx.set_sweep(which)
            ctx.set_share(share)
            g = _hip.DeviceGrid(ctx, pts, G)
            if tables:
                assert g.set_axes(_hip.tensor_grid_axes(pts))
            ml = g.confidence([gp._fitted() for gp in gps], 2.0, fmin)
            assert ctx.last_sweep() == expected(which, share, tables)
            out[(which, share, tables)] = (ml, g.download(_hip.Q), g.download(_hip.S),
                                           g.download(_hip.MEAN), g.download(_hip.VAR))
    finally:
        ctx.set_sweep(old)
        ctx.set_share(old_share)
    first = ("auto", False, True) if grid else ("auto", True, False)
    (max_l, any_safe), Q, S, mean, var = out[first]
    own = out[("auto", False, False)]
    others = [out[k] for k in out if k != first]
    if not grid and nmax <= 128:
        # the shared factor: same bits as every GP swept on its own (factor tables: their
        # covariances are products of table entries, another rounding -- 1e-12 below)
        for x, y in zip(out[("auto", True, False)][1:], own[1:]):
            assert_array_equal(x, y)
    sel = rng.choice(pts.shape[0], size=min(400, pts.shape[0]), replace=False)
    for i, go in enumerate(gos):
        kd = float(gps[i].kern.Kdiag(np.zeros((1, d)))[0])
        mo, vo = go.predict_noiseless(pts[sel])
        check_posterior(mean[i][sel, None], var[i][sel, None], mo, vo, kd)
        sd = np.sqrt(var[i])
        assert_array_equal(Q[:, 2 * i], mean[i] - 2.0 * sd)
        assert_array_equal(Q[:, 2 * i + 1], mean[i] + 2.0 * sd)
        for other in others:
            assert_allclose(mean[i], other[3][i], rtol=0, atol=1e-11 * max(1.0, np.abs(mo).max()))
            assert_allclose(var[i], other[4][i], rtol=0, atol=1e-11 * kd)
    assert_array_equal(S, np.all(Q[:, ::2] > fmin, axis=1))
    assert any_safe == bool(S.any())
    if S.any():
        assert max_l == Q[S, 0].max()
    # a prefix of the rows, handed over per call: the same kernel, the same bits
    k = min(pts.shape[0], 37)
    ctx.set_share(False)
    try:
        m_all, v_all = gps[0].predict_noiseless(pts)
        k_all = ctx.last_sweep()
        m_few, v_few = gps[0].predict_noiseless(pts[:k])
        k_few = ctx.last_sweep()
    finally:
        ctx.set_share(old_share)
    # (point sets of a few thousand rows against 100+ observations go chip-wide, factor.hip
    # "few-points": by the GP and the row count of the CALL -- test_predict_of_a_prefix_of_
    # the_points pins that path; here the sweep kernel)
    ok = ("mid", "few-points") if nmax <= 128 else ("classic", "few-points")   # (points: no tables)
    assert k_few in ok and k_all in ok
    if k_all == k_few == "mid":
        assert_array_equal(m_few, m_all[:k])
        assert_array_equal(v_few, v_all[:k])
        if not grid:
            assert_array_equal(m_all[:, 0], own[3][0])
    else:
        kd = float(gps[0].kern.Kdiag(np.zeros((1, d)))[0])
        assert_allclose(m_few, m_all[:k], rtol=0, atol=1e-11 * max(1.0, np.abs(m_all).max()))
        assert_allclose(v_few, v_all[:k], rtol=0, atol=1e-11 * kd)


@pytest.mark.parametrize("kind,d,ns,N,grid", [("RBF", 2, [1], 700, False), ("Matern52", 2, [20], 40000, False),
                                              ("RBF", 2, [64], 64 * 520 + 3, True),
                                              ("Matern32", 3, [17, 33, 5], 9000, False),
                                              ("RBF", 1, [100], 3000, True), ("RBF", 5, [48, 48], 5000, False),
                                              # the largest that stays (28 positions), one too many
                                              ("RBF", 2, [112], 20000, True), ("RBF", 2, [113], 20000, True),
                                              ("RBF*RBF", 4, [40], 2500, False)])
def test_resident_factor_same_bits(mods, kind, d, ns, N, grid):
    """Small factors (every example of the reference: n <= 20) stay in LDS for the whole
    launch of the 4-wave kernel -- no LDS-DMA, wait or barrier per stage, the waves of a
    workgroup run free: mean, var, Q and S must be the SAME BITS as when the very same
    stages are streamed through the double buffer."""
    sa, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(N + d + sum(ns))
    def kern(ns_):
        if kind == "RBF*RBF":
            return (ns_.RBF(2, 1.3, [0.8, 1.1], ARD=True, active_dims=[0, 1]) *
                    ns_.RBF(2, 0.9, [1.2, 0.7], ARD=True, active_dims=[2, 3]))
        return kernels(ns_, kind, d)
    gps, gos = [], []
    for i, n in enumerate(ns):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 5 + i) + 0.3
        gps.append(gpy.models.GPRegression(X, Y, kern(gpy.kern), noise_var=0.05 ** 2))
        gos.append(gpn.GPRegression(X, Y, kern(gpn), noise_var=0.05 ** 2))
    if grid:
        side = max(2, int(round(N ** (1.0 / d))))
        pts = sa.linearly_spaced_combinations([(-3., 3.)] * d, [side + k for k in range(d)])
    else:
        pts = rng.uniform(-3, 3, size=(N, d))
    G = len(ns)
    fmin = np.full(G, 0.1)
    ctx = gps[0]._fitted().ctx
    out = {}
    old = ctx.set_sweep("classic")
    try:
        for which in ("classic", "classic-streamed"):
            ctx.set_sweep(which)
            g = _hip.DeviceGrid(ctx, pts, G)
            if grid:
                assert g.set_axes(_hip.tensor_grid_axes(pts))
            ml = g.confidence([gp._fitted() for gp in gps], 2.0, fmin)
            out[which] = (ml, g.download(_hip.Q), g.download(_hip.S),
                          g.download(_hip.MEAN), g.download(_hip.VAR))
    finally:
        ctx.set_sweep(old)
    a, b = out["classic"], out["classic-streamed"]
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert_array_equal(x, y)
    # ... and the oracle, on a sample of rows
    sel = rng.choice(pts.shape[0], size=min(300, pts.shape[0]), replace=False)
    for i, go in enumerate(gos):
        mo, vo = go.predict_noiseless(pts[sel])
        kd = float(gps[i].kern.Kdiag(np.zeros((1, d)))[0])
        check_posterior(a[3][i][sel, None], a[4][i][sel, None], mo, vo, kd)


@pytest.mark.parametrize("n,N,layout", [(300, 5000, "aaa"), (530, 3000, "aab"), (400, 20000, "abb"),
                                        (1100, 2500, "aa"),
                                        # riders (up to 2 per leader form alpha . k in the leader's
                                        # stages), a third follower with stages of its own, two
                                        # groups, a rider as the LAST GP, cut remainder tiles
                                        (300, 3000, "aaaa"), (520, 2000, "aabbb"), (280, 999, "abbba"),
                                        (500, 64 * 256 + 64 * 40, "aaa"), (1000, 64 * 300 + 7, "baa"),
                                        # the 4-wave kernel (n <= 256): riders only
                                        (200, 5000, "aaa"), (40, 700, "aabbb"), (256, 3000, "abba"),
                                        (130, 1200, "aaaa"),
                                        # ... with the factors resident in LDS (forced onto the 4-wave
                                        # kernel below: up to 48 observations the VALU kernel would run)
                                        (60, 2500, "aaa"), (30, 900, "aab"), (96, 4000, "aa")])
def test_shared_factor_same_bits(mods, n, N, layout):
    """BASELINE.json config 3 is a multi-output GP: its GPs have the same inputs,
    kernel and noise, hence the same L^-1.  The paired sweep then takes |L^-1 k|^2
    from the first of them and only forms alpha . k for the others
    (sgp_ctx_set_share, default on): Q, S, mean and var must be the same bits as
    with every GP swept on its own (gp_opt.py:466-476 loops independently), also
    after identical one-row appends; GPs that differ in anything are not shared."""
    _, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(n + N)
    d = 2
    Xa = rng.uniform(-2, 2, size=(n, d)); Xb = rng.uniform(-2, 2, size=(n, d))
    gps = []
    for i, c in enumerate(layout):
        X = Xa if c == "a" else Xb
        gps.append(gpy.models.GPRegression(X, smooth(X, 7 + i) + 0.3, kernels(gpy.kern, "Matern52", d),
                                           noise_var=0.05 ** 2))
    pts = rng.uniform(-3, 3, size=(N, d))
    G = len(layout)
    fmin = np.full(G, 0.1)
    ctx = gps[0]._fitted().ctx
    # (riders are a matter of the matrix-core kernels: keep small problems on the 4-wave one)
    forced = ctx.set_sweep("classic") if n <= 112 else None

    def sweep():
        res = {}
        for on in (True, False):
            old = ctx.set_share(on)
            try:
                grid = _hip.DeviceGrid(ctx, pts, G)
                ml = grid.confidence([g._fitted() for g in gps], 2.0, fmin)
                assert ctx.last_sweep() == ("classic" if max(g.X.shape[0] for g in gps) <= 256
                                            else "pair")
                res[on] = (ml, grid.download(_hip.Q), grid.download(_hip.S),
                           grid.download(_hip.MEAN), grid.download(_hip.VAR))
            finally:
                ctx.set_share(old)
        assert res[True][0] == res[False][0]
        for x, y in zip(res[True][1:], res[False][1:]):
            assert_array_equal(x, y)
        return res[True]

    try:
        r = sweep()
        if layout[0] == layout[1]:          # equal factors: equal variances
            assert_array_equal(r[4][0], r[4][1])
        # the same new observation point for every GP (SafeOpt.add_new_data_point)
        xn = rng.uniform(-1, 1, size=(1, d))
        for i, gp in enumerate(gps):
            gp.set_XY(np.vstack([gp.X, xn]), np.vstack([gp.Y, [[0.4 + 0.1 * i]]]))
        sweep()
    finally:
        if forced is not None:
            ctx.set_sweep(forced)


def test_predict_product_kernel_and_refit(mods):
    _, gpy, gpn, _ = mods
    rng = np.random.default_rng(5)
    X = rng.uniform(-2, 2, size=(30, 2)); Y = smooth(X, 3)
    k = gpy.kern.RBF(1, 2., 1., active_dims=[0]) * gpy.kern.RBF(1, 2., 1.3, active_dims=[1], name='c')
    ko = gpn.RBF(1, 2., 1., active_dims=[0]) * gpn.RBF(1, 2., 1.3, active_dims=[1], name='c')
    gp = gpy.models.GPRegression(X, Y, k, noise_var=0.01)
    go = gpn.GPRegression(X, Y, ko, noise_var=0.01)
    Xs = rng.uniform(-3, 3, size=(300, 2))
    check_posterior(*gp.predict_noiseless(Xs), *go.predict_noiseless(Xs), 4.0)
    # set_XY with one more / one fewer row (what SafeOpt does every iteration)
    Xn = np.vstack([X, [[0.3, -0.2]]]); Yn = np.vstack([Y, [[0.5]]])
    gp.set_XY(Xn, Yn); go.set_XY(Xn, Yn)
    check_posterior(*gp.predict_noiseless(Xs), *go.predict_noiseless(Xs), 4.0)
    gp.set_XY(X[:-3], Y[:-3]); go.set_XY(X[:-3], Y[:-3])
    check_posterior(*gp.predict_noiseless(Xs), *go.predict_noiseless(Xs), 4.0)
    assert_array_equal(gp.X, X[:-3])


def test_jitter_and_failure(mods):
    _, gpy, _, _ = mods
    # duplicated inputs with zero noise: needs GPy's jitter escalation
    X = np.zeros((4, 1)); Y = np.ones((4, 1))
    gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(1), noise_var=0.)
    m, v = gp.predict_noiseless(np.zeros((1, 1)))
    assert np.isfinite(m).all() and np.isfinite(v).all()
    with pytest.raises(np.linalg.LinAlgError):
        gpy.models.GPRegression(X, Y, gpy.kern.RBF(1, variance=-1.), noise_var=0.)


# ---------------------------------------------------------------------------


@pytest.mark.parametrize("n,P", [(130, 5), (400, 1), (500, 20), (2000, 64), (2000, 17),
                                 (2000, 300), (300, 1000), (1000, 4096), (700, 4097)])
def test_few_points_path(mods, n, P):
    """P <= 4096 points at n >= 128 (4097: the sweep again): posterior and swarm fitness come out of the
    triangular multi-RHS path (posterior_small) instead of one sweep tile --
    SafeOptSwarm's default swarm (20 particles) and the single-point predictions
    of gp_opt.py:1117, 1132.  Same oracle, same tolerances."""
    safeopt_amd, gpy, gpn, son = mods
    from bench import make_config, build_gps
    cfg = make_config(5)
    cfg["X"], cfg["Y"], cfg["n"] = cfg["X"][:n], cfg["Y"][:n], n
    gps, gos = build_gps(cfg, gpy), build_gps(cfg, gpn)
    parts = np.random.default_rng(n + P).uniform(-3, 3, size=(P, 4))
    for g in range(2):
        m, v = gps[g].predict_noiseless(parts)
        mo, vo = gos[g].predict_noiseless(parts)
        check_posterior(m, v, mo, vo, 2.0)
    opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * 4,
                                   threshold=cfg["threshold"])
    opt.best_lower_bound = 0.4
    for st in ["greedy", "maximizers", "expanders", "safe_set"]:
        v, s = opt._compute_particle_fitness(st, parts)
        vo, so = son.swarm_fitness(gos, parts, st, 2., cfg["fmin"], opt.scaling, 0.4)
        assert_array_equal(s, so)
        assert_allclose(v, vo, rtol=1e-7, atol=1e-8)


def test_append_pop_match_refit(mods):
    """sgp_gp_append / sgp_gp_pop == a fresh fit, across the 16/32/64 padding
    boundaries, and the rank-1 record is consistent with the oracle."""
    _, gpy, gpn, _ = mods
    rng = np.random.default_rng(21)
    X = rng.uniform(-2, 2, size=(90, 2)); Y = smooth(X, 5)
    Xs = rng.uniform(-3, 3, size=(500, 2))
    gp = gpy.models.GPRegression(X[:29], Y[:29], kernels(gpy.kern, "Matern52", 2), noise_var=0.05 ** 2)
    for n in range(30, 71):
        v0 = gp._dev.version
        gp.set_XY(X[:n], Y[:n])
        assert gp._dev.appended and gp._dev.version == v0 + 1 and gp._dev.n == n
        if n in (30, 31, 32, 33, 47, 48, 49, 63, 64, 65, 70):
            go = gpn.GPRegression(X[:n], Y[:n], kernels(gpn, "Matern52", 2), noise_var=0.05 ** 2)
            check_posterior(*gp.predict_noiseless(Xs), *go.predict_noiseless(Xs), 1.7)
            Linv, alpha = gp._dev.factor()
            assert np.max(np.abs(Linv - np.linalg.inv(go.L))) < 1e-8
            assert np.max(np.abs(alpha - go.woodbury_vector.ravel())) < 1e-8 * np.max(np.abs(alpha))
    for n in range(69, 40, -1):
        gp.set_XY(X[:n], Y[:n])
        assert not gp._dev.appended and gp._dev.n == n
    go = gpn.GPRegression(X[:41], Y[:41], kernels(gpn, "Matern52", 2), noise_var=0.05 ** 2)
    check_posterior(*gp.predict_noiseless(Xs), *go.predict_noiseless(Xs), 1.7)
    # a change that is not a one-row append/pop refits
    gp.set_XY(X[10:60], Y[10:60])
    go = gpn.GPRegression(X[10:60], Y[10:60], kernels(gpn, "Matern52", 2), noise_var=0.05 ** 2)
    check_posterior(*gp.predict_noiseless(Xs), *go.predict_noiseless(Xs), 1.7)
    # duplicate point with (almost) no noise: bordered pivot ~ 0 -> falls back to a refit
    g2 = gpy.models.GPRegression(X[:5], Y[:5], gpy.kern.RBF(2), noise_var=0.)
    g2.set_XY(np.vstack([X[:5], X[4:5]]), np.vstack([Y[:5], Y[4:5]]))
    assert np.isfinite(g2.predict_noiseless(Xs[:4])[0]).all()


def test_long_axis_tables_are_skipped(mods):
    """A per-axis factor table is ``n_pad / 16 x count x 128`` bytes: on a grid with ONE long
    axis that is the whole covariance matrix (a 1-D grid of 1e6 points, n = 544: 4.3 GB, past
    the 32-bit offsets of the sweeps).  Beyond 256 MB per GP the tables are not built and the
    covariances are evaluated -- the same bits as with tables switched off -- while a grid
    below the budget still goes through them (last bits differ).  The paired kernel (more
    than 256 rows in a factor) takes tables only while they fit half an L2 next to the
    factor it streams."""
    sa, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(77)
    # (n, rows, tables expected): 13 blocks x 6e5 x 128 B = 998 MB: skipped | 33 MB: tables |
    # paired kernel, 19 x 2e4 x 128 B = 49 MB > 2 MB: skipped | paired, 19 x 600 x 128 B: tables
    for n, N, tables in ((200, 600000, False), (200, 20000, True), (300, 20000, False),
                         (300, 600, True)):
        X = rng.uniform(-2.5, 2.5, size=(n, 1))
        Y = smooth(X, 5) - smooth(X, 5).min() + 0.5
        gp = gpy.models.GPRegression(X, Y, gpy.kern.RBF(1, 2.0, 1.0), noise_var=0.05 ** 2)
        dev = gp._fitted()
        ctx = dev.ctx
        grid = sa.linearly_spaced_combinations([(-3., 3.)], N)
        out = {}
        for name in ("auto", "auto-notables"):
            g = _hip.DeviceGrid(ctx, grid, 1)
            assert g.set_axes(_hip.tensor_grid_axes(grid))
            old = ctx.set_sweep(name)
            try:
                g.confidence([dev], 2.0, np.zeros(1))
            finally:
                ctx.set_sweep(old)
            out[name] = g.download(_hip.Q)
        assert np.max(np.abs(out["auto"] - out["auto-notables"])) < 1e-10
        assert np.array_equal(out["auto"], out["auto-notables"]) == (not tables), (n, N)


@pytest.mark.parametrize("n,kind", [(8, "RBF"), (20, "Matern52"), (40, "RBF"), (100, "Matern32"),
                                    (200, "RBF"), (300, "Matern52")])
def test_predict_of_a_prefix_of_the_points(mods, n, kind):
    """``predict(P)[:k]`` against ``predict(P[:k])`` (gp.predict_noiseless, gp_opt.py:469,
    929, 973).  For a set of points handed over per call the kernel is chosen for latency
    by (n, number of rows) -- VALU kernel / 4-wave / paired / few-points path -- and the
    kernels sum in different orders.  The guarantee, pinned here: within ONE kernel a
    row's posterior does not depend on which other rows it was submitted with (the same
    bits); across kernels it moves by at most 1e-12 of the prior variance.  (Grids are
    different: there the kernel depends on the GPs alone -- rank- and shard-invariant.)"""
    sa, gpy, gpn, son = mods
    rng = np.random.default_rng(n)
    d = 2
    X = rng.uniform(-2, 2, size=(n, d))
    Y = smooth(X, 3)
    gp = gpy.models.GPRegression(X, Y, getattr(gpy.kern, kind)(d, 2.0, [0.9, 1.2], ARD=True),
                                 noise_var=0.05 ** 2)
    ctx = gp._fitted().ctx
    P = rng.uniform(-3, 3, size=(70000, d))
    mf, vf = gp.predict_noiseless(P)
    kf = ctx.last_sweep()
    seen = set()
    for k in (1, 20, 777, 4096, 4097, 30000, 69999):
        m, v = gp.predict_noiseless(P[:k])
        kk = ctx.last_sweep()
        seen.add(kk)
        if kk == kf:
            assert_array_equal(m, mf[:k])
            assert_array_equal(v, vf[:k])
        else:
            assert np.max(np.abs(m - mf[:k])) <= 1e-12 * max(1.0, np.max(np.abs(mf)))
            assert np.max(np.abs(v - vf[:k])) <= 2.0 * 1e-12
        # ... and a second call with the same rows repeats the bits
        m2, v2 = gp.predict_noiseless(P[:k])
        assert ctx.last_sweep() == kk
        assert_array_equal(m, m2)
        assert_array_equal(v, v2)
    print("n = %d: full set by %s, prefixes by %s" % (n, kf, sorted(seen)))


@pytest.mark.timeout(1800)
def test_no_kernel_reads_what_nobody_wrote():
    """SGP_POISON=1 fills every fresh device allocation with 0xFF bytes (NaN as doubles): the
    launches that mix GPs of different sizes, the factor tables, the few-points path and the
    resident-factor kernel give the same results -- nothing depends on what an earlier call
    left at an address.  (Found with it: the resident-factor kernel read a small GP's factor
    tables up to the j-blocks of the largest factor of the launch -- finite garbage times the
    zeros of its L^-1 on most days.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SGP_POISON="1")
    sel = ("tables_match_generic or few_points_path or resident_factor_kernel_49_to_128 "
           "or test_grid_sweep_both_kernels or shared_factor_same_bits")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_posterior.py"),
                        "-q", "-m", "gpu", "-x", "--tb=line", "-k", sel],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-3000:]
