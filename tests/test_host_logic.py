"""Host-side logic that needs no GPU: reference plumbing tests mirrored
(safeopt/tests/test_gps.py), grid generator, shard planner, rank merges, PSO."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import safeopt_amd
import safeopt_amd.gpy as gpy
from safeopt_amd import dist
from safeopt_amd.gp_opt import GaussianProcessOptimization


class FakeGP(object):
    """Duck-typed handle with host-only set_XY (the reference's seam is duck
    typing too); kernels are the product's own objects."""

    def __init__(self, X, Y, kernel):
        self.kern = kernel
        self.input_dim = np.atleast_2d(X).shape[1]
        self.set_XY(X, Y)

    def set_XY(self, X, Y):
        self.X = np.array(np.atleast_2d(X), dtype=float)
        self.Y = np.array(np.atleast_2d(Y), dtype=float)


@pytest.fixture
def gps():
    return (FakeGP([[0]], [[0]], gpy.kern.RBF(1, variance=2)),
            FakeGP([[0]], [[0]], gpy.kern.Matern32(1, variance=4)))


def test_init(gps):                                   # test_gps.py:27-46
    gp1, _ = gps
    opt = GaussianProcessOptimization(gp1, fmin=0, beta=2, num_contexts=1)
    assert opt.beta(0) == 2
    opt = GaussianProcessOptimization(gp1, fmin=[0], beta=lambda x: 5, num_contexts=1)
    assert opt.beta(10) == 5


def test_multi_init(gps):                             # test_gps.py:48-60
    opt = GaussianProcessOptimization(list(gps), fmin=0, beta=2, num_contexts=1)
    assert_allclose(opt.scaling, np.array([np.sqrt(2), np.sqrt(4)]))


def test_scaling(gps):                                # test_gps.py:62-75
    gp1, gp2 = gps
    pytest.raises(ValueError, GaussianProcessOptimization, [gp1, gp2], 2, scaling=[5])
    opt = GaussianProcessOptimization([gp1, gp2], fmin=[1, 0], scaling=[1, 2])
    assert_allclose(opt.scaling, np.array([1, 2]))


def test_data_adding(gps):                            # test_gps.py:77-120
    gp1, gp2 = gps
    gp1.set_XY(np.array([[0.]]), np.array([[1.]]))
    opt = GaussianProcessOptimization(gp1, 0)
    opt.add_new_data_point(2, 3)
    x, y = opt.data
    assert_allclose(x, [[0], [2]]); assert_allclose(y, [[1], [3]])
    gp1.set_XY(np.array([[0.]]), np.array([[1.]]))
    gp2.set_XY(np.array([[0.]]), np.array([[11.]]))
    opt = GaussianProcessOptimization([gp1, gp2], [0, 1])
    opt.add_new_data_point(2, [2, 3])
    assert_allclose(opt.x, [[0], [2]]); assert_allclose(opt.y, [[1, 11], [2, 3]])
    opt.add_new_data_point(3, [2, np.nan])
    assert_allclose(opt.x, [[0], [2], [3]])
    assert_allclose(opt.y, [[1, 11], [2, 3], [2, np.nan]])
    for i, gp in enumerate(opt.gps):
        ok = ~np.isnan(opt.y[:, i])
        assert_allclose(gp.X, opt.x[ok, :]); assert_allclose(gp.Y[:, 0], opt.y[ok, i])
    opt.remove_last_data_point()
    assert_allclose(opt.x, [[0], [2]]); assert_allclose(opt.y, [[1, 11], [2, 3]])
    for i, gp in enumerate(opt.gps):
        assert_allclose(gp.X, opt.x); assert_allclose(gp.Y[:, 0], opt.y[:, i])


def test_contexts():                                  # test_gps.py:122-142
    gp1 = FakeGP([[0, 0]], [[5]], gpy.kern.RBF(2, variance=2))
    gp2 = FakeGP([[0, 0]], [[6]], gpy.kern.Matern32(2, variance=4))
    opt = GaussianProcessOptimization([gp1, gp2], fmin=[0, 0], num_contexts=1)
    opt.add_new_data_point(1, [3, 4], context=2)
    assert_allclose(opt.x, [[0, 0], [1, 2]]); assert_allclose(opt.y, [[5, 6], [3, 4]])
    for i, gp in enumerate(opt.gps):
        assert_allclose(gp.X, opt.x); assert_allclose(gp.Y[:, 0], opt.y[:, i])


def test_different_measurements_rejected():
    a = FakeGP([[0.]], [[1.]], gpy.kern.RBF(1)); b = FakeGP([[1.]], [[1.]], gpy.kern.RBF(1))
    with pytest.raises(NotImplementedError):
        GaussianProcessOptimization([a, b], 0.)


def test_grid_order_and_layout():
    g = safeopt_amd.linearly_spaced_combinations([(-1, 1), (0, 3)], [3, 4])
    assert g.shape == (12, 2) and g.flags["F_CONTIGUOUS"]
    assert_allclose(g[:4, 0], [-1, 0, 1, -1])        # first variable fastest
    assert_allclose(g[:4, 1], [0, 0, 0, 1])
    g1 = safeopt_amd.linearly_spaced_combinations([(-2, 2)], 5)
    assert g1.shape == (5, 1)
    g3 = safeopt_amd.linearly_spaced_combinations([(0, 1)] * 3, 2)
    assert g3.shape == (8, 3)


def test_kernel_descriptor():
    k = gpy.kern.RBF(2, variance=2., lengthscale=[1., 2.], ARD=True)
    d, kinds, var, inv = k._desc(2)
    assert d == 2 and list(kinds) == [0] and var[0] == 2.
    assert_allclose(inv, [[1., .5]])
    kp = gpy.kern.RBF(1, 2., 1., active_dims=[0]) * gpy.kern.Matern52(1, 3., 4., active_dims=[1], name='c')
    d, kinds, var, inv = kp._desc()
    assert d == 2 and list(kinds) == [0, 2]
    assert_allclose(inv, [[1., 0.], [0., .25]])
    assert_allclose(kp.Kdiag(np.zeros((3, 2))), [6., 6., 6.])
    assert kp.c.variance[0] == 3. and kp.copy().c.variance[0] == 3.
    with pytest.raises(ValueError):
        gpy.kern.RBF(2, lengthscale=[1., 2.])          # non-ARD, two lengthscales
    with pytest.raises(ValueError):
        k._desc(1)


def test_shard_range_partitions():
    for N in (1, 7, 1000, 10 ** 6 + 3):
        for world in (1, 2, 3, 8):
            edges = [dist.shard_range(N, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_fewer_rows_than_ranks_is_refused_on_every_rank(gps):
    class Comm(dist.LocalComm):
        rank, world = 1, 4
    with pytest.raises(ValueError, match="3 rows for 4 ranks"):
        safeopt_amd.SafeOpt(gps[0], np.array([[0.], [1.], [2.]]), 0., comm=Comm())


def test_tensor_grid_detection_on_the_host():
    """What SafeOpt hands to sgp_grid_set_axes: counts, strides and axis values of a
    parameter set built like linearly_spaced_combinations builds it (utilities.py:21-54),
    with constant context columns; anything else is None (the device then evaluates)."""
    from safeopt_amd._hip import tensor_grid_axes
    grid = safeopt_amd.linearly_spaced_combinations([(-1., 1.), (0., 3.), (2., 5.)], [4, 3, 5])
    counts, strides, values = tensor_grid_axes(grid)
    assert sorted(counts) == [3, 4, 5] and int(np.prod(counts)) == grid.shape[0]
    for k in range(3):
        idx = (np.arange(grid.shape[0]) // strides[k]) % counts[k]
        assert_array_equal(values[k][idx], grid[:, k])
    # a context column: one point, stride 1
    with_ctx = np.hstack([grid, np.full((grid.shape[0], 1), 0.7)])
    c, st, v = tensor_grid_axes(with_ctx)
    assert c[3] == 1 and v[3][0] == 0.7 and c[:3] == counts
    # one column: a 1-D grid; a single row
    c, st, v = tensor_grid_axes(np.linspace(0, 1, 7)[:, None])
    assert c == [7] and st == [1]
    assert tensor_grid_axes(grid[:1])[0] == [1, 1, 1]
    # two rows swapped: the host looks at run lengths and periods only (necessary
    # conditions) -- its candidate does not reproduce the rows, which is what the device
    # check (DeviceGrid.set_axes, every row, bit for bit) finds and refuses
    perm = grid.copy(); perm[[1, 2]] = perm[[2, 1]]
    cand = tensor_grid_axes(perm)
    if cand is not None:
        idx = (np.arange(perm.shape[0]) // cand[1][2]) % cand[0][2]
        assert not np.array_equal(cand[2][2][idx], perm[:, 2])
    # not tensor grids at all: a ragged tail, random points, no rows
    assert tensor_grid_axes(grid[:-1]) is None
    assert tensor_grid_axes(np.random.default_rng(0).uniform(size=(50, 2))) is None
    assert tensor_grid_axes(np.zeros((0, 2))) is None


def test_merge_topk_and_argmax():
    w, i = dist.merge_topk([[3., 1., -np.inf], [3., 2., 2.]], [[5, 9, -1], [7, 4, 8]], 4)
    assert_array_equal(i, [7, 5, 8, 4]); assert_allclose(w, [3., 3., 2., 2.])
    w, i = dist.merge_topk([[0, 0], [0, 0]], [[9, 2], [5, -1]], 8, by_index=True)
    assert_array_equal(i, [2, 5, 9])
    assert dist.merge_argmax([1., 2., 2.], [3, 9, 4]) == (2., 4)   # lowest index wins
    assert dist.merge_argmax([-np.inf, 1.], [-1, 6]) == (1., 6)
    assert dist.merge_argmax([-np.inf], [-1]) == (-np.inf, -1)
    lc = dist.LocalComm()
    assert_allclose(lc.allreduce_max(np.array([1., 2.])), [1., 2.])
    assert lc.allgather(np.arange(3)).shape == (1, 3)


def test_swarm_optimization_reproducible():
    """PSO on a toy fitness: deterministic under the NumPy global RNG, honours
    the safety mask and the box."""
    target = np.array([0.3, -0.2])

    def fitness(p):
        return -np.sum((p - target) ** 2, axis=1), np.all(np.abs(p) <= 1.0, axis=1)

    def run():
        np.random.seed(0)
        s = safeopt_amd.SwarmOptimization(20, np.array([0.1, 0.1]), fitness,
                                          bounds=[(-2., 2.), (-2., 2.)])
        s.init_swarm(np.random.uniform(-1, 1, size=(20, 2)))
        s.run_swarm(50)
        return s.global_best.copy(), s.best_values.max()
    a, va = run(); b, vb = run()
    assert_array_equal(a, b)
    assert va == vb
    assert np.linalg.norm(a - target) < 0.05 and np.all(np.abs(a) <= 1.0)


# ---------------------------------------------------------------------------
# exact ties in the visiting order (gp_opt.py:542-552)
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_tie_settlement_matches_reference(seed):
    """The product's host driver on the NumPy stand-in backend (device order =
    width descending, index descending) must mark the candidate the REFERENCE
    marks when widths tie exactly -- 4 of the 6 fixtures have a winner that is
    not the highest tied index."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import safeopt_amd
    from oracle import gp_numpy as gpn
    from _golden import load, make_kernel
    from _oracle_backend import OracleGridBackend, use_oracle_backend
    use_oracle_backend()
    z, meta = load("ties_1d_seed%d" % seed)
    gp = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                          noise_var=meta["noise_vars"][0])
    opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"])
    opt.Q = z["Q"]
    be = opt._backend
    for i, g in enumerate(be.gps):      # the stand-in predicts from the GP; Q is assigned
        m, v = g.predict_noiseless(be.x)
        be.mean[:, i], be.var[:, i] = m.ravel(), v.ravel()
    opt.compute_sets()
    assert np.array_equal(opt.S, z["S"]) and np.array_equal(opt.M, z["M"])
    assert np.array_equal(opt.G, z["G"])
    assert np.array_equal(opt.get_new_query_point(), z["x_next"])


def test_sample_gp_function_matches_reference():
    """utilities.sample_gp_function against the reference run under the same
    global seed: node values, noiseless and noisy evaluations (RNG call order),
    kernel and linear interpolation, with and without a mean function.  (The
    covariance comes from the oracle kernel here; the device-handle path of the
    package's own kernels is covered by the GPU suite.)"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import safeopt_amd
    from oracle import gp_numpy as gpn
    from _golden import load, make_kernel
    z, meta = load("sample_gp_function")
    for tag in ("rbf1", "m52_2"):
        m = meta[tag]
        k = make_kernel(gpn, m["kernel"])
        bounds = [tuple(b) for b in m["bounds"]]
        xq = z[tag + "_xq"]
        for interp in ("kernel", "linear"):
            for mean in (None, "mean"):
                mf = None if mean is None else (lambda x: 0.3 * x[:, :1] - 0.1)
                np.random.seed(m["seed"])
                f = safeopt_amd.sample_gp_function(k, bounds, m["noise_var"], m["num_samples"],
                                                   interpolation=interp, mean_function=mf)
                key = "%s_%s_%s" % (tag, interp, "mean" if mean else "nomean")
                assert np.array_equal(f.nodes, z[key + "_nodes"])
                np.testing.assert_allclose(f.values, z[tag + "_output"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(f(xq, noise=False), z[key + "_clean"], rtol=0, atol=1e-9)
                np.testing.assert_allclose(f(xq), z[key + "_noisy"], rtol=0, atol=1e-9)
                np.testing.assert_allclose(f(xq[:7]), z[key + "_noisy2"], rtol=0, atol=1e-9)
    with pytest.raises(ValueError):
        safeopt_amd.sample_gp_function(k, bounds, 0.1, 5, interpolation="cubic")


def test_q_mirror_writes_reach_the_device_and_the_setter_uploads():
    """``opt.Q`` can be mutated in place as in the reference (``gp_opt.py:374-390,
    475-476``): an element-wise write into the mirror is uploaded before the next
    pass that reads the intervals, and gives what assigning the whole array gives.
    ``S`` / ``M`` / ``G`` are writable like the reference's arrays (``gp_opt.py:481,
    505-506, 511, 615``): ``get_new_query_point`` reads an edited mask, ``compute_sets``
    recomputes all three (``S`` from the intervals, as ``compute_safe_set`` does)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import safeopt_amd
    from oracle import gp_numpy as gpn
    from _golden import load, make_kernel
    from _oracle_backend import OracleGridBackend, use_oracle_backend
    use_oracle_backend()
    z, meta = load("ties_1d_seed0")

    def make():
        gp = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                              noise_var=meta["noise_vars"][0])
        opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"])
        opt.update_confidence_intervals()
        return opt
    opt = make()
    # masks edited in place: the arg-max of get_new_query_point is taken over the EDITED
    # M | G (gp_opt.py:635-649), as in the reference where the arrays are live
    opt.compute_sets()
    ref = make(); ref.compute_sets()
    rows = np.flatnonzero(np.asarray(ref.M | ref.G))
    first = ref.get_new_query_point()
    keep = rows[~np.all(z["parameter_set"][rows] == first, axis=1)]
    opt.M[:] = False
    opt.G[:] = False
    opt.M[keep] = True
    x2 = opt.get_new_query_point()
    Qh = np.asarray(ref.Q)
    val = np.max((Qh[:, 1::2] - Qh[:, ::2]) / ref.scaling, axis=1)
    assert np.array_equal(x2, z["parameter_set"][keep[np.argmax(val[keep])]])
    assert not np.array_equal(x2, first) or keep.size == rows.size
    # an edited S: nothing safe -> the reference's error; compute_sets recomputes S from Q
    opt.S[:] = False
    with pytest.raises(EnvironmentError):
        opt.get_new_query_point()
    opt.compute_sets()
    for name in ("S", "M", "G"):
        assert np.array_equal(getattr(opt, name), getattr(ref, name))
    assert np.array_equal(opt.get_new_query_point(), first)
    new = np.array(opt.Q) + 0.25
    opt.Q = new
    assert np.array_equal(opt.Q, new)
    # in place: a slice, a fancy index and a view of a view
    a, b = make(), make()
    target = np.array(a.Q)
    target[:, 0] -= 0.5
    target[[3, 5], 1] = 7.0
    target[10:20][:, 1] += 0.125
    a.Q[:, 0] -= 0.5
    a.Q[[3, 5], 1] = 7.0
    a.Q[10:20][:, 1] += 0.125
    b.Q = target
    assert np.array_equal(a.Q, target)
    a.compute_sets(); b.compute_sets()
    for name in ("S", "M", "G"):
        assert np.array_equal(getattr(a, name), getattr(b, name))
    assert np.array_equal(a.get_new_query_point(), b.get_new_query_point())
    # a fresh sweep overwrites hand-made intervals, written either way
    a.Q[:, 0] = -9.0
    a.update_confidence_intervals()
    assert np.array_equal(a.Q, make().Q)


def test_q_writes_into_copies_and_stale_views():
    """  Only writes that land in the mirror of ``Q`` may be flushed
    to the device: a copy of ``opt.Q`` or an array computed from it inherits the view
    class, not the memory; and a view taken BEFORE a sweep is live like the
    reference's array -- a write through it lands on the fresh intervals."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import safeopt_amd
    from oracle import gp_numpy as gpn
    from _oracle_backend import use_oracle_backend
    use_oracle_backend()
    rng = np.random.default_rng(5)
    X = rng.uniform(-2, 2, size=(6, 1)); Y = np.cos(X) + 1.0
    grid = safeopt_amd.linearly_spaced_combinations([(-3., 3.)], 50)
    gp = gpn.GPRegression(X, Y, gpn.RBF(1, 2., 1.), noise_var=0.05 ** 2)
    opt = safeopt_amd.SafeOpt(gp, grid, 0., threshold=0.1)
    opt.optimize()
    w = opt.Q[:, 1] - opt.Q[:, 0]            # derived array
    c = opt.Q.copy()                         # copy
    held = opt.Q                             # a view of the mirror itself
    opt.add_new_data_point(np.array([[0.5]]), np.array([[1.7]]))
    opt.update_confidence_intervals()
    fresh = np.array(opt._backend.download(safeopt_amd._hip.Q) if False else opt.Q)
    w[...] = 0.0; c[...] = -1.0
    assert not opt._q_written                # neither touched the mirror
    opt.compute_sets()
    assert np.array_equal(np.asarray(opt.Q), fresh)
    # a write through the view held from before the sweep: lands on the FRESH intervals
    held[3, 0] = -5.0
    assert opt._q_written
    expect = fresh.copy(); expect[3, 0] = -5.0
    opt.compute_sets()
    assert np.array_equal(np.asarray(opt.Q), expect)
    assert not opt.S[3]


def test_pass_sizes_follow_the_number_of_observations():
    """SafeOpt._pass_size: the first big pass of the expander loop keeps the candidates' operands
    (4 n^2 flop each) at about 1e9 flop, the following ones take 8 times as many, up to 8192;
    an explicit ``pass_sizes`` wins."""
    from types import SimpleNamespace as NS
    def sizes(n, pass_sizes=None, lip=False):
        o = NS(pass_sizes=pass_sizes, use_lipschitz=lip,
               gps=[NS(X=np.zeros((3, 2))), NS(X=np.zeros((n, 2)))])
        return [safeopt_amd.SafeOpt._pass_size(o, k) for k in range(4)]
    assert sizes(217) == [8192, 8192, 8192, 8192]
    assert sizes(637) == [1024, 8192, 8192, 8192]
    assert sizes(1000) == [256, 2048, 8192, 8192]
    assert sizes(2000) == [256, 2048, 8192, 8192]
    assert sizes(2000, (300, 1000)) == [300, 1000, 1000, 1000]
    assert sizes(2000, lip=True) == [8192] * 4         # (Lipschitz certificates: no operands)
