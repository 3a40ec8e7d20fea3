"""Pin oracle/safeopt_numpy.py against the golden vectors produced by the
reference's own gp_opt.py (tests/golden/make_golden.py) -- CPU only."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from oracle import gp_numpy as gpn
from oracle import safeopt_numpy as son
from _golden import load, make_kernel

CASES = ["safeopt_1d_rbf", "safeopt_2d_rbf", "safeopt_1d_multi",
         "safeopt_2d_mat52_g3", "safeopt_1d_lipschitz", "safeopt_context",
         "safeopt_2d_ucb"]


def build_gps(z, meta, t):
    gps = []
    for i, spec in enumerate(meta["kernels"]):
        gps.append(gpn.GPRegression(z["it%d_X%d" % (t, i)], z["it%d_Y%d" % (t, i)],
                                    make_kernel(gpn, spec), noise_var=meta["noise_vars"][i]))
    return gps


def inputs_for(z, meta, t):
    ps = z["parameter_set"]
    nc = meta["num_contexts"]
    if nc:
        ctx = z["it%d_context" % t]
        return np.hstack([ps, np.broadcast_to(ctx, (ps.shape[0], nc))])
    return ps


@pytest.mark.parametrize("name", CASES)
def test_replay_golden(name):
    z, meta = load(name)
    for t in meta["recorded"]:
        gps = build_gps(z, meta, t)
        inputs = inputs_for(z, meta, t)
        beta = float(z["beta_all"][t])
        lip = meta["lipschitz"]
        idx, Q, S, M, G = son.optimize_grid(gps, inputs, meta["fmin"], meta["scaling"],
                                            meta["threshold"], beta, lipschitz=lip,
                                            ucb=meta["ucb"])
        assert_allclose(Q, z["it%d_Q" % t], rtol=0, atol=1e-12)
        assert_array_equal(S, z["it%d_S" % t])
        if not meta["ucb"]:
            assert_array_equal(M, z["it%d_M" % t])
            assert_array_equal(G, z["it%d_G" % t])
        nc = meta["num_contexts"]
        x = inputs[idx, :-nc] if nc else inputs[idx]
        assert_array_equal(x, z["it%d_x_next" % t])
        mi = son.maximum_index(Q, S)
        assert_array_equal(inputs[mi, :-nc or None], z["it%d_max_x" % t])
        assert_allclose(Q[mi, 0], z["it%d_max_l" % t], rtol=0, atol=1e-12)


def test_full_sets_golden():
    z, meta = load("safeopt_full_sets")
    gp = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                          noise_var=meta["noise_vars"][0])
    Q = son.confidence_intervals([gp], z["parameter_set"], meta["beta"])
    S, M, G = son.compute_sets([gp], z["parameter_set"], Q, meta["fmin"], meta["scaling"],
                               meta["threshold"], meta["beta"], full_sets=True)
    assert_allclose(Q, z["Q"], atol=1e-12, rtol=0)
    assert_array_equal(S, z["S"]); assert_array_equal(M, z["M"]); assert_array_equal(G, z["G"])
    assert G.sum() > 1            # the plotting mode returns every expander


def test_swarm_fitness_golden():
    z, meta = load("swarm_2d_g2")
    gps = [gpn.GPRegression(z["X0"], z["Y0"][:, [i]], make_kernel(gpn, meta["kernels"][i]),
                            noise_var=meta["noise_vars"][i]) for i in range(2)]
    for st in ["greedy", "maximizers", "expanders", "safe_set"]:
        v, s = son.swarm_fitness(gps, z["particles"], st, meta["beta"], meta["fmin"],
                                 meta["scaling"], meta["fit_best_lower_bound"])
        assert_allclose(v, z["fit_%s_values" % st], rtol=1e-12, atol=1e-12)
        assert_array_equal(s, z["fit_%s_safe" % st])


def test_edge_cases():
    # empty safe set: M=G=False and the query raises (gp_opt.py:504-507, 631-632)
    gp = gpn.GPRegression([[0.]], [[-1.]], gpn.RBF(1), noise_var=1e-4)
    grid = np.linspace(-1, 1, 50)[:, None]
    Q = son.confidence_intervals([gp], grid, 2.)
    S, M, G = son.compute_sets([gp], grid, Q, [0.], [1.], 0., 2.)
    assert not S.any() and not M.any() and not G.any()
    with pytest.raises(EnvironmentError):
        son.query_index(Q, S, M, G, [1.])
    assert son.maximum_index(Q, S) is None
    # penalty is piecewise (gp_opt.py:874-899)
    s = np.array([0.5, -0.0005, -0.05, -0.5, -2.])
    assert_allclose(son.swarm_penalty(s), [0., -0.001, -0.25, -5., -1200.])
    # everything safe: no unsafe point -> no expander can be certified
    gp = gpn.GPRegression([[0.]], [[5.]], gpn.RBF(1, variance=0.01), noise_var=1e-4)
    grid = np.linspace(-0.01, 0.01, 20)[:, None]
    Q = son.confidence_intervals([gp], grid, 2.)
    S, M, G = son.compute_sets([gp], grid, Q, [0.], [0.1], 0., 2.)
    assert S.all() and not G.any() and M.any()


@pytest.mark.parametrize("name", ["sets_1d_seed0", "sets_1d_seed7", "sets_1d_g2_seed0",
                                  "sets_1d_g2_seed7", "sets_2d_seed3"])
def test_expander_loop_golden(name):
    """Scenarios where the reference's expander loop runs 1..23 checks."""
    z, meta = load(name)
    gps = [gpn.GPRegression(z["X%d" % i], z["Y%d" % i], make_kernel(gpn, spec),
                            noise_var=meta["noise_vars"][i])
           for i, spec in enumerate(meta["kernels"])]
    grid = z["parameter_set"]
    Q = son.confidence_intervals(gps, grid, meta["beta"])
    assert_allclose(Q, z["Q"], rtol=0, atol=1e-12)
    S, M, G, trace = son.compute_sets(gps, grid, Q, meta["fmin"], meta["scaling"],
                                      meta["threshold"], meta["beta"], return_trace=True)
    assert_array_equal(S, z["S"]); assert_array_equal(M, z["M"]); assert_array_equal(G, z["G"])
    assert len(trace) >= 1 and G.sum() == 1
    idx = son.query_index(Q, S, M, G, meta["scaling"])
    assert_array_equal(grid[idx], z["x_next"])


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_tied_widths_golden(seed):
    """Exact ties in the candidate widths: the oracle runs the reference's own
    ``argsort()[::-1]`` expression, so on the NumPy that wrote the fixtures it
    reproduces which tied candidate the reference marked (gp_opt.py:542-552)."""
    z, meta = load("ties_1d_seed%d" % seed)
    go = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                          noise_var=meta["noise_vars"][0])
    S, M, G, trace = son.compute_sets([go], z["parameter_set"], z["Q"], meta["fmin"],
                                      meta["scaling"], meta["threshold"], meta["beta"],
                                      return_trace=True)
    assert np.array_equal(S, z["S"]) and np.array_equal(M, z["M"])
    assert np.array_equal(G, z["G"]) and len(trace) == int(z["n_checks"])
    wd = z["Q"][:, 1] - z["Q"][:, 0]
    s = S & ~M
    assert int(np.sum(wd[s] == wd[s].max())) == int(z["n_tied_top"]) > 1


@pytest.mark.parametrize("kind", ["RBF", "Matern52"])
def test_rank1_expander_form_equals_the_refit_form(kind):
    """``son.expander_hits_rank1`` -- the closed form the big-pass GPU tests check thousands of
    candidates with -- against the literal append / predict / pop of gp_opt.py:585-606 on a
    problem small enough to refit per candidate: same hit per candidate, and the updated
    lower bounds agree to 1e-9."""
    rng = np.random.default_rng(11)
    X = rng.uniform(-2, 2, size=(25, 2))
    Y = np.sin(X[:, :1]) + 0.5 * np.cos(2 * X[:, 1:]) + 0.6
    kern = getattr(gpn, kind)(2, variance=1.5, lengthscale=[0.8, 1.1], ARD=True)
    gp = gpn.GPRegression(X, Y, kern, noise_var=0.03 ** 2)
    side = 30
    g = np.linspace(-3, 3, side)
    grid = np.array([(a, b) for a in g for b in g])
    beta, fmin = 2.0, 0.0
    Q = son.confidence_intervals([gp], grid, beta)
    S = son.safe_set(Q, [fmin])
    assert S.any() and (~S).any()
    cand = np.flatnonzero(S)[::3]
    fast = son.expander_hits_rank1(gp, grid, ~S, cand, Q[cand, 1], beta, fmin, chunk=17)
    slow = np.zeros(cand.size, dtype=bool)
    margin = np.zeros(cand.size)
    for k, idx in enumerate(cand):
        son._append_point(gp, grid[[idx]], np.atleast_2d(Q[idx, 1]))
        m2, v2 = gp.predict_noiseless(grid[~S])
        son._pop_point(gp)
        l2 = m2.squeeze() - beta * np.sqrt(v2.squeeze())
        slow[k] = np.any(l2 >= fmin)
        margin[k] = np.max(l2) - fmin
    # (a candidate whose best row sits within rounding of fmin may go either way)
    clear = np.abs(margin) > 1e-9
    assert clear.sum() > 0.9 * cand.size
    assert_array_equal(fast[clear], slow[clear])
    assert fast.any() and not fast.all()
