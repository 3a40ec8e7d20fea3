#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box has no
reference tree and only consumes the committed ``*.npz``)::

    python tests/golden/make_golden.py

How: the reference's ``safeopt`` package (``/root/reference/safeopt``) is
imported unmodified, with
  * two compatibility shims for Python 3.10 / NumPy 2 (``collections.Sequence``
    and ``np.float``), and
  * ``sys.modules['GPy']`` pointing at ``oracle.gp_numpy`` (GPy is a
    third-party dependency that is neither vendored nor installable here).
The reference's own ``SafeOpt`` / ``SafeOptSwarm`` code then runs on small
seeded problems; inputs and outputs are stored as plain arrays.  The vectors
therefore pin the reference's L3 logic (confidence intervals -> S/M/G ->
chosen point, swarm fitness, RNG consumption order) given the oracle's GP
arithmetic.  No reference source text is stored -- only numbers.
"""

from __future__ import print_function

import collections
import collections.abc
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

# ---- shims + fake GPy -----------------------------------------------------
collections.Sequence = collections.abc.Sequence
if not hasattr(np, "float"):
    np.float = float

from oracle import gp_numpy  # noqa: E402

fake = types.ModuleType("GPy")
fake.kern = types.SimpleNamespace(RBF=gp_numpy.RBF, Matern32=gp_numpy.Matern32,
                                  Matern52=gp_numpy.Matern52)
fake.models = types.SimpleNamespace(GPRegression=gp_numpy.GPRegression)
sys.modules["GPy"] = fake
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference")
import safeopt as ref  # noqa: E402  (the reference package itself)

GPy = fake


def kernel_spec(k):
    """Serialise a kernel as a list of parts (kind, variance, ls, dims)."""
    parts = k.parts if isinstance(k, gp_numpy.Prod) else [k]
    out = []
    for p in parts:
        out.append(dict(kind=type(p).__name__, variance=float(p.variance[0]),
                        lengthscale=[float(v) for v in p.lengthscale],
                        ARD=bool(p.ARD), input_dim=int(p.input_dim),
                        active_dims=[int(v) for v in p.active_dims]))
    return out


def save(name, **arrs):
    import json
    meta = arrs.pop("meta")
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrs)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024.))


def smooth_fun(x, seed, shift=0.0):
    """Deterministic smooth test objective (sum of RBF bumps)."""
    rng = np.random.default_rng(seed)
    d = x.shape[1]
    c = rng.uniform(-4, 4, size=(12, d))
    w = rng.normal(size=12)
    r2 = ((x[:, None, :] - c[None, :, :]) ** 2).sum(-1)
    return (np.exp(-0.5 * r2 / 2.0) * w).sum(1)[:, None] + shift


def run_safeopt_case(name, gps, kernels, noise_vars, parameter_set, fmin, fun,
                     n_iter, record_at, lipschitz=None, threshold=0.2,
                     num_contexts=0, contexts=None, beta=2., ucb=False,
                     scaling='auto'):
    opt = ref.SafeOpt(gps if len(gps) > 1 else gps[0], parameter_set, fmin,
                      lipschitz=lipschitz, beta=beta, threshold=threshold,
                      num_contexts=num_contexts, scaling=scaling)
    arrs = dict(parameter_set=np.ascontiguousarray(parameter_set))
    x_next_all, recorded, beta_all = [], [], []
    for t in range(n_iter):
        beta_all.append(float(opt.beta(opt.t)))
        ctx = None if contexts is None else contexts[t % len(contexts)]
        Xs = [g.X.copy() for g in opt.gps]
        Ys = [g.Y.copy() for g in opt.gps]
        x_next = opt.optimize(context=ctx, ucb=ucb)
        if t in record_at:
            for i in range(len(opt.gps)):
                arrs["it%d_X%d" % (t, i)] = Xs[i]
                arrs["it%d_Y%d" % (t, i)] = Ys[i]
            arrs["it%d_Q" % t] = opt.Q.copy()
            arrs["it%d_S" % t] = opt.S.copy()
            arrs["it%d_M" % t] = opt.M.copy()
            arrs["it%d_G" % t] = opt.G.copy()
            arrs["it%d_x_next" % t] = np.asarray(x_next).copy()
            arrs["it%d_t" % t] = np.array(opt.t)
            mx = opt.get_maximum(context=ctx)
            if mx is not None:
                arrs["it%d_max_x" % t] = np.asarray(mx[0]).copy()
                arrs["it%d_max_l" % t] = np.asarray(mx[1]).copy()
            if ctx is not None:
                arrs["it%d_context" % t] = np.asarray(ctx, dtype=float)
            recorded.append(t)
        x_next_all.append(np.asarray(x_next).copy())
        xq = np.atleast_2d(x_next)
        if ctx is not None:
            xq = np.hstack([xq, np.atleast_2d(ctx)])
        y = fun(xq)
        opt.add_new_data_point(x_next, y, context=ctx)
    arrs["x_next_all"] = np.array(x_next_all)
    arrs["beta_all"] = np.array(beta_all)
    meta = dict(kernels=[kernel_spec(k) for k in kernels],
                noise_vars=[float(v) for v in noise_vars],
                fmin=[float(v) for v in np.atleast_1d(opt.fmin)],
                scaling=[float(v) for v in opt.scaling],
                threshold=threshold,
                beta=None if callable(beta) else beta,
                lipschitz=None if lipschitz is None else
                [float(v) for v in np.atleast_1d(opt.liptschitz)],
                num_contexts=num_contexts, ucb=ucb, recorded=recorded,
                n_iter=n_iter)
    save(name, meta=meta, **arrs)


def case_1d():
    # BASELINE.json configs[0]: 1D RBF, 1 constraint, 1000-point grid,
    # examples/1d_example.ipynb cells 2+4 constants; 20 training points.
    bounds = [(-10., 10.)]
    ps = ref.linearly_spaced_combinations(bounds, 1000)
    k = GPy.kern.RBF(input_dim=1, variance=2., lengthscale=1.0, ARD=True)
    nv = 0.05 ** 2
    x0 = np.zeros((1, 1))
    f = lambda x: 0.7 * (smooth_fun(x, seed=0) - smooth_fun(np.zeros((1, 1)), seed=0)) + 1.0
    gp = GPy.models.GPRegression(x0, f(x0), k, noise_var=nv)
    run_safeopt_case("safeopt_1d_rbf", [gp], [k], [nv], ps, 0., f, n_iter=20,
                     record_at=[0, 1, 4, 9, 19])


def case_2d():
    bounds = [(-5., 5.), (-5., 5.)]
    ps = ref.linearly_spaced_combinations(bounds, 40)
    k = GPy.kern.RBF(input_dim=2, variance=2., lengthscale=[1.0, 1.5], ARD=True)
    nv = 0.05 ** 2
    rng = np.random.default_rng(1)
    x0 = rng.uniform(-1, 1, size=(6, 2))
    f = lambda x: smooth_fun(x, seed=11) - smooth_fun(np.zeros((1, 2)), seed=11) + 1.0
    gp = GPy.models.GPRegression(x0, f(x0), k, noise_var=nv)
    run_safeopt_case("safeopt_2d_rbf", [gp], [k], [nv], ps, 0., f, n_iter=12,
                     record_at=[0, 3, 11])


def case_multi():
    # 1d_multiple_constraints_example.ipynb shape: objective without a
    # constraint (fmin=-inf) + one Matern-5/2 constraint, tiny second noise.
    bounds = [(-10., 10.)]
    ps = ref.linearly_spaced_combinations(bounds, 500)
    k1 = GPy.kern.RBF(input_dim=1, variance=2., lengthscale=1.0, ARD=True)
    k2 = GPy.kern.Matern52(input_dim=1, variance=1.5, lengthscale=2.0)
    nv1, nv2 = 0.05 ** 2, 1e-5
    x0 = np.zeros((1, 1))
    f1 = lambda x: smooth_fun(x, seed=3)
    f2 = lambda x: smooth_fun(x, seed=4) - smooth_fun(np.zeros((1, 1)), seed=4) + 1.2
    f = lambda x: np.hstack([f1(x), f2(x)])
    y0 = f(x0)
    gp1 = GPy.models.GPRegression(x0, y0[:, 0, None], k1, noise_var=nv1)
    gp2 = GPy.models.GPRegression(x0, y0[:, 1, None], k2, noise_var=nv2)
    run_safeopt_case("safeopt_1d_multi", [gp1, gp2], [k1, k2], [nv1, nv2], ps,
                     [-np.inf, 0.], f, n_iter=10, record_at=[0, 2, 9],
                     threshold=0.1)


def case_three():
    # BASELINE.json configs[2] twin: 2D Matern-5/2, 3 constraints (G=3)
    bounds = [(-5., 5.), (-5., 5.)]
    ps = ref.linearly_spaced_combinations(bounds, 30)
    ks = [GPy.kern.Matern52(input_dim=2, variance=2., lengthscale=[1.0, 1.0],
                            ARD=True) for _ in range(3)]
    nv = 0.05 ** 2
    rng = np.random.default_rng(2)
    x0 = rng.uniform(-1.5, 1.5, size=(10, 2))
    fs = [lambda x, s=s: smooth_fun(x, seed=s) - smooth_fun(np.zeros((1, 2)), seed=s) + 1.0
          for s in (21, 22, 23)]
    f = lambda x: np.hstack([g(x) for g in fs])
    y0 = f(x0)
    gps = [GPy.models.GPRegression(x0, y0[:, i, None], ks[i], noise_var=nv)
           for i in range(3)]
    run_safeopt_case("safeopt_2d_mat52_g3", gps, ks, [nv] * 3, ps, [0., 0., 0.],
                     f, n_iter=8, record_at=[0, 4, 7])


def case_lipschitz():
    bounds = [(-10., 10.)]
    ps = ref.linearly_spaced_combinations(bounds, 400)
    k = GPy.kern.Matern32(input_dim=1, variance=2., lengthscale=1.5)
    nv = 0.05 ** 2
    x0 = np.zeros((1, 1))
    f = lambda x: smooth_fun(x, seed=5) - smooth_fun(np.zeros((1, 1)), seed=5) + 1.0
    gp = GPy.models.GPRegression(x0, f(x0), k, noise_var=nv)
    run_safeopt_case("safeopt_1d_lipschitz", [gp], [k], [nv], ps, 0., f,
                     n_iter=10, record_at=[0, 5, 9], lipschitz=1.5)


def case_context():
    # context_example.ipynb cells 2+4: product of two RBFs on disjoint columns
    ps = ref.linearly_spaced_combinations([(-5., 5.)], 300)
    kp = GPy.kern.RBF(input_dim=1, variance=2., lengthscale=1.0, active_dims=[0])
    kc = GPy.kern.RBF(input_dim=1, variance=2., lengthscale=1.0,
                      active_dims=[1], name='context')
    k = kp * kc
    nv = 0.05 ** 2
    x = np.array([[0., 0.]])
    f = lambda x: smooth_fun(x, seed=7) - smooth_fun(np.zeros((1, 2)), seed=7) + 1.0
    gp = GPy.models.GPRegression(x, f(x), k, noise_var=nv)
    ctxs = [np.array([[0.]]), np.array([[0.1]]), np.array([[-0.2]])]
    run_safeopt_case("safeopt_context", [gp], [k], [nv], ps, 0., f, n_iter=9,
                     record_at=[0, 4, 8], threshold=0.5, num_contexts=1,
                     contexts=ctxs)


def case_ucb():
    bounds = [(-5., 5.), (-5., 5.)]
    ps = ref.linearly_spaced_combinations(bounds, 25)
    k = GPy.kern.RBF(input_dim=2, variance=2., lengthscale=1.0, ARD=True)
    nv = 0.05 ** 2
    x0 = np.zeros((1, 2))
    f = lambda x: smooth_fun(x, seed=9) - smooth_fun(np.zeros((1, 2)), seed=9) + 1.0
    gp = GPy.models.GPRegression(x0, f(x0), k, noise_var=nv)
    run_safeopt_case("safeopt_2d_ucb", [gp], [k], [nv], ps, 0., f, n_iter=8,
                     record_at=[0, 7], ucb=True, beta=lambda t: 2. + 0.05 * t)


def case_full_sets():
    bounds = [(-10., 10.)]
    ps = ref.linearly_spaced_combinations(bounds, 200)
    k = GPy.kern.RBF(input_dim=1, variance=2., lengthscale=1.0, ARD=True)
    nv = 0.05 ** 2
    X = np.array([[-1.0], [0.], [0.8]])
    f = lambda x: smooth_fun(x, seed=13) - smooth_fun(np.zeros((1, 1)), seed=13) + 1.0
    gp = GPy.models.GPRegression(X, f(X), k, noise_var=nv)
    opt = ref.SafeOpt(gp, ps, 0., threshold=0.2)
    opt.update_confidence_intervals()
    opt.compute_sets(full_sets=True)
    save("safeopt_full_sets", meta=dict(kernels=[kernel_spec(k)],
         noise_vars=[nv], fmin=[0.], scaling=[float(opt.scaling[0])],
         threshold=0.2, beta=2.),
         parameter_set=ps, X0=X, Y0=gp.Y.copy(), Q=opt.Q.copy(),
         S=opt.S.copy(), M=opt.M.copy(), G=opt.G.copy())


def case_sets():
    """Direct compute_sets() scenarios that exercise the expander loop
    (gp_opt.py:557-612): first candidate is an expander, the 22nd is, 2 GPs,
    2-D.  ``n_checks`` counts the reference's own gp.set_XY calls / 2."""
    def one(tag, gps, ks, nvs, grid, fmin, scaling, thr):
        opt = ref.SafeOpt(gps if len(gps) > 1 else gps[0], grid, fmin,
                          threshold=thr, scaling=scaling)
        opt.update_confidence_intervals()
        calls = [0]
        for g in opt.gps:
            orig = g.set_XY
            def counted(X, Y, orig=orig):
                calls[0] += 1
                return orig(X, Y)
            g.set_XY = counted
        opt.compute_sets()
        x = opt.get_new_query_point()
        arrs = dict(parameter_set=grid, Q=opt.Q.copy(), S=opt.S.copy(),
                    M=opt.M.copy(), G=opt.G.copy(), x_next=np.asarray(x).copy(),
                    n_checks=np.array(calls[0] // 2))
        for i, g in enumerate(opt.gps):
            arrs["X%d" % i] = g.X.copy()
            arrs["Y%d" % i] = g.Y.copy()
        save(tag, meta=dict(kernels=[kernel_spec(k) for k in ks], noise_vars=nvs,
             fmin=[float(v) for v in opt.fmin],
             scaling=[float(v) for v in opt.scaling], threshold=thr, beta=2.),
             **arrs)

    grid1 = ref.linearly_spaced_combinations([(-10., 10.)], 600)
    for seed in (0, 7):
        rng = np.random.default_rng(seed)
        X = np.concatenate([rng.normal(0, 0.3, 8), rng.uniform(-4, 4, 6)])[:, None]
        f1 = lambda x: 1.0 + 2.0 * np.exp(-x[:, :1] ** 2 / 2.0) + 0.3 * np.sin(2 * x[:, :1])
        f2 = lambda x: 1.5 + 0.5 * np.cos(0.7 * x[:, :1])
        Y1 = f1(X) + 0.05 * rng.normal(size=(X.shape[0], 1))
        Y2 = f2(X) + 0.02 * rng.normal(size=(X.shape[0], 1))
        k1 = GPy.kern.RBF(1, variance=2., lengthscale=1.0, ARD=True)
        k2 = GPy.kern.Matern52(1, variance=1.0, lengthscale=2.0)
        g1 = GPy.models.GPRegression(X, Y1, k1, noise_var=0.05 ** 2)
        one("sets_1d_seed%d" % seed, [g1], [k1], [0.05 ** 2], grid1, 0., 'auto', 0.)
        g1 = GPy.models.GPRegression(X, Y1, k1, noise_var=0.05 ** 2)
        g2 = GPy.models.GPRegression(X, Y2, k2, noise_var=0.02 ** 2)
        one("sets_1d_g2_seed%d" % seed, [g1, g2], [k1, k2], [0.05 ** 2, 0.02 ** 2],
            grid1, [0., 0.5], 'auto', 0.05)
    rng = np.random.default_rng(3)
    grid2 = ref.linearly_spaced_combinations([(-5., 5.), (-5., 5.)], 35)
    X = np.vstack([rng.normal(0, 0.25, (10, 2)), rng.uniform(-3, 3, (10, 2))])
    Y = 1.0 + 2.5 * np.exp(-(X ** 2).sum(1, keepdims=True)) + 0.05 * rng.normal(size=(20, 1))
    k = GPy.kern.RBF(2, variance=2., lengthscale=[1., 1.2], ARD=True)
    g = GPy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
    one("sets_2d_seed3", [g], [k], [0.05 ** 2], grid2, 0., 'auto', 0.)


def case_swarm():
    # SafeOptSwarm: 2 GPs, swarm_size 20
    bounds = [(-5., 5.), (-5., 5.)]
    k1 = GPy.kern.RBF(input_dim=2, variance=2., lengthscale=1.0, ARD=True)
    k2 = GPy.kern.Matern52(input_dim=2, variance=1.0, lengthscale=[1.5, 1.0],
                           ARD=True)
    nv = 0.05 ** 2
    rng = np.random.default_rng(17)
    x0 = np.vstack([np.zeros((1, 2)), rng.uniform(-0.7, 0.7, size=(4, 2))])
    f1 = lambda x: smooth_fun(x, seed=31) - smooth_fun(np.zeros((1, 2)), seed=31) + 1.0
    f2 = lambda x: smooth_fun(x, seed=32) - smooth_fun(np.zeros((1, 2)), seed=32) + 1.5
    f = lambda x: np.hstack([f1(x), f2(x)])
    y0 = f(x0)
    gp1 = GPy.models.GPRegression(x0, y0[:, 0, None], k1, noise_var=nv)
    gp2 = GPy.models.GPRegression(x0, y0[:, 1, None], k2, noise_var=nv)
    opt = ref.SafeOptSwarm([gp1, gp2], [0., 0.2], bounds=bounds, threshold=0.2)
    arrs = dict(X0=x0, Y0=y0, optimal_velocities=opt.optimal_velocities.copy())
    # fitness of fixed particles for every swarm type
    parts = np.random.default_rng(18).uniform(-3, 3, size=(64, 2))
    arrs["particles"] = parts
    opt.best_lower_bound = 0.35
    for st in ['greedy', 'maximizers', 'expanders', 'safe_set']:
        v, s = opt._compute_particle_fitness(st, parts.copy())
        arrs["fit_%s_values" % st] = np.asarray(v, dtype=float).copy()
        arrs["fit_%s_safe" % st] = np.asarray(s, dtype=bool).copy()
    opt.best_lower_bound = -np.inf
    # full optimize() iterations with a pinned global RNG
    np.random.seed(1234)
    xs, ys = [], []
    for t in range(4):
        x = opt.optimize()
        xs.append(np.asarray(x).copy())
        arrs["opt%d_S" % t] = opt.S.copy()
        arrs["opt%d_greedy_point" % t] = np.asarray(opt.greedy_point).copy()
        arrs["opt%d_best_lower_bound" % t] = np.array(opt.best_lower_bound)
        y = f(np.atleast_2d(x))
        ys.append(np.asarray(y).ravel().copy())
        opt.add_new_data_point(x, y)
    arrs["opt_x"] = np.array(xs)
    arrs["opt_y"] = np.array(ys)
    save("swarm_2d_g2", meta=dict(kernels=[kernel_spec(k1), kernel_spec(k2)],
         noise_vars=[nv, nv], fmin=[0., 0.2], bounds=bounds, threshold=0.2,
         beta=2., swarm_size=20, seed=1234,
         scaling=[float(v) for v in opt.scaling], fit_best_lower_bound=0.35),
         **arrs)


def case_ties():
    """Exact ties in the candidate widths (gp_opt.py:542-552): the intervals are
    ASSIGNED (opt.Q[:] = quantised values, as a user may do) so that many
    candidates share the largest width bit for bit; the reference's own
    argsort()[::-1] then decides which tied candidate is visited -- and found to
    be an expander -- first.  Recorded: the quantised Q and the resulting sets."""
    grid = ref.linearly_spaced_combinations([(-6., 6.)], 400)
    # (seed, quantum): chosen so that the winner is NOT always the highest tied
    # index (what a stable sort reversed would give): 4 of the 6 discriminate
    for tag, (seed, q) in enumerate(((0, 4.0), (1, 4.0), (11, 2.0), (11, 4.0),
                                     (0, 2.0), (2, 2.0))):
        rng = np.random.default_rng(100 + seed)
        X = np.concatenate([rng.normal(0, 0.4, 6), rng.uniform(-3, 3, 5)])[:, None]
        Y = 1.2 + 1.5 * np.exp(-X ** 2 / 3.0) + 0.05 * rng.normal(size=X.shape)
        k = GPy.kern.RBF(1, variance=2., lengthscale=1.0, ARD=True)
        gp = GPy.models.GPRegression(X, Y, k, noise_var=0.05 ** 2)
        opt = ref.SafeOpt(gp, grid, 0., threshold=0.)
        opt.update_confidence_intervals()
        opt.Q[:] = np.round(opt.Q * q) / q           # exact ties by construction
        calls = [0]
        orig = gp.set_XY
        def counted(X, Y, orig=orig):
            calls[0] += 1
            return orig(X, Y)
        gp.set_XY = counted
        opt.compute_sets()
        x = opt.get_new_query_point()
        s = opt.S & ~opt.M
        widths = (opt.Q[:, 1] - opt.Q[:, 0])
        save("ties_1d_seed%d" % tag, meta=dict(
            kernels=[kernel_spec(k)], noise_vars=[0.05 ** 2], fmin=[0.],
            scaling=[float(opt.scaling[0])], threshold=0., beta=2., quantum=q,
            # which tied candidate argsort()[::-1] visits first is NumPy's business:
            # a host with this very NumPy must reproduce G / x_next of the fixture
            numpy_version=np.__version__),
            parameter_set=grid, X0=X, Y0=Y, Q=opt.Q.copy(), S=opt.S.copy(),
            M=opt.M.copy(), G=opt.G.copy(), x_next=np.asarray(x).copy(),
            n_checks=np.array(calls[0] // 2),
            n_tied_top=np.array(int(np.sum(widths[s] == widths[s].max())) if s.any() else 0))


def case_sample_gp_function():
    """utilities.sample_gp_function (utilities.py:57-143) with a pinned global RNG:
    the sampled node values (read out of the closure), noiseless and noisy
    evaluations, kernel and linear interpolation, with and without a mean."""
    arrs, meta = {}, {}
    for tag, k, bounds, ns in (
            ("rbf1", GPy.kern.RBF(1, variance=2., lengthscale=1.0), [(-5., 5.)], 40),
            ("m52_2", GPy.kern.Matern52(2, variance=1.5, lengthscale=[1.0, 1.4], ARD=True),
             [(-2., 2.), (-1., 3.)], [9, 11])):
        xq = np.random.default_rng(5).uniform([b[0] for b in bounds], [b[1] for b in bounds],
                                              size=(25, len(bounds)))
        for interp in ("linear", "kernel"):
            for mean in (None, "lin"):
                mf = None if mean is None else (lambda x: 0.3 * x[:, :1] - 0.1)
                np.random.seed(77)
                seen = {}
                orig_mvn = np.random.multivariate_normal

                def mvn(mean, cov, *a, _o=orig_mvn, _s=seen, **kw):
                    _s["cov"] = np.array(cov)          # what the reference hands over:
                    return _o(mean, cov, *a, **kw)     # kernel.K(nodes) + 1e-6 I
                np.random.multivariate_normal = mvn
                try:
                    f = ref.sample_gp_function(k, bounds, 0.05 ** 2, ns, interpolation=interp,
                                               mean_function=mf)
                finally:
                    np.random.multivariate_normal = orig_mvn
                arrs[tag + "_cov"] = seen["cov"]
                cl = dict(zip(f.__code__.co_freevars, [c.cell_contents for c in f.__closure__]))
                key = "%s_%s_%s" % (tag, interp, "mean" if mean else "nomean")
                arrs[key + "_nodes"] = np.asarray(cl["inputs"]).copy()
                if "output" in cl:           # (the kernel interpolant only keeps alpha;
                    arrs[tag + "_output"] = np.asarray(cl["output"]).copy()   # same seed, same draw)
                arrs[key + "_clean"] = f(xq, noise=False).copy()
                arrs[key + "_noisy"] = f(xq).copy()          # consumes randn(25, 1)
                arrs[key + "_noisy2"] = f(xq[:7]).copy()     # and randn(7, 1)
        arrs[tag + "_xq"] = xq
        meta[tag] = dict(kernel=kernel_spec(k), bounds=[list(b) for b in bounds],
                         num_samples=ns, noise_var=0.05 ** 2, seed=77)
    save("sample_gp_function", meta=meta, **arrs)


def case_gp_sklearn():
    """Independent pin of the GP arithmetic: scikit-learn GPR."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel, RBF, Matern
    rng = np.random.default_rng(40)
    arrs, meta = {}, {}
    for tag, kind, nu in [("rbf", "RBF", None), ("m32", "Matern32", 1.5),
                          ("m52", "Matern52", 2.5)]:
        n, N, d = 60, 400, 2
        X = rng.uniform(-2, 2, size=(n, d))
        Y = smooth_fun(X, seed=41)
        Xs = rng.uniform(-4, 4, size=(N, d))
        var, ls, nv = 2.0, np.array([1.0, 1.7]), 0.05 ** 2
        if nu is None:
            sk = ConstantKernel(var, 'fixed') * RBF(ls, 'fixed')
        else:
            sk = ConstantKernel(var, 'fixed') * Matern(ls, 'fixed', nu=nu)
        gpr = GaussianProcessRegressor(kernel=sk, alpha=nv + 1e-8,
                                       optimizer=None).fit(X, Y)
        mu, std = gpr.predict(Xs, return_std=True)
        arrs.update({tag + "_X": X, tag + "_Y": Y, tag + "_Xs": Xs,
                     tag + "_mean": mu.ravel(), tag + "_var": std.ravel() ** 2})
        meta[tag] = dict(kind=kind, variance=var, lengthscale=list(ls),
                         noise_var=nv)
    save("gp_sklearn", meta=meta, **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1:            # only the named cases
        for name in sys.argv[1:]:
            globals()["case_" + name]()
        sys.exit(0)
    case_1d()
    case_2d()
    case_multi()
    case_three()
    case_lipschitz()
    case_context()
    case_ucb()
    case_full_sets()
    case_sets()
    case_swarm()
    case_ties()
    case_sample_gp_function()
    case_gp_sklearn()
