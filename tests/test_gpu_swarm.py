"""GPU parity, SURVEY.md 8(a) row A7 and next rows f2 / f3: swarm fitness, safe-set growth, the PSO on the device."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal
from _golden import load, make_kernel

from _gpu_common import (  # noqa: F401
    MEAN_TOL, VAR_TOL, mods, smooth, kernels, check_posterior, product_kernel, GOLD, build_opt, _swarm_problem, _grow_reference, kernels_from, _PretendWorld, _PretendWorldPadded, _dev_script)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["classic", "pair"])
def test_swarm_fitness_both_kernels(mods, which):
    """_compute_particle_fitness (gp_opt.py:901-1013) on more particles than the
    few-points path takes, through both sweep kernels."""
    safeopt_amd, gpy, gpn, son = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(11)
    d, P = 3, 5000
    gps, gos = [], []
    for i, n in enumerate([300, 530]):
        X = rng.uniform(-2, 2, size=(n, d)); Y = smooth(X, 9 + i) + 0.2
        gps.append(gpy.models.GPRegression(X, Y, kernels(gpy.kern, "RBF", d), noise_var=0.05 ** 2))
        gos.append(gpn.GPRegression(X, Y, kernels(gpn, "RBF", d), noise_var=0.05 ** 2))
    parts = rng.uniform(-2.5, 2.5, size=(P, d))
    fmin = np.array([0.0, 0.1]); scaling = np.array([1.3, 1.1])
    ctx = gps[0]._fitted().ctx
    old = ctx.set_sweep(which)
    try:
        for st in ["greedy", "maximizers", "expanders", "safe_set"]:
            v, s = _hip.swarm_fitness(ctx, [g._fitted() for g in gps], st, parts, 2.0,
                                      fmin, scaling, 0.4)
            vo, so = son.swarm_fitness(gos, parts, st, 2., fmin, scaling, 0.4)
            assert_array_equal(s, so)
            assert_allclose(v, vo, rtol=1e-7, atol=1e-8)
    finally:
        ctx.set_sweep(old)


def test_swarm_fitness_golden(mods):
    safeopt_amd, gpy, _, _ = mods
    z, meta = load("swarm_2d_g2")
    gps = [gpy.models.GPRegression(z["X0"], z["Y0"][:, [i]], make_kernel(gpy.kern, meta["kernels"][i]),
                                   noise_var=meta["noise_vars"][i]) for i in range(2)]
    opt = safeopt_amd.SafeOptSwarm(gps, meta["fmin"], bounds=[tuple(b) for b in meta["bounds"]],
                                   threshold=meta["threshold"])
    assert_allclose(opt.optimal_velocities, z["optimal_velocities"], rtol=1e-12)
    opt.best_lower_bound = meta["fit_best_lower_bound"]
    for st in ["greedy", "maximizers", "expanders", "safe_set"]:
        v, s = opt._compute_particle_fitness(st, z["particles"].copy())
        assert_allclose(v, z["fit_%s_values" % st], rtol=1e-8, atol=1e-9)
        assert_array_equal(s, z["fit_%s_safe" % st])


@pytest.mark.parametrize("pso", ["device", "host"])
def test_swarm_optimize_golden(mods, pso):
    """Whole SafeOptSwarm.optimize() iterations against the reference run with
    the same NumPy global RNG seed (host RNG order is part of the contract) --
    with the swarm loop on the GPU (default) and with the host loop."""
    safeopt_amd, gpy, _, _ = mods
    z, meta = load("swarm_2d_g2")
    gps = [gpy.models.GPRegression(z["X0"], z["Y0"][:, [i]], make_kernel(gpy.kern, meta["kernels"][i]),
                                   noise_var=meta["noise_vars"][i]) for i in range(2)]
    opt = safeopt_amd.SafeOptSwarm(gps, meta["fmin"], bounds=[tuple(b) for b in meta["bounds"]],
                                   threshold=meta["threshold"], pso=pso)
    np.random.seed(meta["seed"])
    # all four recorded iterations: the measurements of the reference run are fed
    # back, so every later iteration also checks the RNG consumption order across
    # add_new_data_point, the safe-set growth and the greedy-point bookkeeping
    for t in range(z["opt_x"].shape[0]):
        x = opt.optimize()
        assert_allclose(x, z["opt_x"][t], rtol=0, atol=1e-6, err_msg="iteration %d" % t)
        assert_allclose(opt.S, z["opt%d_S" % t], rtol=0, atol=1e-6)
        assert_allclose(opt.greedy_point, z["opt%d_greedy_point" % t], rtol=0, atol=1e-6)
        assert_allclose(opt.best_lower_bound, z["opt%d_best_lower_bound" % t], atol=1e-7)
        opt.add_new_data_point(z["opt_x"][t], z["opt_y"][t][None, :])


@pytest.mark.parametrize("swarm_type", ["greedy", "maximizers", "expanders"])
def test_device_pso_bit_identical_to_host_loop(mods, swarm_type):
    """SwarmOptimization with its state in HBM
    (sgp_swarm_run) against the host loop of swarm.py:61-146, same np.random
    stream: every state array bit-identical, generator left in the same state."""
    host = _swarm_problem(mods, "host")
    dev = _swarm_problem(mods, "device")
    for o in (host, dev):
        o.best_lower_bound = 0.3
    start = np.random.default_rng(3).uniform(-0.5, 0.5, size=(40, 2))
    out = []
    for o in (host, dev):
        np.random.seed(11)
        sw = o.swarms[swarm_type]
        sw.init_swarm(start.copy())
        sw.run_swarm(25)
        out.append((sw.positions.copy(), sw.velocities.copy(), sw.best_positions.copy(),
                    np.array(sw.best_values), np.array(sw.global_best), np.random.rand()))
    for a, b in zip(out[0], out[1]):
        assert_array_equal(a, b)


def test_device_pso_device_rng(mods):
    """rng on the GPU: not NumPy-reproducible by design; check the invariants of
    the algorithm, determinism per seed and that NumPy's stream is untouched."""
    from safeopt_amd import DeviceSwarmOptimization
    np.random.seed(7)        # the generator key is ONE draw from NumPy's stream at construction
    o = _swarm_problem(mods, "device-rng", swarm_size=500)
    o.best_lower_bound = 0.3
    # every swarm of an optimiser (and every optimiser) has its own key
    assert len({o.swarms[t]._seed for t in ("greedy", "maximizers", "expanders")}) == 3
    sw = o.swarms["maximizers"]
    assert isinstance(sw, DeviceSwarmOptimization)
    start = np.random.default_rng(5).uniform(-0.5, 0.5, size=(500, 2))
    np.random.seed(1)
    sw.init_swarm(start.copy())
    v0, _ = o._compute_particle_fitness("maximizers", start)
    assert_allclose(sw.best_values, v0, rtol=1e-12)
    assert np.all((sw.velocities >= 0) & (sw.velocities <= o.optimal_velocities))
    assert len(np.unique(sw.velocities)) > 900            # really random
    sw.run_swarm(20)
    assert np.random.rand() == np.random.RandomState(1).rand()
    lo, hi = np.asarray(o.bounds).T
    assert np.all((sw.positions >= lo) & (sw.positions <= hi))
    assert np.all(np.abs(sw.velocities) <= 10 * o.optimal_velocities + 1e-15)
    assert np.all(sw.best_values >= v0)                   # personal bests never get worse
    vb, sb = o._compute_particle_fitness("maximizers", sw.best_positions)
    assert_allclose(vb, sw.best_values, rtol=1e-9, atol=1e-12)
    moved = sw.best_values > v0
    assert moved.any() and np.all(sb[moved])              # improvements are safe points
    assert_array_equal(sw.global_best, sw.best_positions[np.argmax(sw.best_values)])
    # same NumPy seed at construction, same call sequence -> same run; another seed -> another
    runs = []
    for seed in (7, 8):
        np.random.seed(seed)
        o2 = _swarm_problem(mods, "device-rng", swarm_size=500)
        o2.best_lower_bound = 0.3
        sw2 = o2.swarms["maximizers"]
        sw2.init_swarm(start.copy())
        sw2.run_swarm(20)
        runs.append(sw2.best_positions.copy())
    assert_array_equal(runs[0], sw.best_positions)
    assert not np.array_equal(runs[1], sw.best_positions)


@pytest.mark.parametrize("kind,d,m,n", [("RBF", 2, 300, 40), ("Matern52", 3, 9000, 64),
                                         ("Matern32", 1, 5, 30), ("RBF", 4, 0, 25),
                                         ("prod", 3, 700, 50)])
def test_swarm_grow_matches_reference_loop(mods, kind, d, m, n):
    """the correlation filter that grows SafeOptSwarm's safe
    set (gp_opt.py:1089-1111), device kernels vs the reference's host loop."""
    _, gpy, gpn, _ = mods
    from safeopt_amd import _hip
    rng = np.random.default_rng(m + n)

    def kern(ns):
        if kind == "prod":
            return (ns.RBF(2, variance=1.5, lengthscale=[0.7, 1.1], ARD=True, active_dims=[0, 1]) *
                    ns.Matern52(1, variance=1.2, lengthscale=0.9, active_dims=[2], name="context"))
        return getattr(ns, kind)(d, variance=2.0, lengthscale=list(0.5 + 0.2 * np.arange(d)), ARD=True)
    X0 = rng.normal(size=(5, d))
    gp = gpy.models.GPRegression(X0, rng.normal(size=(5, 1)), kern(gpy.kern), noise_var=0.01)
    ko = kern(gpn)
    S = rng.uniform(-2, 2, size=(m, d))
    # candidates: some close to S / to each other (rejected), some far (accepted)
    B = rng.uniform(-3, 3, size=(n, d))
    if m:
        B[::5] = S[rng.integers(0, m, size=B[::5].shape[0])] + 0.02 * rng.normal(size=B[::5].shape)
    B[1::7] = B[:1] + 0.03 * rng.normal(size=B[1::7].shape)
    scale2 = float(ko.Kdiag(np.zeros((1, d)))[0])
    ref, cov = _grow_reference(ko.K(B, np.vstack((S, B))), m, scale2)
    off = cov[~np.eye(n, m + n, k=m, dtype=bool)]
    assert np.min(np.abs(off - 0.95)) > 1e-9          # no knife-edge decisions
    dev = gp._fitted()
    got = _hip.swarm_grow(dev.ctx, dev, S, B, scale2, 0.95)
    assert_array_equal(got, ref)
    assert 0 < ref.sum() < n


def test_swarm_empty_safe_set_raises(mods):
    """safeopt/tests/test_swarm.py:13-22"""
    safeopt_amd, gpy, _, _ = mods
    gp = gpy.models.GPRegression(np.array([[0.]]), np.array([[-1.]]), noise_var=0.01 ** 2)
    opt = safeopt_amd.SafeOptSwarm(gp, fmin=[0.], bounds=[[-1., 1.]])
    with pytest.raises(RuntimeError):
        opt.optimize()


def test_swarm_fitness_config5_reduced(mods):
    """configs[4] (4-D RBF, 2 constraints, n=2000) on 3000 particles."""
    safeopt_amd, gpy, gpn, son = mods
    from bench import make_config, build_gps
    cfg = make_config(5, side=3000)
    gps = build_gps(cfg, gpy); gos = build_gps(cfg, gpn)
    opt = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * 4,
                                   threshold=cfg["threshold"])
    opt.best_lower_bound = 0.4
    for st in ["greedy", "maximizers", "expanders", "safe_set"]:
        v, s = opt._compute_particle_fitness(st, cfg["particles"])
        vo, so = son.swarm_fitness(gos, cfg["particles"], st, 2., cfg["fmin"],
                                   opt.scaling, 0.4)
        assert_array_equal(s, so)
        assert_allclose(v, vo, rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("swarm_size", [30, 100])
def test_device_pso_few_points_path_bit_identical(mods, swarm_size):
    """Device PSO == host loop also when the fitness takes the few-points path
    (n = 600 observations; 30 particles: the whole step in one workgroup, 100:
    few-points posterior + the separate PSO kernels)."""
    safeopt_amd, gpy, _, _ = mods
    from bench import make_config, build_gps
    cfg = make_config(5)
    cfg["X"], cfg["Y"], cfg["n"] = cfg["X"][:600], cfg["Y"][:600], 600
    out = []
    for pso in ("host", "device"):
        gps = build_gps(cfg, gpy)
        o = safeopt_amd.SafeOptSwarm(gps, cfg["fmin"], bounds=[(-5., 5.)] * 4,
                                     threshold=cfg["threshold"], swarm_size=swarm_size, pso=pso)
        o.best_lower_bound = 0.4
        np.random.seed(3)
        sw = o.swarms["expanders"]
        sw.init_swarm(np.random.default_rng(1).uniform(-1, 1, size=(swarm_size, 4)))
        sw.run_swarm(15)
        out.append((sw.positions.copy(), sw.velocities.copy(), sw.best_positions.copy(),
                    np.array(sw.best_values), np.array(sw.global_best)))
    for a, b in zip(out[0], out[1]):
        assert_array_equal(a, b)
