"""The expander loop where it goes far (gp_opt.py:557-612): no expander among the first
candidates, none at all (a converged run: EVERY candidate is visited), and ``full_sets``
(:553-555: every safe row is).  ``sgp_grid_expander_pass`` tests hundreds to thousands of
candidates per device pass; the oracle side is ``son.expander_hits_rank1`` -- the closed form of
append / predict / pop, pinned against the refit form in tests/test_oracle_safeopt.py."""
import numpy as np
import pytest
from numpy.testing import assert_array_equal, assert_allclose

import _scenarios as sc

pytestmark = pytest.mark.gpu

STATE = dict(r0=2.0, rings=8, ls=0.4, dmid=0.45, plateau=0.6)


@pytest.fixture(scope="module")
def mods():
    import safeopt_amd
    import safeopt_amd.gpy as gpy
    from oracle import gp_numpy as gpn
    from oracle import safeopt_numpy as son
    return safeopt_amd, gpy, gpn, son


def oracle_front(son, go, grid, beta, thr):
    """S, M, the candidate rows in the reference's visiting order, the arg-max value per row."""
    Q = son.confidence_intervals([go], grid, beta)
    S = son.safe_set(Q, [0.0])
    M = np.zeros(len(grid), dtype=bool)
    M[S] = Q[S, 1] >= Q[S, 0].max()
    w = Q[:, 1] - Q[:, 0]
    s = S & ~M
    s[s] = w[s] > w[M].max()
    s[s] = w[s] > thr * beta
    cand = np.flatnonzero(s)
    return Q, S, M, cand[w[cand].argsort()[::-1]], w


@pytest.mark.timeout(1200)
def test_converged_run_visits_every_candidate_and_marks_none(mods):
    """>= 1e5 rows, >= 5000 candidates, NO expander: ``S / M / G`` and the chosen row equal the
    oracle's; the loop reaches its end in big passes (not in 16-candidate round trips), and the
    16-candidate loop gives the same answer."""
    safeopt_amd, gpy, gpn, son = mods
    data = sc.rim_data(320, **STATE)
    go = sc.make_gp(gpn, data)
    grid = np.ascontiguousarray(data["grid"][sc.converged_rows(go, data["grid"], 0.05)])
    beta, thr = 2.0, 0.1
    Q, S, M, order, w = oracle_front(son, go, grid, beta, thr)
    assert len(grid) >= 100000 and order.size >= 5000
    hits = son.expander_hits_rank1(go, grid, ~S, order, Q[order, 1], beta, 0.0)
    assert not hits.any()                    # the oracle: no candidate is an expander
    Gm = np.zeros(len(grid), dtype=bool)
    idx = son.query_index(Q, S, M, Gm, np.array([1.0]))

    opt = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, threshold=thr)
    calls = []
    orig = opt._backend.expander_pass
    opt._backend.expander_pass = lambda *a: calls.append(a[7]) or orig(*a)       # (a[7]: want)
    x = opt.optimize()
    assert_allclose(opt.Q, Q, rtol=0, atol=1e-8)
    assert_array_equal(opt.S, S)
    assert_array_equal(opt.M, M)
    assert_array_equal(opt.G, Gm)
    assert_array_equal(x, grid[idx])
    assert calls == [1024, 8192]             # (n = 637: SafeOpt._pass_size) the first candidate + 1024 + the rest
    # ... and in three passes
    opt3 = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, threshold=thr)
    opt3.pass_sizes = (256, 2048, 8192)
    calls3 = []
    orig3 = opt3._backend.expander_pass
    opt3._backend.expander_pass = lambda *a: calls3.append(a[7]) or orig3(*a)
    x3 = opt3.optimize()
    assert calls3 == [256, 2048, 8192]
    assert_array_equal(opt3.G, Gm)
    assert_array_equal(x3, grid[idx])
    # the 16-candidates-per-round-trip loop (round 5): same sets, same point
    ref = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, threshold=thr)
    ref.big_passes = False
    assert_array_equal(ref.optimize(), x)
    assert_array_equal(ref.G, Gm)


@pytest.mark.timeout(1200)
def test_first_expander_far_down_the_visiting_order(mods):
    """A state WITH rows just below fmin: some candidate lifts one of them -- but not one of the
    first sixteen.  The marked row is the first expander of the reference's visiting
    order (oracle: the candidates in ``argsort()[::-1]`` order up to the first hit)."""
    safeopt_amd, gpy, gpn, son = mods
    # (a lower plateau: wide rows of the plateau itself are unsafe, the 111th candidate of the
    # visiting order is the first one that lifts a row across fmin)
    data = sc.rim_data(320, **dict(STATE, plateau=0.45))
    go, grid = sc.make_gp(gpn, data), data["grid"]
    beta, thr = 2.0, 0.1
    Q, S, M, order, w = oracle_front(son, go, grid, beta, thr)
    first = None
    for a in range(0, order.size, 512):
        part = order[a:a + 512]
        h = son.expander_hits_rank1(go, grid, ~S, part, Q[part, 1], beta, 0.0)
        if h.any():
            first = int(part[np.argmax(h)])
            n_before = a + int(np.argmax(h))
            break
    assert first is not None and n_before >= 16
    # (no exact tie at that width: the visiting order is unambiguous)
    assert (w[order] == w[first]).sum() == 1
    for big in (True, False):
        opt = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, threshold=thr)
        opt.big_passes = big
        opt.optimize()
        assert_array_equal(np.flatnonzero(opt.G), [first])
        assert_array_equal(opt.S, S)
        assert_array_equal(opt.M, M)


@pytest.mark.timeout(1200)
def test_full_sets_in_big_passes(mods):
    """``compute_sets(full_sets=True)`` (gp_opt.py:527-528, 553-555): every safe row is a
    candidate and every expander among them is marked -- on the device, pass by pass; equal to
    the oracle's closed form and to the 16-candidate loop."""
    safeopt_amd, gpy, gpn, son = mods
    data = sc.rim_data(120, **STATE)
    go, grid = sc.make_gp(gpn, data), data["grid"]
    beta, thr = 2.0, 0.1
    Q = son.confidence_intervals([go], grid, beta)
    S = son.safe_set(Q, [0.0])
    rows = np.flatnonzero(S)
    hits = son.expander_hits_rank1(go, grid, ~S, rows, Q[rows, 1], beta, 0.0)
    # rows whose best lift lands within rounding of fmin may go either way: find them
    hits_lo = son.expander_hits_rank1(go, grid, ~S, rows, Q[rows, 1], beta, 1e-9)
    hits_hi = son.expander_hits_rank1(go, grid, ~S, rows, Q[rows, 1], beta, -1e-9)
    clear = hits_lo == hits_hi
    assert clear.sum() >= rows.size - 3 and hits.any() and not hits.all()
    got = []
    for big in (True, False):
        opt = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, threshold=thr)
        opt.big_passes = big
        opt.pass_sizes = (300, 1000)          # (several passes even on this small grid)
        opt.update_confidence_intervals()
        opt.compute_sets(full_sets=True)
        got.append(np.array(opt.G))
        assert_array_equal(got[-1][rows][clear], hits[clear])
        assert not got[-1][~S].any()
    assert_array_equal(got[0], got[1])


def lipschitz_hits(grid, S, cand, u_c, L, fmin, chunk=256):
    """gp_opt.py:558-576 for many candidates: any(u_c - L d >= fmin) over the unsafe rows."""
    from scipy.spatial.distance import cdist
    unsafe = grid[~S]
    out = np.zeros(cand.size, dtype=bool)
    for a in range(0, cand.size, chunk):
        d = cdist(grid[cand[a:a + chunk]], unsafe)
        out[a:a + chunk] = np.any(u_c[a:a + chunk, None] - L * d >= fmin, axis=1)
    return out


@pytest.mark.timeout(1200)
def test_lipschitz_certificates_in_big_passes(mods):
    """The same loop with Lipschitz certificates (gp_opt.py:558-576; ``sgp_grid_lipschitz_pass``):
    a constant so large that no candidate reaches an unsafe row (every candidate is visited,
    nothing is marked), one with which the first expander is far down the visiting order, and
    ``full_sets`` -- against the formula of the reference in numpy and against the 16-candidate
    loop."""
    safeopt_amd, gpy, gpn, son = mods
    data = sc.rim_data(320, **STATE)
    go, grid = sc.make_gp(gpn, data), data["grid"]
    beta, thr = 2.0, 0.1
    Q, S, M, order, w = oracle_front(son, go, grid, beta, thr)
    assert order.size >= 5000
    # (1) no expander at all: the candidates are rows of the plateau, u / (distance to the nearest
    # unsafe row) is 0.6 .. 0.75 for them
    from scipy.spatial.distance import cdist
    unsafe = grid[~S]
    dmin = np.concatenate([cdist(grid[order[a:a + 256]], unsafe).min(axis=1)
                           for a in range(0, order.size, 256)])
    ratio = Q[order, 1] / dmin
    L_far = 1.0
    assert ratio.max() < 0.9
    assert not lipschitz_hits(grid, S, order, Q[order, 1], L_far, 0.0).any()
    # (2) the first expander far down the order: a constant just above what the first 64
    # candidates would need
    L_mid = float(ratio[:64].max() * 1.02)
    h = lipschitz_hits(grid, S, order, Q[order, 1], L_mid, 0.0)
    assert h.any()
    first, n_before = int(order[np.argmax(h)]), int(np.argmax(h))
    assert n_before >= 64
    assert (w[order] == w[first]).sum() == 1
    for L, want_G in ((L_far, []), (L_mid, [first])):
        got = []
        for big in (True, False):
            opt = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, lipschitz=L, threshold=thr)
            opt.big_passes = big
            calls = []
            if big:
                orig = opt._backend.lipschitz_pass
                opt._backend.lipschitz_pass = lambda *a, _o=orig: calls.append(a[7]) or _o(*a)
            x = opt.optimize()
            assert_array_equal(opt.S, S)
            assert_array_equal(opt.M, M)
            assert_array_equal(np.flatnonzero(opt.G), want_G)
            assert big == bool(calls)
            got.append(x)
        assert_array_equal(got[0], got[1])
    # (3) full_sets on a smaller grid: every safe row is tested, every expander marked
    data = sc.rim_data(120, **STATE)
    go, grid = sc.make_gp(gpn, data), data["grid"]
    Q = son.confidence_intervals([go], grid, beta)
    S = son.safe_set(Q, [0.0])
    rows = np.flatnonzero(S)
    L = 1.0
    hits = lipschitz_hits(grid, S, rows, Q[rows, 1], L, 0.0)
    clear = hits == lipschitz_hits(grid, S, rows, Q[rows, 1], L, 1e-9)
    clear &= hits == lipschitz_hits(grid, S, rows, Q[rows, 1], L, -1e-9)
    assert hits.any() and not hits.all() and clear.sum() >= rows.size - 3
    got = []
    for big in (True, False):
        opt = safeopt_amd.SafeOpt(sc.make_gp(gpy, data), grid, 0.0, lipschitz=L, threshold=thr)
        opt.big_passes = big
        opt.pass_sizes = (300, 1000)          # (several passes even on this small grid)
        opt.update_confidence_intervals()
        opt.compute_sets(full_sets=True)
        Gd = np.array(opt.G)
        assert_array_equal(Gd[rows][clear], hits[clear])
        assert not Gd[~S].any()
        got.append(Gd)
    assert_array_equal(got[0], got[1])
