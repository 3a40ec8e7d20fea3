"""NumPy restatement of SafeOpt's confidence-interval / safe-set sweep.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Each function states the reference lines it follows (``/root/reference/
safeopt/gp_opt.py``).  The functions are array-in / array-out (no optimiser
object) so they can be driven from tests, from the golden-vector generator and
from ``bench.py``'s CPU-baseline leg alike.  Pinned by ``tests/golden/*.npz``
(outputs of the reference's own ``gp_opt.py``, see tests/golden/make_golden.py).
"""

from __future__ import annotations

import numpy as np
from scipy.spatial.distance import cdist
from scipy.special import expit
from scipy.stats import norm

__all__ = ["confidence_intervals", "safe_set", "compute_sets",
           "query_index", "maximum_index", "swarm_penalty", "swarm_fitness",
           "optimize_grid"]


def confidence_intervals(gps, inputs, beta, out=None):
    """``SafeOpt.update_confidence_intervals`` -- gp_opt.py:453-476.

    Q[:, 2i] = mean_i - beta*std_i ; Q[:, 2i+1] = mean_i + beta*std_i.
    """
    N = inputs.shape[0]
    Q = np.empty((N, 2 * len(gps))) if out is None else out
    for i, gp in enumerate(gps):
        mean, var = gp.predict_noiseless(inputs)
        mean = mean.squeeze()
        std = np.sqrt(var.squeeze())
        Q[:, 2 * i] = mean - beta * std
        Q[:, 2 * i + 1] = mean + beta * std
    return Q


def safe_set(Q, fmin):
    """``SafeOpt.compute_safe_set`` -- gp_opt.py:478-481 (strict ``>``)."""
    return np.all(Q[:, ::2] > np.asarray(fmin, dtype=float), axis=1)


def _append_point(gp, x, y):
    # gp_opt.py:207-228 (_add_data_point): vstack onto the GP's own data
    gp.set_XY(np.vstack([gp.X, x]), np.vstack([gp.Y, y]))


def _pop_point(gp):
    # gp_opt.py:257-267 (_remove_last_data_point)
    gp.set_XY(gp.X[:-1, :], gp.Y[:-1, :])


def compute_sets(gps, inputs, Q, fmin, scaling, threshold, beta,
                 lipschitz=None, full_sets=False, return_trace=False):
    """``SafeOpt.compute_sets`` -- gp_opt.py:483-615.

    Returns boolean arrays ``S, M, G`` (and, with ``return_trace``, the list
    of candidate indices visited by the expander loop, in visiting order).

    ``inputs`` must already carry the context columns (the reference adds
    ``self.context`` to ``parameter_set[s][index]`` at :585-588, which equals
    the corresponding ``inputs`` row).
    """
    fmin = np.atleast_1d(np.asarray(fmin, dtype=float))
    scaling = np.asarray(scaling, dtype=float)
    N = Q.shape[0]
    S = safe_set(Q, fmin)
    M = np.zeros(N, dtype=bool)
    G = np.zeros(N, dtype=bool)
    trace = []

    def done():
        return (S, M, G, trace) if return_trace else (S, M, G)

    if not S.any():                                   # :504-507
        return done()

    l0, u0 = Q[:, 0], Q[:, 1]
    M[S] = u0[S] >= np.max(l0[S])                     # :511-512
    max_var = np.max(u0[M] - l0[M]) / scaling[0]      # :513

    lo = Q[:, ::2]
    up = Q[:, 1::2]

    if full_sets:                                     # :527-528
        s = S.copy()
    else:
        s = np.logical_and(S, ~M)                     # :531
        s[s] = np.max((up[s, :] - lo[s, :]) / scaling, axis=1) > max_var
        s[s] = np.any(up[s, :] - lo[s, :] > threshold * beta, axis=1)
        if not s.any():                               # :538-540
            return done()

    cand = np.flatnonzero(s)
    G_safe = np.zeros(cand.size, dtype=bool)
    if full_sets:                                     # :553-555
        order = range(cand.size)
    else:                                             # :542-552
        order = np.max(up[s, :] - lo[s, :], axis=1).argsort()[::-1]

    unsafe = ~S
    for k in order:                                   # :557
        idx = cand[k]
        trace.append(int(idx))
        if lipschitz is not None:                     # :558-576
            d = cdist(inputs[[idx], :], inputs[unsafe, :])
            for i in range(len(gps)):
                if fmin[i] == -np.inf:
                    continue
                G_safe[k] = np.any(up[idx, i] - lipschitz[i] * d >= fmin[i])
                if not G_safe[k]:
                    break
        else:                                         # :577-606
            for i, gp in enumerate(gps):
                if fmin[i] == -np.inf:
                    continue
                _append_point(gp, inputs[[idx], :], np.atleast_2d(up[idx, i]))
                mean2, var2 = gp.predict_noiseless(inputs[unsafe])
                _pop_point(gp)
                l2 = mean2.squeeze() - beta * np.sqrt(var2.squeeze())
                G_safe[k] = np.any(l2 >= fmin[i])
                if not G_safe[k]:
                    break
        if G_safe[k] and not full_sets:               # :611-612
            break

    G[cand] = G_safe                                  # :615
    return done()


def expander_hits_rank1(gp, inputs, unsafe, cand, u_cand, beta, fmin_i, chunk=256):
    """``np.any(l2 >= fmin[i])`` of gp_opt.py:585-606 for MANY candidates of ONE GP without
    refitting: appending ``(x_c, u)`` to a GP with ``Ky^-1 = woodbury_inv`` changes the
    posterior at ``x`` by the rank-1 (Schur complement) update

        c(x) = k(x, x_c) - k(X, x)^T Ky^-1 k(X, x_c),     s2 = var(x_c) + noise + 1e-8,
        mean2 = mean + c (u - mean(x_c)) / s2,             var2 = max(var - c^2 / s2, 1e-15)

    -- algebraically what ``_add_data_point`` -> ``set_XY`` -> ``predict_noiseless`` ->
    ``_remove_last_data_point`` computes from scratch (``var`` taken BEFORE GPy's 1e-15 clip
    would matter only at rows the GP already knows exactly).  Test infrastructure for the
    big-pass expander loop, where the refit form (O(n^3 + N n^2) per candidate) cannot be run
    for thousands of candidates; pinned against the refit form on small problems in
    tests/test_oracle_safeopt.py.  Returns a boolean per candidate."""
    inputs = np.asarray(inputs, dtype=float)
    U = inputs[np.asarray(unsafe, dtype=bool)]
    cand = np.asarray(cand, dtype=np.int64)
    Xc = inputs[cand]
    Wi, alpha = gp.woodbury_inv, gp.woodbury_vector[:, 0]
    KUX = gp.kern.K(U, gp.X)                             # (Nu, n)
    mean = KUX.dot(alpha)
    var = gp.kern.Kdiag(U) - np.einsum('ij,jk,ik->i', KUX, Wi, KUX)
    out = np.zeros(cand.size, dtype=bool)
    for a in range(0, cand.size, chunk):
        xb = Xc[a:a + chunk]
        KcX = gp.kern.K(gp.X, xb)                        # (n, m)
        W = Wi.dot(KcX)
        var_c = gp.kern.Kdiag(xb) - np.sum(KcX * W, axis=0)
        mu_c = KcX.T.dot(alpha)
        s2 = var_c + gp.noise_var + 1e-8
        C = gp.kern.K(U, xb) - KUX.dot(W)                # (Nu, m)
        mean2 = mean[:, None] + C * ((np.asarray(u_cand)[a:a + chunk] - mu_c) / s2)[None, :]
        var2 = np.clip(var[:, None] - C * C / s2[None, :], 1e-15, np.inf)
        out[a:a + chunk] = np.any(mean2 - beta * np.sqrt(var2) >= fmin_i, axis=0)
    return out


def query_index(Q, S, M, G, scaling, ucb=False):
    """``SafeOpt.get_new_query_point`` -- gp_opt.py:617-649 (global index).

    ``np.argmax`` = first index among equal values.  Raises ``EnvironmentError``
    when the safe set is empty (:631-632).
    """
    if not np.any(S):
        raise EnvironmentError('There are no safe points to evaluate.')
    if ucb:
        return int(np.flatnonzero(S)[np.argmax(Q[S, 1])])
    lo = Q[:, ::2]
    up = Q[:, 1::2]
    MG = np.logical_or(M, G)
    value = np.max((up[MG] - lo[MG]) / np.asarray(scaling), axis=1)
    return int(np.flatnonzero(MG)[np.argmax(value)])


def maximum_index(Q, S):
    """``SafeOpt.get_maximum`` -- gp_opt.py:677-712; ``None`` if S is empty."""
    if not np.any(S):
        return None
    return int(np.flatnonzero(S)[np.argmax(Q[S, 0])])


def optimize_grid(gps, inputs, fmin, scaling, threshold, beta,
                  lipschitz=None, ucb=False):
    """One ``SafeOpt.optimize()`` -- gp_opt.py:651-675 -- on arrays.

    Returns ``(index, Q, S, M, G)``.
    """
    Q = confidence_intervals(gps, inputs, beta)
    if ucb:
        S = safe_set(Q, fmin)
        M = np.zeros_like(S)
        G = np.zeros_like(S)
    else:
        S, M, G = compute_sets(gps, inputs, Q, fmin, scaling, threshold, beta,
                               lipschitz=lipschitz)
    return query_index(Q, S, M, G, scaling, ucb=ucb), Q, S, M, G


# --------------------------------------------------------------------------
# SafeOptSwarm particle fitness
# --------------------------------------------------------------------------
def swarm_penalty(slack):
    """``SafeOptSwarm._compute_penalty`` -- gp_opt.py:874-899."""
    slack = np.atleast_1d(np.asarray(slack, dtype=float))
    pen = np.clip(slack, None, 0)
    pen[(slack < 0) & (slack > -0.001)] *= 2
    pen[(slack <= -0.001) & (slack > -0.1)] *= 5
    pen[(slack <= -0.1) & (slack > -1)] *= 10
    far = slack < -1
    pen[far] = -300 * pen[far] ** 2
    return pen


def swarm_fitness(gps, particles, swarm_type, beta, fmin, scaling,
                  best_lower_bound=-np.inf):
    """``SafeOptSwarm._compute_particle_fitness`` -- gp_opt.py:901-1013.

    Returns ``(values, global_safe)``; for ``'greedy'`` the safety mask is all
    True (:938-939) and for ``'safe_set'`` the first output is the lower bound
    of the *last* GP evaluated (:1001-1004).
    """
    fmin = np.atleast_1d(np.asarray(fmin, dtype=float))
    scaling = np.asarray(scaling, dtype=float)
    particles = np.atleast_2d(particles)
    P = particles.shape[0]

    mean, var = gps[0].predict_noiseless(particles)
    mean = mean.squeeze()
    std = np.sqrt(var.squeeze())
    lower = np.atleast_1d(mean - beta * std)
    upper = np.atleast_1d(mean + beta * std)

    if swarm_type == 'greedy':
        return lower, np.ones(P, dtype=bool)

    values = np.atleast_1d(std / scaling[0]).astype(float)
    is_safe = swarm_type == 'safe_set'
    if is_safe:
        interest = None
    elif swarm_type == 'expanders':
        interest = len(gps) * np.ones(P)
    elif swarm_type == 'maximizers':
        interest = expit(10 * (upper - best_lower_bound) / scaling[0])
    else:
        raise AssertionError("Invalid swarm type")

    global_safe = np.ones(P, dtype=bool)
    total_penalty = np.zeros(P)
    for i, gp in enumerate(gps):
        if i > 0:
            mean, var = gp.predict_noiseless(particles)
            std = np.sqrt(var.squeeze())
            lower = np.atleast_1d(mean.squeeze() - beta * std)
            values = np.maximum(values, std / scaling[i])
        if fmin[i] == -np.inf:
            continue
        slack = np.atleast_1d(lower - fmin[i])
        global_safe &= slack >= 0
        if is_safe:
            continue
        slack = slack / scaling[i]
        total_penalty += swarm_penalty(slack)
        if swarm_type == 'expanders':
            interest = interest * norm.pdf(slack, scale=0.2)

    if is_safe:
        return lower, global_safe
    return (values + total_penalty) * interest, global_safe
