"""Shared by the GPU parity modules (tests/test_gpu_*.py): the ``mods`` fixture, kernels and problems.

HIP path vs the CPU oracle and the golden vectors (needs an MI355X).  Everything in those
modules goes through the C ABI of libsafeopt_hip.so (ctypes wrappers in safeopt_amd/_hip.py);
the oracle is only the checker.  Tolerances: the north-star asks posterior mean / variance
within 1e-5 relative in fp64; the kernels are held to 1e-9 (mean, relative to max|mean|;
variance, absolute relative to the prior variance k(x,x)) and masks / chosen points to equality.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from _golden import load, make_kernel

MEAN_TOL = 1e-9
VAR_TOL = 1e-9


@pytest.fixture(scope="module")
def mods(hip_device):
    import safeopt_amd
    import safeopt_amd.gpy as gpy
    from oracle import gp_numpy as gpn
    from oracle import safeopt_numpy as son
    return safeopt_amd, gpy, gpn, son


def smooth(x, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-3, 3, size=(10, x.shape[1]))
    w = rng.normal(size=10)
    r2 = ((x[:, None, :] - c[None]) ** 2).sum(-1)
    return (np.exp(-0.25 * r2) * w).sum(1)[:, None]


def kernels(ns, kind, d, rng=None):
    ls = np.linspace(0.8, 1.6, d)
    return getattr(ns, kind)(d, variance=1.7, lengthscale=ls, ARD=True)


def check_posterior(m, v, m_ref, v_ref, kdiag):
    scale = max(np.max(np.abs(m_ref)), 1e-300)
    assert np.max(np.abs(m - m_ref)) / scale < MEAN_TOL
    assert np.max(np.abs(v - v_ref)) / kdiag < VAR_TOL
    big = v_ref > 1e-6 * kdiag
    assert np.max(np.abs(v[big] - v_ref[big]) / v_ref[big]) < 1e-5


# ---------------------------------------------------------------------------


def product_kernel(ns, d, spec, seed):
    """Prod kernel from ``spec`` = [(kind, columns), ...] (columns may overlap)."""
    rng = np.random.default_rng(seed)
    k = None
    for kind, cols in spec:
        part = getattr(ns, kind)(len(cols), variance=float(rng.uniform(0.6, 1.8)),
                                 lengthscale=rng.uniform(0.7, 1.9, size=len(cols)),
                                 ARD=True, active_dims=list(cols))
        k = part if k is None else k * part
    return k


# Products of parts (GPy's Prod kernel; the reference's context example multiplies a
# kernel over the parameters by one over the context): KernFast::product_n adds the
# parts' exponents -- disjoint and OVERLAPPING column sets, 2 to 4 parts, every kind,
# a single-part GP in the same launch, both sweep kernels, n across 256 / 512.


GOLD = ["safeopt_1d_rbf", "safeopt_2d_rbf", "safeopt_1d_multi",
        "safeopt_2d_mat52_g3", "safeopt_1d_lipschitz", "safeopt_context",
        "safeopt_2d_ucb"]


def build_opt(mods, z, meta, t, **kw):
    safeopt_amd, gpy, _, _ = mods
    gps = [gpy.models.GPRegression(z["it%d_X%d" % (t, i)], z["it%d_Y%d" % (t, i)],
                                   make_kernel(gpy.kern, spec),
                                   noise_var=meta["noise_vars"][i])
           for i, spec in enumerate(meta["kernels"])]
    lip = meta["lipschitz"]
    if lip is not None and len(lip) == 1:
        lip = lip[0]
    return safeopt_amd.SafeOpt(gps if len(gps) > 1 else gps[0], z["parameter_set"],
                               meta["fmin"] if len(gps) > 1 else meta["fmin"][0],
                               lipschitz=lip, beta=float(z["beta_all"][t]),
                               threshold=meta["threshold"],
                               num_contexts=meta["num_contexts"], **kw)


def _swarm_problem(mods, pso, swarm_size=40):
    safeopt_amd, gpy, _, _ = mods
    z, meta = load("swarm_2d_g2")
    gps = [gpy.models.GPRegression(z["X0"], z["Y0"][:, [i]], make_kernel(gpy.kern, meta["kernels"][i]),
                                   noise_var=meta["noise_vars"][i]) for i in range(2)]
    return safeopt_amd.SafeOptSwarm(gps, meta["fmin"], bounds=[tuple(b) for b in meta["bounds"]],
                                    threshold=meta["threshold"], swarm_size=swarm_size, pso=pso)


def _grow_reference(K, m, scale2, thr=0.95):
    """The host loop of gp_opt.py:1089-1111 on a covariance matrix K (n, m + n)."""
    cov = K / scale2
    n = cov.shape[0]
    mask = np.zeros(m + n, dtype=bool)
    mask[:m] = True
    acc = np.zeros(n, dtype=bool)
    for j in range(n):
        if np.all(cov[j, mask] <= thr):
            acc[j] = True
            mask[m + j] = True
    return acc, cov


def kernels_from(cfg, ns):
    return [make_kernel(ns, spec) for spec in cfg["kernels"]]


class _PretendWorld(object):
    """A one-rank RCCL communicator that claims ``world`` ranks towards the host
    driver: SafeOpt takes every N-rank branch (sharding, packed all-gathers,
    in-stream all-reduces) while the collectives themselves run for real."""
    rank, in_stream = 0, True

    def __init__(self, comm, world):
        self._c, self.world = comm, world

    def allreduce_max(self, a):
        return self._c.allreduce_max(a)

    def allgather(self, a):
        return self._c.allgather(a)

    def barrier(self):
        self._c.barrier()


class _PretendWorldPadded(_PretendWorld):
    """... and whose all-gathers return ``world`` blocks: the ranks that do not
    exist contribute zeros (no rows, no candidates), which is what a rank with an
    empty share of the candidates sends."""

    def allgather(self, a):
        got = self._c.allgather(a)
        pad = np.zeros((self.world - got.shape[0],) + got.shape[1:], dtype=got.dtype)
        return np.concatenate([got, pad])


def _dev_script(name):
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "scripts", "dev", name + ".py")
    spec = importlib.util.spec_from_file_location("dev_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
