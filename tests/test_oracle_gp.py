"""Pin the oracle's GP arithmetic (oracle/gp_numpy.py) -- CPU only.

GPy is unavailable, so the restatement is pinned by (1) scikit-learn vectors
committed in tests/golden/gp_sklearn.npz (+ a live sklearn run when sklearn is
importable), (2) closed-form answers, (3) the one kernel-arithmetic assertion
the reference's own tests make (safeopt/tests/test_gps.py:48-60, Kdiag).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import gp_numpy as gpn
from _golden import load


@pytest.mark.parametrize("tag", ["rbf", "m32", "m52"])
def test_matches_sklearn_vectors(tag):
    z, meta = load("gp_sklearn")
    m = meta[tag]
    k = getattr(gpn, m["kind"])(2, variance=m["variance"],
                                lengthscale=m["lengthscale"], ARD=True)
    gp = gpn.GPRegression(z[tag + "_X"], z[tag + "_Y"], k, noise_var=m["noise_var"])
    mu, var = gp.predict_noiseless(z[tag + "_Xs"])
    assert_allclose(mu.ravel(), z[tag + "_mean"], rtol=1e-9, atol=1e-11)
    # sklearn clips at 0, GPy at 1e-15; compare relative to prior variance
    assert np.max(np.abs(var.ravel() - z[tag + "_var"])) / m["variance"] < 1e-9


def test_live_sklearn_agrees():
    sk = pytest.importorskip("sklearn.gaussian_process")
    from sklearn.gaussian_process.kernels import ConstantKernel, RBF
    rng = np.random.default_rng(0)
    X = rng.uniform(-2, 2, (40, 3))
    Y = np.sin(X.sum(1))[:, None]
    Xs = rng.uniform(-3, 3, (100, 3))
    ls = np.array([0.7, 1.1, 2.0])
    gpr = sk.GaussianProcessRegressor(ConstantKernel(1.5, 'fixed') * RBF(ls, 'fixed'),
                                      alpha=0.01 + 1e-8, optimizer=None).fit(X, Y)
    mu, std = gpr.predict(Xs, return_std=True)
    gp = gpn.GPRegression(X, Y, gpn.RBF(3, 1.5, ls, ARD=True), noise_var=0.01)
    m, v = gp.predict_noiseless(Xs)
    assert_allclose(m.ravel(), mu.ravel(), rtol=1e-9, atol=1e-11)
    assert np.max(np.abs(v.ravel() - std ** 2)) / 1.5 < 1e-9


@pytest.mark.parametrize("cls", [gpn.RBF, gpn.Matern32, gpn.Matern52])
def test_single_point_closed_form(cls):
    # n=1: mu = k(x,x0) y0 / (s2 + nv + 1e-8), var = s2 - k^2/(...)
    s2, nv, y0 = 2.0, 0.05 ** 2, 1.3
    k = cls(1, variance=s2, lengthscale=0.8)
    gp = gpn.GPRegression([[0.]], [[y0]], k, noise_var=nv)
    xs = np.linspace(-3, 3, 31)[:, None]
    kx = k.K(np.zeros((1, 1)), xs).ravel()
    m, v = gp.predict_noiseless(xs)
    den = s2 + nv + 1e-8
    assert_allclose(m.ravel(), kx * y0 / den, rtol=1e-12)
    assert_allclose(v.ravel(), np.clip(s2 - kx ** 2 / den, 1e-15, None), rtol=1e-10)


def test_kernel_values():
    x = np.array([[0., 0.]]); y = np.array([[1., 2.]])
    r = np.sqrt((1 / 0.5) ** 2 + (2 / 2.0) ** 2)
    assert_allclose(gpn.RBF(2, 3., [0.5, 2.], ARD=True).K(x, y), 3 * np.exp(-0.5 * r * r))
    assert_allclose(gpn.Matern32(2, 3., [0.5, 2.], ARD=True).K(x, y),
                    3 * (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r))
    assert_allclose(gpn.Matern52(2, 3., [0.5, 2.], ARD=True).K(x, y),
                    3 * (1 + np.sqrt(5) * r + 5 / 3 * r * r) * np.exp(-np.sqrt(5) * r))
    # non-ARD: one lengthscale for every dimension
    r1 = np.sqrt(5.) / 1.5
    assert_allclose(gpn.RBF(2, 1., 1.5).K(x, y), np.exp(-0.5 * r1 * r1))
    # reference test_gps.py:48-60: scaling='auto' reads sqrt(Kdiag)
    assert_allclose(gpn.RBF(1, variance=2).Kdiag(np.zeros((1, 1))), [2.])
    assert_allclose(gpn.Matern32(1, variance=4).Kdiag(np.zeros((1, 1))), [4.])
    # product on disjoint columns (context_example.ipynb cell 2)
    kp = gpn.RBF(1, 2., 1., active_dims=[0]) * gpn.RBF(1, 2., 1., active_dims=[1], name='context')
    assert_allclose(kp.K(x, y), 4 * np.exp(-0.5 * 5.))
    assert kp.context.variance[0] == 2.


def test_chunked_predict_is_identical():
    rng = np.random.default_rng(3)
    X = rng.normal(size=(30, 2)); Y = rng.normal(size=(30, 1))
    Xs = rng.normal(size=(1000, 2))
    a = gpn.GPRegression(X, Y, gpn.Matern52(2, 1.3, 0.9), noise_var=0.01, chunk=10 ** 6)
    b = gpn.GPRegression(X, Y, gpn.Matern52(2, 1.3, 0.9), noise_var=0.01, chunk=128)
    ma, va = a.predict_noiseless(Xs); mb, vb = b.predict_noiseless(Xs)
    assert_allclose(ma, mb, rtol=0, atol=1e-13); assert_allclose(va, vb, rtol=0, atol=1e-13)


def test_jitchol_retries_and_fails():
    A = np.ones((3, 3))                       # singular PSD -> needs jitter
    L = gpn.jitchol(A)
    assert np.all(np.isfinite(L))
    with pytest.raises(np.linalg.LinAlgError):
        gpn.jitchol(-np.eye(3))
