"""GPy-model-like handle backed by the HIP library.

``import safeopt_amd.gpy as GPy`` gives the slice of GPy's API that SafeOpt
touches (``/root/reference/safeopt/gp_opt.py:83, 121-126, 227, 267, 275, 469,
591, 847, 929, 973, 1093, 1117, 1132``; ``utilities.py:89, 135, 203, 282,
355``)::

    kernel = GPy.kern.RBF(input_dim=2, variance=2., lengthscale=1.0, ARD=True)
    gp = GPy.models.GPRegression(x0, y0, kernel, noise_var=0.05**2)
    gp.set_XY(X, Y); mean, var = gp.predict_noiseless(Xnew); gp.kern.K(X, X2)

Every number is produced on the GPU: ``set_XY`` builds the covariance matrix,
factorises it and inverts the factor on the device; ``predict_noiseless`` and
``kern.K`` run HIP kernels.  The Python objects only carry hyper-parameters
and the host copies of ``X`` / ``Y`` that SafeOpt's plumbing reads back.
"""
from __future__ import annotations

import copy as _copy
import types as _types

import numpy as np

from . import _hip

__all__ = ["kern", "models"]


class _Kern(object):
    name = "kern"

    def __init__(self, input_dim, active_dims=None, name=None):
        self.input_dim = int(input_dim)
        if active_dims is None:
            active_dims = np.arange(self.input_dim)
        self.active_dims = np.atleast_1d(np.asarray(active_dims, dtype=int))
        if name is not None:
            self.name = name

    def __mul__(self, other):
        return Prod([self, other])

    def copy(self):
        return _copy.deepcopy(self)

    # -- device descriptor: (d, kinds, variances, inv_ls[n_parts, d])
    def _parts(self):
        raise NotImplementedError

    def _desc(self, d=None):
        parts = self._parts()
        need = max(int(p.active_dims.max()) + 1 for p in parts)
        d = need if d is None else int(d)
        if d < need:
            raise ValueError("kernel acts on column %d but inputs have %d "
                             "columns" % (need - 1, d))
        if d > _hip.MAX_D or len(parts) > _hip.MAX_PARTS:
            raise ValueError("at most %d input columns and %d kernel factors "
                             "are supported" % (_hip.MAX_D, _hip.MAX_PARTS))
        kinds = np.array([p._kind for p in parts], dtype=np.int32)
        variances = np.array([float(np.asarray(p.variance).ravel()[0])
                              for p in parts])
        inv_ls = np.zeros((len(parts), d))
        for i, p in enumerate(parts):
            ls = np.asarray(p.lengthscale, dtype=float).ravel()
            if ls.size == 1:
                inv_ls[i, p.active_dims] = 1.0 / ls[0]
            else:
                inv_ls[i, p.active_dims] = 1.0 / ls
        return d, kinds, variances, inv_ls

    def _signature(self):
        """The hyper-parameters as they are NOW (cheap: a few byte strings): the GP
        handle compares it with the one it was fitted with, so that an edit in place
        (``kern.lengthscale[0] = 2.``, ``kern.variance = 3.``) takes effect at the next
        use, as GPy's parameter observers make it."""
        return tuple((np.asarray(p.variance, dtype=float).tobytes(),
                      np.asarray(p.lengthscale, dtype=float).tobytes())
                     for p in self._parts())

    def K(self, X, X2=None):
        """Covariance matrix ``k(X, X2)`` (``X2=None``: ``k(X, X)``)."""
        X = np.atleast_2d(np.asarray(X, dtype=float))
        X2 = X if X2 is None else np.atleast_2d(np.asarray(X2, dtype=float))
        desc = self._desc(X.shape[1])
        return _hip.Context.default().kern_K(desc, X, X2)

    def Kdiag(self, X):
        """``k(x, x)`` = product of the variances (stationary kernels)."""
        X = np.atleast_2d(np.asarray(X, dtype=float))
        out = np.empty(X.shape[0])
        out[:] = np.prod([float(np.asarray(p.variance).ravel()[0])
                          for p in self._parts()])
        return out


class _Stationary(_Kern):
    _kind = None

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False,
                 active_dims=None, name=None):
        super(_Stationary, self).__init__(input_dim, active_dims, name)
        if self.active_dims.size != self.input_dim:
            raise ValueError("active_dims must list input_dim columns")
        self.ARD = bool(ARD)
        if lengthscale is None:
            lengthscale = np.ones(self.input_dim if self.ARD else 1)
        lengthscale = np.atleast_1d(np.asarray(lengthscale, dtype=float))
        if self.ARD and lengthscale.size == 1:
            lengthscale = np.ones(self.input_dim) * lengthscale
        if not self.ARD and lengthscale.size != 1:
            raise ValueError("a non-ARD kernel takes one lengthscale")
        if self.ARD and lengthscale.size != self.input_dim:
            raise ValueError("ARD needs one lengthscale per input dimension")
        self.lengthscale = lengthscale
        self.variance = np.atleast_1d(np.asarray(variance, dtype=float))

    def _parts(self):
        return [self]


class RBF(_Stationary):
    """``variance * exp(-r^2 / 2)``"""
    name = "rbf"
    _kind = _hip.RBF


class Matern32(_Stationary):
    """``variance * (1 + sqrt(3) r) exp(-sqrt(3) r)``"""
    name = "Mat32"
    _kind = _hip.MATERN32


class Matern52(_Stationary):
    """``variance * (1 + sqrt(5) r + 5/3 r^2) exp(-sqrt(5) r)``"""
    name = "Mat52"
    _kind = _hip.MATERN52


class Prod(_Kern):
    """``k1 * k2`` -- factors stay reachable by name (``kernel.context``)."""
    name = "mul"

    def __init__(self, parts):
        flat = []
        for p in parts:
            flat.extend(p.parts if isinstance(p, Prod) else [p])
        self.parts = flat
        dims = np.unique(np.concatenate([p.active_dims for p in flat]))
        super(Prod, self).__init__(int(dims.max()) + 1, dims)
        for p in flat:
            setattr(self, p.name, p)

    def _parts(self):
        return self.parts


class GPRegression(object):
    """Exact GP regression with Gaussian noise, zero mean, no normaliser.

    Mirrors ``GPy.models.GPRegression(X, Y, kernel=None, noise_var=1.)`` as
    SafeOpt uses it.  Hyper-parameters are taken from the kernel object each
    time the model is (re)fitted; they are never optimised (the reference never
    calls ``gp.optimize()`` either).
    """

    def __init__(self, X, Y, kernel=None, noise_var=1., device=None):
        X = np.atleast_2d(np.asarray(X, dtype=float))
        Y = np.atleast_2d(np.asarray(Y, dtype=float))
        if Y.shape[1] != 1:
            raise ValueError("one output column per GP (SafeOpt passes a list "
                             "of GPs for several constraints)")
        if kernel is None:
            kernel = RBF(X.shape[1])
        self.kern = kernel
        self.noise_var = float(noise_var)
        self.input_dim = X.shape[1]
        self._ctx = _hip.Context.default(device)
        self._dev = None
        self._dev_key = None
        self._sig = None
        self._dev_fitted = False
        #: one-row changes of the data use bordered updates (set False to
        #: re-factorise from scratch on every ``set_XY`` like GPy)
        self.incremental = True
        self.X = X
        self.Y = Y
        self.set_XY(X, Y)

    @property
    def Gaussian_noise_variance(self):
        return self.noise_var

    def _device_gp(self):
        desc = self.kern._desc(self.input_dim)
        key = (desc[0], desc[1].tobytes(), desc[2].tobytes(),
               desc[3].tobytes(), self.noise_var)
        if self._dev is None or key != self._dev_key:
            self._dev = _hip.DeviceGP(self._ctx, desc, self.noise_var)
            self._dev_key = key
            self._dev_fitted = False
        self._sig = (self.kern._signature(), self.noise_var)
        return self._dev

    def set_XY(self, X, Y):
        """Replace the training data and refit (device Cholesky + inverse)."""
        X = np.array(np.atleast_2d(X), dtype=float)
        Y = np.array(np.atleast_2d(Y), dtype=float)
        if X.shape[0] != Y.shape[0] or X.shape[1] != self.input_dim:
            raise ValueError("inconsistent X %r / Y %r" % (X.shape, Y.shape))
        old_X, old_Y = self.X, self.Y
        self.X, self.Y = X, Y
        dev = self._device_gp()
        n = X.shape[0]
        # what SafeOpt does every iteration is one row more (or one fewer):
        # bordered O(n^2) update instead of the O(n^3) re-factorisation
        if self._dev_fitted and dev.n == old_X.shape[0] and self.incremental:
            if (n == dev.n + 1 and np.array_equal(X[:-1], old_X)
                    and np.array_equal(Y[:-1], old_Y)):
                if dev.append(X[-1], Y[-1, 0]):
                    return
            elif (n == dev.n - 1 and n >= 1 and np.array_equal(X, old_X[:-1])
                    and np.array_equal(Y, old_Y[:-1])):
                dev.pop()
                return
        dev.set_data(X, Y[:, 0])
        self._dev_fitted = True

    def _fitted(self):
        """Device GP, fitted.  Hot path: a comparison of the hyper-parameter bytes
        when nothing changed; an edited kernel parameter or ``noise_var`` refits the
        device model here, at the next use (GPy refits through its observers)."""
        if self._dev is not None and self._dev_fitted:
            if (self.kern._signature(), self.noise_var) == self._sig:
                return self._dev
        dev = self._device_gp()
        if not self._dev_fitted:
            dev.set_data(self.X, self.Y[:, 0])
            self._dev_fitted = True
        return dev

    def parameters_changed(self):
        """Refit the device model with the current hyper-parameters NOW.  Not needed
        for correctness -- an edit of ``kern.variance`` / ``kern.lengthscale`` /
        ``noise_var`` is noticed at the next use -- kept for code that called it."""
        self._dev_fitted = False
        self._fitted()

    def predict_noiseless(self, Xnew, full_cov=False):
        """Posterior mean and variance of the latent function, ``(N,1)`` each;
        the variance is clipped to ``[1e-15, inf)`` as in GPy."""
        if full_cov:
            raise NotImplementedError("full_cov is not on SafeOpt's path")
        return self._fitted().predict(Xnew)

    def _raw_predict(self, Xnew, full_cov=False):
        return self.predict_noiseless(Xnew, full_cov=full_cov)

    def predict(self, Xnew, full_cov=False, include_likelihood=True):
        mean, var = self.predict_noiseless(Xnew, full_cov=full_cov)
        if include_likelihood:
            var = var + self.noise_var
        return mean, var


kern = _types.SimpleNamespace(Kern=_Kern, RBF=RBF, Matern32=Matern32, Matern52=Matern52,
                              Prod=Prod)
models = _types.SimpleNamespace(GPRegression=GPRegression)
