"""Input generators of the SafeOpt examples (host side).

``linearly_spaced_combinations`` defines the candidate grid every benchmark
config uses (``/root/reference/safeopt/utilities.py:21-54``): the first
variable runs fastest for two inputs (NumPy ``meshgrid`` 'xy' order) and the
result is an F-ordered ``(N, d)`` array -- which is exactly the SoA layout the
device keeps, so uploading it is one contiguous copy.

``sample_gp_function`` (``utilities.py:57-143``) draws a synthetic objective
from a GP prior; with the package's kernels the interpolant between the
sampled nodes is the posterior mean of a device GP handle.  Plotting helpers of
the reference are out of scope.
"""
from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import scipy.linalg

__all__ = ['linearly_spaced_combinations', 'sample_gp_function']


def linearly_spaced_combinations(bounds, num_samples):
    """All combinations of linearly spaced values inside ``bounds``.

    Parameters
    ----------
    bounds : sequence of (min, max)
    num_samples : int or sequence of int
        Samples per dimension.

    Returns
    -------
    ndarray, shape (prod(num_samples), len(bounds))
    """
    num_vars = len(bounds)
    if not isinstance(num_samples, Sequence):
        num_samples = [num_samples] * num_vars
    if num_vars == 1:
        return np.linspace(bounds[0][0], bounds[0][1], num_samples[0])[:, None]
    axes = [np.linspace(b[0], b[1], n) for b, n in zip(bounds, num_samples)]
    return np.array([g.ravel() for g in np.meshgrid(*axes)]).T


class _GridSample(object):
    """One draw of a GP prior on a grid, turned into a callable ``f(x, noise)``.

    ``values`` are the sampled function values at ``nodes``.  Two ways to
    continue them between the nodes (``utilities.py:100-143`` of the reference):

    * ``'kernel'``: the RKHS interpolant ``k(x, nodes) (K + 1e-6 I)^-1 values``.
      That is the posterior mean of a GP with the sample as data, so for the
      package's own kernels it is one ``predict_noiseless`` of a device handle
      (factorisation, covariance rows and the contraction all on the GPU);
      any other object with a ``.K`` method gets the same formula on the host.
    * ``'linear'``: piecewise-linear interpolation (SciPy ``griddata``).
    """

    JITTER = 1e-6                  # added to the prior covariance of the nodes

    def __init__(self, kernel, nodes, values, cov, noise_var, mode, mean_function):
        if mode not in ('kernel', 'linear'):
            raise ValueError("interpolation must be 'kernel' or 'linear'")
        self.nodes, self.values = nodes, values
        self.noise_std = np.sqrt(noise_var)
        self.mean_function = mean_function
        self._between = self._linear if mode == 'linear' else None
        if mode == 'kernel':
            from . import gpy
            if isinstance(kernel, gpy.kern.Kern):
                # GPRegression factorises K + (noise_var + 1e-8) I
                handle = gpy.models.GPRegression(
                    nodes, values[:, None], kernel, noise_var=self.JITTER - 1e-8)
                self._between = lambda x: handle.predict_noiseless(x)[0][:, 0]
            else:
                weights = scipy.linalg.cho_solve(scipy.linalg.cho_factor(cov), values)
                self._between = lambda x: kernel.K(x, nodes).dot(weights)

    def _linear(self, x):
        from scipy.interpolate import griddata
        return np.atleast_1d(griddata(self.nodes, self.values, x,
                                      method='linear').squeeze())

    def __call__(self, x, noise=True):
        x = np.atleast_2d(x)
        y = np.asarray(self._between(x), dtype=float).reshape(-1, 1)
        if self.mean_function is not None:
            y = y + self.mean_function(x)
        if noise:                                  # one randn call per evaluation
            y = y + self.noise_std * np.random.randn(x.shape[0], 1)
        return y


def sample_gp_function(kernel, bounds, noise_var, num_samples,
                       interpolation='kernel', mean_function=None):
    """Draw a function from the GP prior with covariance ``kernel`` on
    ``bounds`` (same arguments as ``utilities.py:57-143`` of the reference).

    The prior is sampled on the ``linearly_spaced_combinations(bounds,
    num_samples)`` grid with ONE ``np.random.multivariate_normal`` call (the
    global NumPy stream, like the reference).  Returns ``f(x, noise=True)``
    mapping ``(n, d)`` inputs to ``(n, 1)`` values; ``noise=True`` adds
    ``sqrt(noise_var) * randn``.
    """
    nodes = linearly_spaced_combinations(bounds, num_samples)
    cov = kernel.K(nodes) + _GridSample.JITTER * np.eye(nodes.shape[0])
    values = np.random.multivariate_normal(np.zeros(nodes.shape[0]), cov)
    return _GridSample(kernel, nodes, values, cov, noise_var, interpolation,
                       mean_function)
