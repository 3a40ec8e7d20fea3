"""Input generators of the SafeOpt examples (host side).

``linearly_spaced_combinations`` defines the candidate grid every benchmark
config uses (``/root/reference/safeopt/utilities.py:21-54``): the first
variable runs fastest for two inputs (NumPy ``meshgrid`` 'xy' order) and the
result is an F-ordered ``(N, d)`` array -- which is exactly the SoA layout the
device keeps, so uploading it is one contiguous copy.

``sample_gp_function`` (``utilities.py:57-143``) draws a synthetic objective
from a GP prior; its kernel evaluations go through ``kernel.K`` (HIP).  It is a
test-input generator, not part of the accelerated path.  Plotting helpers of
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


def sample_gp_function(kernel, bounds, noise_var, num_samples,
                       interpolation='kernel', mean_function=None):
    """Draw a function from a GP prior (see the reference docstring).

    Returns ``f(x, noise=True)`` mapping ``(n, d)`` inputs to ``(n, 1)``
    (noisy) function values.  Uses the global NumPy RNG like the reference.
    """
    inputs = linearly_spaced_combinations(bounds, num_samples)
    cov = kernel.K(inputs) + np.eye(inputs.shape[0]) * 1e-6
    output = np.random.multivariate_normal(np.zeros(inputs.shape[0]), cov)

    if interpolation == 'linear':
        from scipy.interpolate import griddata

        def evaluate_gp_function_linear(x, noise=True):
            x = np.atleast_2d(x)
            y = griddata(inputs, output, x, method='linear')
            y = np.atleast_2d(y.squeeze()).T
            if mean_function is not None:
                y += mean_function(x)
            if noise:
                y += np.sqrt(noise_var) * np.random.randn(x.shape[0], 1)
            return y
        return evaluate_gp_function_linear

    if interpolation == 'kernel':
        factor = scipy.linalg.cho_factor(cov)
        alpha = scipy.linalg.cho_solve(factor, output)

        def evaluate_gp_function_kernel(x, noise=True):
            x = np.atleast_2d(x)
            y = kernel.K(x, inputs).dot(alpha)[:, None]
            if mean_function is not None:
                y += mean_function(x)
            if noise:
                y += np.sqrt(noise_var) * np.random.randn(x.shape[0], 1)
            return y
        return evaluate_gp_function_kernel

    raise ValueError("interpolation must be 'kernel' or 'linear'")
