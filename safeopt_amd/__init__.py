"""safeopt_amd -- SafeOpt's GP-posterior + safe-set hot path on MI355X.

Drop-in for the hot path of befelix/SafeOpt::

    import safeopt_amd as safeopt
    import safeopt_amd.gpy as GPy

    gp = GPy.models.GPRegression(x0, y0, GPy.kern.RBF(2, variance=2., ARD=True),
                                 noise_var=0.05 ** 2)
    opt = safeopt.SafeOpt(gp, safeopt.linearly_spaced_combinations(bounds, 1000),
                          fmin=0., threshold=0.2)
    x = opt.optimize(); opt.add_new_data_point(x, measure(x))

The arithmetic (covariance build, Cholesky, posterior sweep, set passes) runs
in hand-written HIP kernels for gfx950 behind the C ABI of
``include/safeopt_hip.h``; there is no CPU fallback.  Build the library with
``python -m safeopt_amd.build``.
"""
from .utilities import linearly_spaced_combinations, sample_gp_function
from .swarm import SwarmOptimization, DeviceSwarmOptimization
from .gp_opt import SafeOpt, SafeOptSwarm, GaussianProcessOptimization
from . import gpy
from . import dist

__all__ = ['SafeOpt', 'SafeOptSwarm', 'linearly_spaced_combinations',
           'sample_gp_function', 'SwarmOptimization', 'DeviceSwarmOptimization',
           'gpy', 'dist']
__version__ = "0.1.0"
