"""Constrained particle swarm used by ``SafeOptSwarm`` (host side).

Same behaviour as ``/root/reference/safeopt/swarm.py:17-146``: the update
order, the two ``np.random.rand`` draws per call site (``:75`` once per
``init_swarm``, ``:104`` once per iteration, shape ``(2*swarm_size, ndim)``)
and the aliasing quirks (``positions`` is the caller's array, ``best_values``
is the fitness output, ``global_best`` is a view into ``best_positions``) are
kept, because the chosen point of ``SafeOptSwarm.optimize`` depends on them.
Only the fitness callback does arithmetic of any size, and that runs on the
GPU (``SafeOptSwarm._compute_particle_fitness``).
"""
from __future__ import annotations

import numpy as np

__all__ = ['SwarmOptimization', 'DeviceSwarmOptimization']


class SwarmOptimization(object):
    """Particle swarm maximising ``fitness`` subject to a safety mask.

    Parameters
    ----------
    swarm_size : int
    velocity : ndarray
        Velocity scale per dimension.
    fitness : callable
        ``fitness(positions) -> (values, safe_mask)``.
    bounds : list of (low, high), optional
        Box the particles are clipped to.
    """

    def __init__(self, swarm_size, velocity, fitness, bounds=None):
        self.c1 = self.c2 = 1
        self.fitness = fitness
        self.bounds = None if bounds is None else np.asarray(bounds)
        self.initial_inertia = 1.0
        self.final_inertia = 0.1
        self.velocity_scale = velocity
        self.ndim = len(velocity)
        self.swarm_size = swarm_size

        shape = (swarm_size, self.ndim)
        self.positions = np.empty(shape, dtype=float)
        self.velocities = np.empty(shape, dtype=float)
        self.best_positions = np.empty(shape, dtype=float)
        self.best_values = np.empty(swarm_size, dtype=float)
        self.global_best = None

    @property
    def max_velocity(self):
        """Velocity clip: ten times the velocity scale."""
        return 10 * self.velocity_scale

    def init_swarm(self, positions):
        """Start a run from ``positions`` (kept by reference).

        One ``np.random.rand(swarm_size, ndim)`` draw for the velocities; the
        global best is the arg-max of the raw fitness (the safety mask is not
        applied here, as in the reference)."""
        self.positions = positions
        draw = np.random.rand(*self.velocities.shape)
        self.velocities = draw * self.velocity_scale
        fit, _mask = self.fitness(self.positions)
        np.copyto(self.best_positions, self.positions)
        self.best_values = fit
        self._pick_global_best()

    def _pick_global_best(self):
        # a view into best_positions, first index among equal values
        self.global_best = self.best_positions[int(np.argmax(self.best_values)), :]

    def _move(self, inertia):
        """Velocity and position update of one iteration (one draw of
        ``np.random.rand(2 * swarm_size, ndim)``: own pull first, then global).
        The expression order is part of the contract -- bit-identical runs."""
        pull_global = self.global_best - self.positions
        pull_own = self.best_positions - self.positions
        u = np.random.rand(2 * self.swarm_size, self.ndim)
        u_own, u_global = u[:self.swarm_size], u[self.swarm_size:]
        v = self.velocities
        v *= inertia
        v += (self.c1 * u_own * pull_own +
              self.c2 * u_global * pull_global) / self.velocity_scale
        limit = self.max_velocity
        np.clip(v, -limit, limit, out=v)
        x = self.positions
        x += v
        if self.bounds is not None:
            np.clip(x, self.bounds[:, 0], self.bounds[:, 1], out=x)

    def _keep_improvements(self):
        """Personal bests move only to safe points with a higher fitness."""
        fit, mask = self.fitness(self.positions)
        take = (fit > self.best_values) & mask
        self.best_values[take] = fit[take]
        self.best_positions[take] = self.positions[take]
        self._pick_global_best()

    def run_swarm(self, max_iter):
        """Iterate the swarm ``max_iter`` times, inertia going linearly from
        ``initial_inertia`` towards ``final_inertia``."""
        step = (self.final_inertia - self.initial_inertia) / max_iter
        inertia = self.initial_inertia
        for _ in range(max_iter):
            self._move(inertia)
            inertia += step
            self._keep_improvements()

def _hip_swarm_types():
    from . import _hip
    return _hip.SWARM_TYPES


class DeviceSwarmOptimization(SwarmOptimization):
    """The same swarm with its state in HBM: ``init_swarm`` and ``run_swarm``
    are one C-ABI call each (``sgp_swarm_run``) -- velocity / position update,
    fused posterior fitness, personal and global bests all run on the GPU and
    nothing crosses PCIe between iterations.

    ``rng='numpy'`` (default): the uniform numbers are drawn with
    ``np.random.rand`` on the host exactly where and in the order the reference
    draws them (``swarm.py:75, 104``) and shipped with the call, so the run --
    and the state of NumPy's global generator afterwards -- is bit-identical to
    :class:`SwarmOptimization`.  ``rng='device'``: a counter-based generator
    on the GPU (Philox4x32-10); nothing but the swarm state is transferred,
    results are reproducible per ``seed`` but differ from ``np.random``.

    ``owner`` is the :class:`SafeOptSwarm` whose GPs / beta / fmin / scaling /
    best lower bound define the fitness of ``swarm_type``.
    """

    def __init__(self, swarm_size, velocity, owner, swarm_type, bounds=None,
                 rng='numpy', seed=None):
        super(DeviceSwarmOptimization, self).__init__(
            swarm_size, velocity, None, bounds=bounds)
        if rng not in ('numpy', 'device'):
            raise ValueError("rng must be 'numpy' or 'device'")
        self._owner = owner
        self._type = swarm_type
        self._rng = rng
        # Philox key of the device generator: `seed`, or one draw from NumPy's
        # global stream (so np.random.seed controls it and every swarm object gets
        # its own), mixed with the swarm type -- the greedy / maximizers /
        # expanders swarms of one optimiser must not draw the same numbers
        if rng == 'device' and seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self._seed = (int(seed or 0) * 4 + _hip_swarm_types()[swarm_type]) & (2 ** 43 - 1)
        self._calls = 0
        self.global_best = np.zeros(self.ndim)

    def fitness(self, positions):                 # kept for API parity
        return self._owner._compute_particle_fitness(self._type, positions)

    def _device_run(self, init, iters, inertia0, step):
        from . import _hip
        o = self._owner
        devs = [g._fitted() for g in o.gps]
        P, d = self.positions.shape
        if self._rng == 'numpy':
            rand = np.random.rand((P * d if init else 0) + 2 * P * d * iters)
        else:
            rand = None
        self._calls += 1
        _hip.swarm_run(
            devs[0].ctx, devs, self._type, o.beta(o.t), o.fmin, o.scaling,
            o.best_lower_bound, self.positions, self.velocities,
            self.best_positions, self.best_values, self.global_best,
            np.broadcast_to(self.velocity_scale, (d,)), self.bounds, init,
            iters, inertia0, step, rand,
            seed=(self._seed << 20) + self._calls)

    def init_swarm(self, positions):
        self.positions = np.ascontiguousarray(positions, dtype=float)
        shape = self.positions.shape
        self.velocities = np.empty(shape)
        self.best_positions = np.empty(shape)
        self.best_values = np.empty(shape[0])
        self.global_best = np.empty(shape[1])
        self._device_run(True, 0, self.initial_inertia, 0.0)

    def run_swarm(self, max_iter):
        step = (self.final_inertia - self.initial_inertia) / max_iter
        self._device_run(False, max_iter, self.initial_inertia, step)
