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

__all__ = ['SwarmOptimization']


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
        """Start a run from ``positions`` (kept by reference)."""
        self.positions = positions
        self.velocities = (np.random.rand(*self.velocities.shape) *
                           self.velocity_scale)
        values, _safe = self.fitness(self.positions)
        self.best_positions[:] = self.positions
        self.best_values = values
        self.global_best = self.best_positions[np.argmax(values), :]

    def run_swarm(self, max_iter):
        """Iterate the swarm ``max_iter`` times."""
        inertia = self.initial_inertia
        step = (self.final_inertia - self.initial_inertia) / max_iter

        for _ in range(max_iter):
            to_global = self.global_best - self.positions
            to_own = self.best_positions - self.positions

            r = np.random.rand(2 * self.swarm_size, self.ndim)
            r1, r2 = r[:self.swarm_size], r[self.swarm_size:]

            self.velocities *= inertia
            self.velocities += ((self.c1 * r1 * to_own +
                                 self.c2 * r2 * to_global) /
                                self.velocity_scale)
            inertia += step

            np.clip(self.velocities, -self.max_velocity, self.max_velocity,
                    out=self.velocities)
            self.positions += self.velocities
            if self.bounds is not None:
                np.clip(self.positions, self.bounds[:, 0], self.bounds[:, 1],
                        out=self.positions)

            values, safe = self.fitness(self.positions)

            better = values > self.best_values
            better &= safe
            self.best_values[better] = values[better]
            self.best_positions[better] = self.positions[better]

            self.global_best = self.best_positions[
                np.argmax(self.best_values), :]
