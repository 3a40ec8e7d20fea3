// The fitness shaping of SafeOptSwarm._compute_particle_fitness
// (gp_opt.py:925-1013) and SafeOptSwarm._compute_penalty (gp_opt.py:874-899),
// shared by k_fitness_small (swarm.hip: the pass behind the posterior sweep of a
// fitness call, launch_sweep in sweep.hip) and the few-points kernels (swarm.hip).
#pragma once
#include "common.h"

// SafeOptSwarm._compute_penalty (gp_opt.py:874-899) for one value.
__device__ __forceinline__ double swarm_penalty(double slack) {
  double pen = fmin(slack, 0.0);
  if (slack < 0.0 && slack > -0.001) pen *= 2.0;
  if (slack <= -0.001 && slack > -0.1) pen *= 5.0;
  if (slack <= -0.1 && slack > -1.0) pen *= 10.0;
  if (slack < -1.0) pen = -300.0 * pen * pen;
  return pen;
}

// One particle from the posterior of its G GPs (post(g, &mu, &var)): the value
// the swarm maximises and the particle's safety flag.  Same formulas, same order
// as the sweep's epilogue.
template <typename Post>
__device__ __forceinline__ void shape_particle(const FitnessArgs& f, int G, Post post,
                                               double* value, bool* is_safe) {
  const int st = f.swarm_type;
  const int Geff = (st == SGP_SWARM_GREEDY) ? 1 : G;
  bool safe = true;
  double values = 0.0, interest = 1.0, total_pen = 0.0, lower = 0.0;
  for (int g = 0; g < Geff; ++g) {
    double mu, var;
    post(g, &mu, &var);
    const double sd = sqrt(var);
    lower = mu - f.beta * sd;
    if (g == 0) {
      values = sd / f.scaling[0];
      if (st == SGP_SWARM_EXPANDERS) interest = double(G);
      if (st == SGP_SWARM_MAXIMIZERS) {
        const double upper = mu + f.beta * sd;
        const double z = 10.0 * (upper - f.best_lower_bound) / f.scaling[0];
        interest = 1.0 / (1.0 + exp(-z));  // scipy.special.expit
      }
    } else {
      values = fmax(values, sd / f.scaling[g]);
    }
    if (f.fmin[g] != -INFINITY) {
      double slack = lower - f.fmin[g];
      safe = safe && (slack >= 0.0);
      if (st != SGP_SWARM_SAFE_SET) {
        slack = slack / f.scaling[g];
        total_pen += swarm_penalty(slack);
        if (st == SGP_SWARM_EXPANDERS) {
          const double z = slack / 0.2;   // scipy.stats.norm.pdf(slack, scale=0.2)
          interest *= exp(-0.5 * z * z) / 2.5066282746310002 / 0.2;
        }
      }
    }
  }
  if (st == SGP_SWARM_GREEDY) {
    *value = lower;
    *is_safe = true;
  } else if (st == SGP_SWARM_SAFE_SET) {
    *value = lower;
    *is_safe = safe;
  } else {
    *value = (values + total_pen) * interest;
    *is_safe = safe;
  }
}
