// SafeOptSwarm safe-set growth on the device.
//
// Replaces the step after every maximizer / expander swarm run
// (safeopt/gp_opt.py:1089-1111): with C = k(B, [S; B]) / scaling[0]^2 for the
// swarm's best positions B (n x d) against the current safe set S (m x d),
// candidate j is appended iff C[j, p] <= 0.95 for every point p of S and for
// every candidate accepted before it.  The reference materialises the
// n x (m + n) matrix on the host and loops over j; here
//   k_grow_old : max_p C[j, p] over S, tiled over (candidate, chunk of S)
//   k_grow_new : the order-dependent part -- one workgroup walks over the
//                candidates in order and tests each one against the accepted
//                list in parallel.
// The covariance matrix is never stored.
#include "kern_eval.h"
#include "fitness.h"
#include "small_path.h"

namespace {

constexpr int kGrowChunk = 4096;   // safe-set points per workgroup of k_grow_old

template <int D>
__global__ __launch_bounds__(256) void k_grow_old(KernDesc kd, const double* S,
                                                  int64_t m, const double* B,
                                                  double scale2, double* part,
                                                  int nchunks) {
  __shared__ double sh[4];
  const int j = blockIdx.y;
  const int64_t p0 = int64_t(blockIdx.x) * kGrowChunk;
  double b[D];
#pragma unroll
  for (int k = 0; k < D; ++k) b[k] = B[int64_t(j) * D + k];
  double mx = -INFINITY;
  for (int64_t p = p0 + threadIdx.x; p < min(p0 + kGrowChunk, m); p += 256) {
    double s[D];
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = S[p * D + k];
    mx = fmax(mx, kern_eval<D>(kd, b, s) / scale2);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0)
    part[int64_t(j) * nchunks + blockIdx.x] =
        fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

template <int D>
__global__ __launch_bounds__(1024) void k_grow_new(KernDesc kd, const double* B,
                                                   int n, const double* part,
                                                   int nchunks, double scale2,
                                                   double thr, int* list,
                                                   uint8_t* accept) {
  __shared__ int clash[1024 / 64];
  __shared__ int count;
  const int tid = threadIdx.x;
  if (tid == 0) count = 0;
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    // against the old safe set (all threads evaluate the same few partials)
    double mx = -INFINITY;
    for (int c = 0; c < nchunks; ++c) mx = fmax(mx, part[int64_t(j) * nchunks + c]);
    bool bad = !(mx <= thr);             // NaN rejects, as `<=` does in NumPy
    const int na = count;
    if (!bad) {
      double b[D];
#pragma unroll
      for (int k = 0; k < D; ++k) b[k] = B[int64_t(j) * D + k];
      for (int a = tid; a < na; a += 1024) {
        const int i = list[a];
        double o[D];
#pragma unroll
        for (int k = 0; k < D; ++k) o[k] = B[int64_t(i) * D + k];
        bad = bad || !(kern_eval<D>(kd, b, o) / scale2 <= thr);
      }
    }
    const unsigned long long any = __ballot(bad);
    if ((tid & 63) == 0) clash[tid >> 6] = any != 0ULL;
    __syncthreads();
    bool rej = false;
    for (int w = 0; w < 1024 / 64; ++w) rej = rej || clash[w];
    if (tid == 0) {
      accept[j] = rej ? 0 : 1;
      if (!rej) {
        list[na] = j;
        count = na + 1;
      }
    }
    __syncthreads();
  }
}

}  // namespace

int launch_swarm_grow(sgp_ctx* ctx, const KernDesc& kd, const double* S, int64_t m,
                      const double* B, int n, double scale2, double thr,
                      double* part, int* list, uint8_t* accept) {
  const int nchunks = int((m + kGrowChunk - 1) / kGrowChunk);
#define GROW_CASE(DD)                                                           \
  case DD:                                                                      \
    if (nchunks > 0)                                                            \
      hipLaunchKernelGGL(k_grow_old<DD>, dim3(nchunks, n), dim3(256), 0,        \
                         ctx->stream, kd, S, m, B, scale2, part, nchunks);      \
    hipLaunchKernelGGL(k_grow_new<DD>, dim3(1), dim3(1024), 0, ctx->stream, kd, \
                       B, n, part, nchunks, scale2, thr, list, accept);         \
    break;
  switch (kd.d) {
    GROW_CASE(1) GROW_CASE(2) GROW_CASE(3) GROW_CASE(4)
    GROW_CASE(5) GROW_CASE(6) GROW_CASE(7) GROW_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", kd.d, SGP_MAX_D);
      return -2;
  }
#undef GROW_CASE
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int swarm_grow_chunks(int64_t m) { return int((m + kGrowChunk - 1) / kGrowChunk); }

// ---- particle swarm on the device ---------------------------------------------------
// SwarmOptimization.init_swarm / run_swarm (safeopt/swarm.py:61-146) with the
// swarm state resident in HBM for the whole run; the fitness is the posterior
// sweep + shaping pass (launch_sweep_fitness) on the same buffers.  This file is
// built with -ffp-contract=off and the update below mirrors NumPy's evaluation
// order, so that with the reference's own random numbers (rand != nullptr,
// drawn by np.random.rand on the host in the reference's order) the run is
// bit-identical to the host implementation.
namespace {

// Philox4x32-10 (Salmon et al., SC'11): counter-based, no state to keep.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = uint64_t(0xD2511F53u) * c[0];
  const uint64_t p1 = uint64_t(0xCD9E8D57u) * c[2];
  const uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = uint32_t(p0 >> 32) ^ c[3] ^ k1;
  c[1] = uint32_t(p1);
  c[3] = uint32_t(p0);
  c[0] = n0;
  c[2] = n2;
}

// e-th uniform double in [0, 1) of stream (seed, draw): 53 random bits.
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t draw,
                                                 uint64_t e) {
  uint32_t c[4] = {uint32_t(e >> 1), uint32_t(e >> 33), draw, 0x5afe0b7u};
  uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint64_t bits = (e & 1) ? ((uint64_t(c[2]) << 32) | c[3])
                                : ((uint64_t(c[0]) << 32) | c[1]);
  return double(bits >> 11) * (1.0 / 9007199254740992.0);
}

// velocities = rand(P, d) * velocity_scale          (swarm.py:75-76)
__global__ void k_pso_init_vel(int64_t P, int d, double* vel,
                               const double* vscale, const double* rand,
                               uint64_t seed) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= P * d) return;
  const double r = rand ? rand[e] : philox_uniform(seed, 0u, uint64_t(e));
  vel[e] = r * vscale[e % d];
}

// one velocity / position update                     (swarm.py:98-123)
__global__ void k_pso_move(int64_t P, int d, double* pos, double* vel,
                           const double* best, const double* gbest,
                           const double* vscale, const double* bounds,
                           double inertia, const double* rand, uint64_t seed,
                           uint32_t draw) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= P * d) return;
  const int k = int(e % d);
  const double x = pos[e];
  const double to_global = gbest[k] - x;
  const double to_own = best[e] - x;
  // r = rand(2 P, d); r1 = r[:P], r2 = r[P:]
  const double r1 = rand ? rand[e] : philox_uniform(seed, draw, uint64_t(e));
  const double r2 = rand ? rand[P * d + e]
                         : philox_uniform(seed, draw, uint64_t(P * d + e));
  double v = vel[e] * inertia;
  v = v + (r1 * to_own + r2 * to_global) / vscale[k];
  const double vmax = 10.0 * vscale[k];
  v = fmin(fmax(v, -vmax), vmax);
  vel[e] = v;
  double xn = x + v;
  if (bounds) xn = fmin(fmax(xn, bounds[2 * k]), bounds[2 * k + 1]);
  pos[e] = xn;
}

// personal bests                                     (swarm.py:77-79, 138-143)
__global__ void k_pso_best(int64_t P, int d, const double* values,
                           const uint8_t* safe, const double* pos, double* best,
                           double* best_values, int init) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const bool better = init || (values[i] > best_values[i] && safe[i]);
  if (better) {
    best_values[i] = values[i];
    for (int k = 0; k < d; ++k) best[i * d + k] = pos[i * d + k];
  }
}

// global_best = best_positions[argmax(best_values)], first index on ties
__global__ __launch_bounds__(1024) void k_pso_gbest(int64_t P, int d,
                                                    const double* best_values,
                                                    const double* best,
                                                    double* gbest) {
  __shared__ double sv[1024 / 64];
  __shared__ long long si[1024 / 64];
  double v = -INFINITY;
  long long idx = -1;
  for (int64_t i = threadIdx.x; i < P; i += 1024) {
    const double x = best_values[i];
    if (idx < 0 || x > v) {      // strided scan keeps the lowest index per thread
      v = x;
      idx = i;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double ov = __shfl_xor(v, o, 64);
    const long long oi = __shfl_xor(idx, o, 64);
    if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) {
      v = ov;
      idx = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 1024 / 64; ++w)
      if (si[w] >= 0 && (idx < 0 || sv[w] > v || (sv[w] == v && si[w] < idx))) {
        v = sv[w];
        idx = si[w];
      }
    for (int k = 0; k < d; ++k) gbest[k] = best[idx * d + k];
  }
}

// Fitness of a few particles from resident mean / var ([G][P]): the path of
// SafeOptSwarm._compute_particle_fitness for P <= kSmallPoints.
__global__ void k_fitness_small(int G, int64_t P, const double* mean,
                                const double* var, FitnessArgs f) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= P) return;
  double out;
  bool ok;
  shape_particle(f, G,
                 [&](int g, double* mu, double* v) {
                   *mu = mean[int64_t(g) * P + i];
                   *v = var[int64_t(g) * P + i];
                 },
                 &out, &ok);
  f.values[i] = out;
  f.safe[i] = ok ? 1 : 0;
}

// One PSO iteration of a small swarm (P <= kSmallSwarm) behind the two posterior launches
// (k_small_kb, k_small_mfma), in ONE workgroup: block sums -> mean / var of every
// GP, fitness, personal bests, global best (first index on ties) and the move
// that opens the next iteration.  Same arithmetic, statement for statement, as
// k_small_post + k_fitness_small + k_pso_best + k_pso_gbest + k_pso_move.
// post_mean != nullptr: the posterior comes from a sweep ([G][P] mean | var, the GPs with
// fewer observations than the few-points path wants): no block sums, the rest as above.
__global__ __launch_bounds__(1024) void k_pso_small_step(const GpDev* gps, int G,
                                                         SmallBufs sb, FitnessArgs f,
                                                         PsoSmallArgs ps,
                                                         const double* post_mean,
                                                         const double* post_var) {
  __shared__ double sh[4][16][16];
  __shared__ double smean[SGP_MAX_GPS][kSmallSwarm], svar[SGP_MAX_GPS][kSmallSwarm];
  __shared__ double sbv[kSmallSwarm];
  const int t = threadIdx.x, P = ps.P, d = ps.d;
  const int Geff = (f.swarm_type == SGP_SWARM_GREEDY) ? 1 : G;
  // block sums of (GP, pass) pairs, four pairs side by side
  const int grp = t >> 8, tl = t & 255, npairs = post_mean ? 0 : Geff * sb.passes;
  if (post_mean) {
    for (int e = t; e < Geff * P; e += 1024) {
      smean[e / P][e % P] = post_mean[e];
      svar[e / P][e % P] = post_var[e];
    }
  }
  for (int base = 0; base < npairs; base += 4) {
    const int pair = base + grp;
    const bool valid = pair < npairs;
    const int g = valid ? pair / sb.passes : 0, pass = valid ? pair % sb.passes : 0;
    const double tot = small_block_sum(sb.part + g * sb.part_stride, sb.nblk_max,
                                       valid ? gps[g].nblk : 0, pass, sh[grp], tl);
    const int p = pass * 16 + (tl & 15);
    if (valid && (tl >> 4) == 0 && p < P) {
      smean[g][p] = sb.mtmp[g * sb.passes * 16 + p];
      svar[g][p] = fmax(gps[g].kern.kdiag - tot, 1e-15);      // GPy clip
    }
  }
  __syncthreads();
  if (t < P) {
    double out;
    bool ok;
    shape_particle(f, G,
                   [&](int g, double* mu, double* v) {
                     *mu = smean[g][t];
                     *v = svar[g][t];
                   },
                   &out, &ok);
    f.values[t] = out;
    f.safe[t] = ok ? 1 : 0;
    // personal bests (k_pso_best)
    const bool better = ps.init || (out > ps.best_values[t] && ok);
    double bv = ps.best_values[t];
    if (better) {
      bv = out;
      ps.best_values[t] = out;
      for (int k = 0; k < d; ++k) ps.best[t * d + k] = ps.pos[t * d + k];
    }
    sbv[t] = bv;
  }
  __syncthreads();
  // global best: argmax of the personal bests, first index on ties (k_pso_gbest)
  if (t < 64) {
    double v = t < P ? sbv[t] : -INFINITY;
    long long idx = t < P ? t : -1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double ov = __shfl_xor(v, o, 64);
      const long long oi = __shfl_xor(idx, o, 64);
      if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) {
        v = ov;
        idx = oi;
      }
    }
    if (t < d) ps.gbest[t] = ps.best[idx * d + t];
  }
  if (!ps.move) return;
  __syncthreads();
  // the move that opens the next iteration (k_pso_move)
  for (int e = t; e < P * d; e += 1024) {
    const int k = e % d;
    const double x = ps.pos[e];
    const double to_global = ps.gbest[k] - x;
    const double to_own = ps.best[e] - x;
    const double r1 = ps.rand ? ps.rand[e] : philox_uniform(ps.seed, ps.draw, uint64_t(e));
    const double r2 = ps.rand ? ps.rand[P * d + e]
                              : philox_uniform(ps.seed, ps.draw, uint64_t(P * d + e));
    double v = ps.vel[e] * ps.inertia;
    v = v + (r1 * to_own + r2 * to_global) / ps.vscale[k];
    const double vmax = 10.0 * ps.vscale[k];
    v = fmin(fmax(v, -vmax), vmax);
    ps.vel[e] = v;
    double xn = x + v;
    if (ps.bounds) xn = fmin(fmax(xn, ps.bounds[2 * k]), ps.bounds[2 * k + 1]);
    ps.pos[e] = xn;
  }
}

}  // namespace

int launch_pso_init_vel(sgp_ctx* ctx, int64_t P, int d, double* vel,
                        const double* vscale, const double* rand, uint64_t seed) {
  hipLaunchKernelGGL(k_pso_init_vel, dim3(unsigned((P * d + 255) / 256)),
                     dim3(256), 0, ctx->stream, P, d, vel, vscale, rand, seed);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_pso_move(sgp_ctx* ctx, int64_t P, int d, double* pos, double* vel,
                    const double* best, const double* gbest, const double* vscale,
                    const double* bounds, double inertia, const double* rand,
                    uint64_t seed, uint32_t draw) {
  hipLaunchKernelGGL(k_pso_move, dim3(unsigned((P * d + 255) / 256)), dim3(256), 0,
                     ctx->stream, P, d, pos, vel, best, gbest, vscale, bounds,
                     inertia, rand, seed, draw);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_pso_best(sgp_ctx* ctx, int64_t P, int d, const double* values,
                    const uint8_t* safe, const double* pos, double* best,
                    double* best_values, double* gbest, int init) {
  hipLaunchKernelGGL(k_pso_best, dim3(unsigned((P + 255) / 256)), dim3(256), 0,
                     ctx->stream, P, d, values, safe, pos, best, best_values, init);
  hipLaunchKernelGGL(k_pso_gbest, dim3(1), dim3(1024), 0, ctx->stream, P, d,
                     best_values, best, gbest);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_fitness_small(sgp_ctx* ctx, int G, int64_t P, const double* mean,
                         const double* var, FitnessArgs fa) {
  hipLaunchKernelGGL(k_fitness_small, dim3(unsigned((P + 255) / 256)), dim3(256),
                     0, ctx->stream, G, P, mean, var, fa);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_pso_small_step(sgp_ctx* ctx, const GpDev* gps_dev, int G, const SmallBufs& sb,
                          FitnessArgs fa, PsoSmallArgs ps, const double* post_mean,
                          const double* post_var) {
  hipLaunchKernelGGL(k_pso_small_step, dim3(1), dim3(1024), 0, ctx->stream, gps_dev, G,
                     sb, fa, ps, post_mean, post_var);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}
