// Shared by the few-points posterior (factor.hip) and the small-swarm step
// (swarm.hip): layout of the partial results and the fixed-order block sum.
#pragma once
#include "common.h"

// Scratch of posterior_small_all, per GP g (strides in doubles):
//   Kb   + g * kb_stride   : [passes][n_pad_max / 4][64]  k(X, pts) as MFMA B operands
//   part + g * part_stride : [passes][nblk_max][16]       |L^-1 Kb|^2 per row block, point
//   mtmp + g * passes * 16 : [passes * 16]                alpha . Kb
struct SmallBufs {
  double* Kb;
  double* part;
  double* mtmp;
  int64_t kb_stride, part_stride;
  int nsteps_max, nblk_max, passes;
};

// Sum over the row blocks of one GP for the 16 points of `pass`, in a fixed
// order: thread (c = t & 15, q = t >> 4) of a 256-thread workgroup adds the row
// blocks q, q + 16, ...; the 16 groups are folded through LDS by the q == 0
// threads (a larger workgroup may run several such sums side by side: tl, sh).  Returns the total for point c in the threads with q == 0; every
// thread of the workgroup must call it (two barriers).
__device__ __forceinline__ double small_block_sum(const double* part_g, int nblk_max,
                                                  int nblk, int pass,
                                                  double (*sh)[16],
                                                  int tl = -1) {
  if (tl < 0) tl = threadIdx.x;             // (tl: index within a group of 256)
  const int c = tl & 15, q = tl >> 4;
  double ss = 0.0;
  for (int b = q; b < nblk; b += 16)
    ss += part_g[(int64_t(pass) * nblk_max + b) * 16 + c];
  __syncthreads();
  sh[q][c] = ss;
  __syncthreads();
  double tot = 0.0;
  if (q == 0) {
    tot = sh[0][c];
#pragma unroll
    for (int g = 1; g < 16; ++g) tot += sh[g][c];
  }
  return tot;
}

// factor.hip
int small_reserve(sgp_ctx* ctx, const GpDev* gps_host, int G, int P, SmallBufs* sb);
int posterior_small_all(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                        int G, const double* pts_rowmajor, int P, const SmallBufs& sb,
                        double* mean, double* var);   // mean == nullptr: no block sums
bool small_path_pays(const sgp_gp* gp, int64_t P);
// swarm.hip
int launch_fitness_small(sgp_ctx* ctx, int G, int64_t P, const double* mean,
                         const double* var, FitnessArgs fa);
struct PsoSmallArgs {
  double *pos, *vel, *best, *best_values, *gbest;   // (P, d) row-major state
  const double *vscale, *bounds;                    // bounds may be null
  const double* rand;                               // draws of the NEXT move, or null
  uint64_t seed;
  uint32_t draw;                                    // Philox stream of the next move
  double inertia;                                   // of the next move
  int P, d, init, move;
};
// block sums + fitness + personal / global bests + (optionally) the next move
// (post_mean / post_var: [G][P] posterior of a sweep instead of the block sums of sb)
int launch_pso_small_step(sgp_ctx* ctx, const GpDev* gps_dev, int G, const SmallBufs& sb,
                          FitnessArgs fa, PsoSmallArgs ps, const double* post_mean = nullptr,
                          const double* post_var = nullptr);
