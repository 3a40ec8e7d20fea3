// Orderings and workgroup reductions of the set passes (sets.hip) and of the
// one-workgroup step (step_small.hip): the visiting order of the expander candidates
// (safeopt/gp_opt.py:542-552: width descending, among equal widths the larger index
// first) and np.argmax (first index among equals, gp_opt.py:635, 642-644).
#pragma once

#include "kern_eval.h"

struct Pair {
  double v;
  int64_t i;
};

// ordering used for the expander visiting order: larger w first, ties ->
// larger global index first (a stable ascending sort, reversed)
__device__ __forceinline__ bool before_desc(const Pair& a, const Pair& b) {
  return a.v > b.v || (a.v == b.v && a.i > b.i);
}
// ordering of np.argmax: larger value first, ties -> smaller index first
__device__ __forceinline__ bool before_first(const Pair& a, const Pair& b) {
  if (b.i < 0) return a.i >= 0;
  if (a.i < 0) return false;
  return a.v > b.v || (a.v == b.v && a.i < b.i);
}

template <bool FIRST>
__device__ __forceinline__ Pair block_best(Pair p, Pair* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    Pair q;
    q.v = __shfl_xor(p.v, o, 64);
    q.i = __shfl_xor(p.i, o, 64);
    if (FIRST ? before_first(q, p) : before_desc(q, p)) p = q;
  }
  __syncthreads();
  if (lane == 0) sh[wave] = p;
  __syncthreads();
  Pair best = sh[0];
  const int nw = blockDim.x >> 6;
  for (int w = 1; w < nw; ++w) {
    const Pair q = sh[w];
    if (FIRST ? before_first(q, best) : before_desc(q, best)) best = q;
  }
  return best;
}

__device__ __forceinline__ double block_max(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double m = sh[0];
  const int nw = blockDim.x >> 6;
  for (int w = 1; w < nw; ++w) m = fmax(m, sh[w]);
  return m;
}

struct Vec8 {
  double v[SGP_MAX_GPS];
};
