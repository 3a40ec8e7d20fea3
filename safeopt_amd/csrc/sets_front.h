// The last step of the front half of the fused single-rank chain (sets.hip): from the
// per-workgroup results of k_candidates_f to the first candidate of the whole shard
// (visiting order of safeopt/gp_opt.py:542-552), the totals, and the candidate staged
// as the operand of the expander test.  A device function, so that it runs either as
// a launch of its own (k_front_final: N ranks, no active constraint) or at the top of
// the FIRST kernel that needs its result (factor.hip:k_expkt, one rank): every
// workgroup of that kernel finds the first candidate for itself from the <= 1024
// partial results (28 KB, L2 hits), one of them writes the result block -- a launch
// and its gap less in the chain, no atomics, no fences.
#pragma once

#include "set_order.h"

struct FrontArgs {
  const unsigned* block_counts;   // [nb][2] candidates, unsafe rows
  const double* best_w;           // [nb] width of the workgroup's first candidate
  const int64_t* best_i;          // [nb] its global index (-1: none)
  const unsigned* best_ties;      // [nb] candidates of the workgroup with that width
  int nb;                         // 0: nothing to fold (the operand is staged already)
  const double* pts;              // [d][N]
  const double* mean;             // [G][N]
  const double* Q;                // [N][2 G]
  int64_t N, goff;
  int d, G;
  double* res;                    // result block (device), layout: sgp_grid_sets_fused
  double* res_host;               // the same block in mapped host memory, or null
  double* xc;                     // xc | resid[G][16]: n_xc_resid doubles
  int n_xc_resid;
  int32_t* flags;
  int n_flag_words;
};

// Returns the global index of the first candidate (-1: none) to every thread of the
// workgroup; `writer`: this workgroup also leaves the result block and the staged
// operand.  blockDim.x = 256; contains barriers (call it from uniform control flow).
__device__ __forceinline__ int64_t front_final_fold(const FrontArgs& a, bool writer) {
  constexpr int kT = 256;
  __shared__ Pair shp[kT / 64];
  __shared__ unsigned long long shc[2][kT / 64];
  __shared__ int64_t top;
  __shared__ double topw;
  unsigned long long ca = 0, cb = 0;
  Pair best{-INFINITY, -1};
  for (int e = threadIdx.x; e < a.nb; e += kT) {
    if (writer) {
      ca += a.block_counts[2 * e];
      cb += a.block_counts[2 * e + 1];
    }
    const Pair p{a.best_w[e], a.best_i[e]};
    if (p.i >= 0 && (best.i < 0 || before_desc(p, best))) best = p;
  }
  if (writer) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      ca += __shfl_xor(ca, o, 64);
      cb += __shfl_xor(cb, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      shc[0][threadIdx.x >> 6] = ca;
      shc[1][threadIdx.x >> 6] = cb;
    }
  }
  const Pair win = block_best<false>(best, shp);     // (syncs)
  if (!writer) return win.i;
  double* rh = a.res_host;
  for (int e = threadIdx.x; e < a.n_xc_resid; e += kT) a.xc[e] = 0.0;
  for (int e = threadIdx.x; e < a.n_flag_words; e += kT) a.flags[e] = 0;
  if (threadIdx.x == 0) {
    unsigned long long ta = 0, tb = 0;
    for (int wv = 0; wv < kT / 64; ++wv) {
      ta += shc[0][wv];
      tb += shc[1][wv];
    }
    reinterpret_cast<unsigned long long*>(a.res)[1] = ta;
    reinterpret_cast<unsigned long long*>(a.res)[2] = tb;
    a.res[3] = win.v;
    reinterpret_cast<int64_t*>(a.res)[4] = win.i;
    reinterpret_cast<int*>(a.res + 5)[0] = win.i >= 0 ? 1 : 0;
    if (rh) {
      rh[0] = a.res[0];           // max width (k_candidates_f)
      reinterpret_cast<unsigned long long*>(rh)[1] = ta;
      reinterpret_cast<unsigned long long*>(rh)[2] = tb;
      rh[3] = win.v;
      reinterpret_cast<int64_t*>(rh)[4] = win.i;
      reinterpret_cast<int*>(rh + 5)[0] = win.i >= 0 ? 1 : 0;
    }
    top = win.i;
    topw = win.v;
  }
  __syncthreads();
  if (top < 0) {
    if (threadIdx.x == 0) {
      reinterpret_cast<int*>(a.res + 5)[1] = 0;
      if (rh) reinterpret_cast<int*>(rh + 5)[1] = 0;
    }
    return -1;
  }
  {   // candidates of the whole shard that share the first one's width
    unsigned nt = 0;
    for (int e = threadIdx.x; e < a.nb; e += kT)
      if (a.best_i[e] >= 0 && a.best_w[e] == topw) nt += a.best_ties[e];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nt += __shfl_xor(nt, o, 64);
    if ((threadIdx.x & 63) == 0) shc[0][threadIdx.x >> 6] = nt;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int wv = 0; wv < kT / 64; ++wv) t += unsigned(shc[0][wv]);
      reinterpret_cast<int*>(a.res + 5)[1] = int(t);
      if (rh) reinterpret_cast<int*>(rh + 5)[1] = int(t);
    }
  }
  const int64_t li = top - a.goff;
  const int d = a.d, G = a.G;
  double* resid = a.xc + (a.n_xc_resid - G * 16);   // the block is xc | resid[G][16]
  for (int k = threadIdx.x; k < d; k += kT) {
    const double v = a.pts[int64_t(k) * a.N + li];
    a.res[6 + k] = v;
    if (rh) rh[6 + k] = v;
    a.xc[k] = v;
  }
  for (int g = threadIdx.x; g < G; g += kT) {
    const double mu = a.mean[int64_t(g) * a.N + li];
    const double up = a.Q[li * 2 * G + 2 * g + 1];
    a.res[6 + d + g] = mu;
    if (rh) rh[6 + d + g] = mu;
    resid[g * 16] = up - mu;
  }
  for (int q = threadIdx.x; q < 2 * G; q += kT) {
    const double v = a.Q[li * 2 * G + q];
    a.res[6 + d + G + q] = v;
    if (rh) rh[6 + d + G + q] = v;
  }
  return top;
}
