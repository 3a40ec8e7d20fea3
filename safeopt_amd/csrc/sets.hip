// HBM-bound passes of SafeOpt.compute_sets / get_new_query_point
// (safeopt/gp_opt.py:478-649): safe set, maximisers, candidate-expander mask,
// visiting order (top-k by interval width), Lipschitz scan, masked arg-max.
// Every comparison keeps the reference's strictness (> for S, >= for M, > for
// both candidate filters, >= for the expander tests) and NumPy's IEEE
// expression order; this file is compiled with -ffp-contract=off so the masks
// are bit-exact with the NumPy restatement for identical Q.
#include "kern_eval.h"
#include "set_order.h"
#include "sets_front.h"

namespace {

constexpr int T = 256;
static __device__ __forceinline__ double fmin2(double a, double b) { return ::fmin(a, b); }

// S = all(Q[:, ::2] > fmin); partial[block] = max l0 over safe rows
__global__ __launch_bounds__(T) void k_safe_set(const double* Q, int64_t N,
                                                int G, Vec8 fmin, uint8_t* S,
                                                double* partial) {
  __shared__ double sh[T / 64];
  const int64_t i = int64_t(blockIdx.x) * T + threadIdx.x;
  double v = -INFINITY;
  if (i < N) {
    bool safe = true;
    for (int g = 0; g < G; ++g) safe = safe && (Q[(i * G + g) * 2] > fmin.v[g]);
    S[i] = safe ? 1 : 0;
    if (safe) v = Q[i * G * 2];
  }
  const double m = block_max(v, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = m;
}

// M = S & (u0 >= max_l); partial = max(u0 - l0) over M
__global__ __launch_bounds__(T) void k_maximizers(const double* Q,
                                                  const uint8_t* S, int64_t N,
                                                  int G, double max_l,
                                                  const double* max_l_dev,
                                                  uint8_t* M, double* partial) {
  __shared__ double sh[T / 64];
  if (max_l_dev) max_l = max_l_dev[0];   // still on the device (deferred sync)
  const int64_t i = int64_t(blockIdx.x) * T + threadIdx.x;
  double v = -INFINITY;
  if (i < N) {
    const double l0 = Q[i * G * 2], u0 = Q[i * G * 2 + 1];
    const bool m = S[i] && (u0 >= max_l);
    M[i] = m ? 1 : 0;
    if (m) v = u0 - l0;
  }
  const double mx = block_max(v, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = mx;
}

// candidate mask + widths; counts[0] += #candidates, counts[1] += #unsafe
__global__ __launch_bounds__(T) void k_candidates(
    const double* Q, const uint8_t* S, const uint8_t* M, int64_t N, int G,
    double max_var, const double* max_width_dev, Vec8 scaling, Vec8 thr_beta,
    int full_sets, uint8_t* cand, double* w, uint8_t* Gm,
    unsigned* block_counts) {
  __shared__ unsigned shc[2 * (T / 64)];
  // single-rank fast path: max(u0[M]-l0[M]) is still on the device
  if (max_width_dev) max_var = max_width_dev[0] / scaling.v[0];
  const int64_t i = int64_t(blockIdx.x) * T + threadIdx.x;
  bool c = false, unsafe = false;
  if (i < N) {
    const bool s = S[i] != 0;
    unsafe = !s;
    double wmax = -INFINITY;
    if (s) {
      double smax = -INFINITY;
      bool above = false;
      for (int g = 0; g < G; ++g) {
        const double width = Q[(i * G + g) * 2 + 1] - Q[(i * G + g) * 2];
        wmax = fmax(wmax, width);
        smax = fmax(smax, width / scaling.v[g]);
        above = above || (width > thr_beta.v[g]);
      }
      c = full_sets ? true : (!M[i] && (smax > max_var) && above);
    }
    cand[i] = c ? 1 : 0;
    w[i] = wmax;
    Gm[i] = 0;
  }
  // one (candidates, unsafe) pair per block; summed by k_sum_counts
  const unsigned long long bc = __ballot(c), bu = __ballot(unsafe);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    shc[2 * wave] = unsigned(__popcll(bc));
    shc[2 * wave + 1] = unsigned(__popcll(bu));
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned t = 0;
    for (int wv = 0; wv < T / 64; ++wv) t += shc[2 * wv + threadIdx.x];
    block_counts[2 * blockIdx.x + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void k_sum_counts(const unsigned* bc,
                                                     int64_t nblocks,
                                                     unsigned long long* out) {
  __shared__ unsigned long long sh[2][1024 / 64];
  unsigned long long a = 0, b = 0;
  for (int64_t e = threadIdx.x; e < nblocks; e += blockDim.x) {
    a += bc[2 * e];
    b += bc[2 * e + 1];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = a;
    sh[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned long long t = 0;
    for (int wv = 0; wv < 1024 / 64; ++wv) t += sh[threadIdx.x][wv];
    out[threadIdx.x] = t;
  }
}

// ---- top-k: the next k elements after the cut, in visiting order ----------------
// level 1 (src_idx == nullptr): elements are grid rows [chunk of the block],
//   key w[i] (or -gidx when index_key), valid = cand[i]
// level 2: elements are (src_w, src_idx) pairs, valid = idx >= 0; single block
constexpr int TK_CHUNK = 4096;
__global__ __launch_bounds__(T) void k_topk(const uint8_t* cand,
                                            const double* src_w,
                                            const int64_t* src_idx, int64_t n,
                                            int64_t goff, int index_key,
                                            double cut_w, int64_t cut_idx,
                                            int k, double* out_w,
                                            int64_t* out_idx, int* n_out) {
  __shared__ Pair sh[T / 64];
  int64_t begin, end;
  if (src_idx == nullptr) {
    begin = int64_t(blockIdx.x) * TK_CHUNK;
    end = min(n, begin + TK_CHUNK);
  } else {
    begin = 0;
    end = n;
  }
  Pair cut{cut_w, cut_idx};
  int found = 0;
  for (int r = 0; r < k; ++r) {
    Pair best{-INFINITY, -1};
    for (int64_t e = begin + threadIdx.x; e < end; e += T) {
      Pair p;
      if (src_idx == nullptr) {
        if (!cand[e]) continue;
        p.i = goff + e;
        p.v = index_key ? -double(p.i) : src_w[e];
      } else {
        p.i = src_idx[e];
        if (p.i < 0) continue;
        p.v = src_w[e];
      }
      // strictly after the cut in visiting order
      if (!(p.v < cut.v || (p.v == cut.v && p.i < cut.i))) continue;
      if (best.i < 0 || before_desc(p, best)) best = p;
    }
    // empty slots carry (-inf, -1); make them lose against every real entry
    Pair win = block_best<false>(best, sh);
    if (threadIdx.x == 0) {
      out_w[int64_t(blockIdx.x) * k + r] = win.v;
      out_idx[int64_t(blockIdx.x) * k + r] = win.i;
    }
    if (win.i >= 0) {
      ++found;
      cut = win;
    } else {
      // nothing left: the remaining slots are empty (wave-uniform exit)
      if (threadIdx.x == 0) {
        for (int rr = r + 1; rr < k; ++rr) {
          out_w[int64_t(blockIdx.x) * k + rr] = -INFINITY;
          out_idx[int64_t(blockIdx.x) * k + rr] = -1;
        }
      }
      break;
    }
  }
  if (n_out && threadIdx.x == 0) *n_out = found;
}

// Lipschitz expander test (gp_opt.py:558-576)
__global__ __launch_bounds__(T) void k_lipschitz(
    const double* pts, const uint8_t* S, int64_t N, int d, int G, Vec8 fmin,
    Vec8 lips, int m, const double* xc, const double* uc, int32_t* flags) {
  const int64_t i = int64_t(blockIdx.x) * T + threadIdx.x;
  const bool unsafe = (i < N) && (S[i] == 0);
  double x[SGP_MAX_D];
  for (int k = 0; k < d; ++k) x[k] = unsafe ? pts[int64_t(k) * N + i] : 0.0;
  for (int c = 0; c < m; ++c) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) {
      const double df = xc[c * d + k] - x[k];
      s += df * df;
    }
    const double dist = sqrt(s);
    for (int g = 0; g < G; ++g) {
      if (fmin.v[g] == -INFINITY) continue;
      const bool hit = unsafe && (uc[c * G + g] - lips.v[g] * dist >= fmin.v[g]);
      const unsigned long long b = __ballot(hit);
      if (b != 0ull && (threadIdx.x & 63) == 0) atomicOr(&flags[c * G + g], 1);
    }
  }
}

// masked arg-max, first index wins.  level 1: rows; level 2: partial pairs
__global__ __launch_bounds__(T) void k_argmax(const double* Q, const uint8_t* S,
                                              const uint8_t* M,
                                              const uint8_t* Gm, int64_t N,
                                              int G, int64_t goff, int mode,
                                              Vec8 scaling, double* out_v,
                                              int64_t* out_i) {
  __shared__ Pair sh[T / 64];
  Pair best{-INFINITY, -1};
  const int64_t begin = int64_t(blockIdx.x) * (T * 4);
  for (int r = 0; r < 4; ++r) {
    const int64_t i = begin + r * T + threadIdx.x;
    if (i >= N) continue;
    Pair p{0.0, goff + i};
    if (mode == SGP_ARGMAX_MG_WIDTH) {
      if (!(M[i] || Gm[i])) continue;
      double v = -INFINITY;
      for (int g = 0; g < G; ++g)
        v = fmax(v, (Q[(i * G + g) * 2 + 1] - Q[(i * G + g) * 2]) / scaling.v[g]);
      p.v = v;
    } else {
      if (!S[i]) continue;
      p.v = Q[i * G * 2 + (mode == SGP_ARGMAX_UCB ? 1 : 0)];
    }
    if (before_first(p, best)) best = p;
  }
  const Pair win = block_best<true>(best, sh);
  if (threadIdx.x == 0) {
    out_v[blockIdx.x] = win.v;
    out_i[blockIdx.x] = win.i;
  }
}

__global__ __launch_bounds__(T) void k_argmax_final(const double* in_v,
                                                    const int64_t* in_i,
                                                    int64_t n, double* out_v,
                                                    int64_t* out_i) {
  __shared__ Pair sh[T / 64];
  Pair best{-INFINITY, -1};
  for (int64_t e = threadIdx.x; e < n; e += T) {
    const Pair p{in_v[e], in_i[e]};
    if (before_first(p, best)) best = p;
  }
  const Pair win = block_best<true>(best, sh);
  if (threadIdx.x == 0) {
    out_v[0] = win.v;
    out_i[0] = win.i;
  }
}

__global__ void k_reduce_max(const double* in, int64_t n, double* out) {
  __shared__ double sh[1024 / 64];
  double v = -INFINITY;
  for (int64_t e = threadIdx.x; e < n; e += blockDim.x) v = fmax(v, in[e]);
  const double m = block_max(v, sh);
  if (threadIdx.x == 0) out[0] = m;
}

__global__ void k_fill_cols(double* pts, int64_t N, int d, int nc, Vec8 c) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int t = 0; t < nc; ++t) pts[int64_t(d - nc + t) * N + i] = c.v[t];
}

__global__ void k_gather_rows(const double* pts, const double* mean,
                              const double* var, const double* Q, int64_t N,
                              int d, int G, const int64_t* lidx, int m,
                              double* x, double* mo, double* vo, double* qo) {
  const int j = blockIdx.x;
  if (j >= m) return;
  const int64_t li = lidx[j];
  for (int k = threadIdx.x; k < d; k += blockDim.x)
    x[j * d + k] = pts[int64_t(k) * N + li];
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    mo[j * G + g] = mean[int64_t(g) * N + li];
    vo[j * G + g] = var[int64_t(g) * N + li];
  }
  for (int q = threadIdx.x; q < 2 * G; q += blockDim.x)
    qo[j * 2 * G + q] = Q[li * 2 * G + q];
}

// rows of ONE candidate whose GLOBAL index is still on the device (-1: none)
__global__ void k_gather_top(const double* pts, const double* mean,
                             const double* Q, int64_t N, int d, int G,
                             const int64_t* gidx, int64_t goff, double* x,
                             double* mo, double* qo) {
  const int64_t gi = gidx[0];
  if (gi < 0) return;
  const int64_t li = gi - goff;
  for (int k = threadIdx.x; k < d; k += blockDim.x)
    x[k] = pts[int64_t(k) * N + li];
  for (int g = threadIdx.x; g < G; g += blockDim.x)
    mo[g] = mean[int64_t(g) * N + li];
  for (int q = threadIdx.x; q < 2 * G; q += blockDim.x)
    qo[q] = Q[li * 2 * G + q];
}

// G[row] = 1 when every active GP certified the (single) candidate
__global__ void k_mark_if(uint8_t* Gm, int64_t li, const int32_t* flags, int G,
                          Vec8 fmin) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  bool ok = true, any = false;
  for (int g = 0; g < G; ++g) {
    if (fmin.v[g] == -INFINITY) continue;
    any = true;
    ok = ok && (flags[g] != 0);
  }
  if (ok && any) Gm[li] = 1;
}

// Fused single-GPU path: the first candidate (found by k_topk / k_gather_top
// into the result block) becomes the operand of the expander test without a
// host round trip.  xc[d] = its row, resid[g * 16] = u_g(x_c) - mu_g(x_c).
__global__ void k_stage_top(const double* x_top, const double* mean_top,
                            const double* q_top, int d, int G, double* xc,
                            double* resid) {
  const int t = threadIdx.x;
  if (t < d) xc[t] = x_top[t];
  if (t < G) resid[t * 16] = q_top[2 * t + 1] - mean_top[t];
}

// A batch of candidates (the next k in visiting order, k_topk: global indices, -1 / beyond
// *nfound = none) becomes the operand block of the expander test on the device: xc[c][d] =
// the rows, resid[g * 16 + c] = u_g - mu_g, everything else of the block and the flags zero.
__global__ __launch_bounds__(256) void k_stage_batch(
    const int64_t* gidx, const int* nfound, int K, const double* pts, const double* mean,
    const double* Q, int64_t N, int d, int G, int64_t goff, double* xc, int n_xc_resid,
    int32_t* flags, int n_flag_words) {
  for (int e = threadIdx.x; e < n_xc_resid; e += 256) xc[e] = 0.0;
  for (int e = threadIdx.x; e < n_flag_words; e += 256) flags[e] = 0;
  __syncthreads();
  const int m = *nfound < K ? *nfound : K;
  double* resid = xc + (n_xc_resid - G * 16);      // the block is xc | resid[G][16]
  for (int e = threadIdx.x; e < m * d; e += 256) {
    const int c = e / d, k = e - c * d;
    xc[e] = pts[int64_t(k) * N + (gidx[c] - goff)];
  }
  for (int e = threadIdx.x; e < m * G; e += 256) {
    const int c = e / G, g = e - c * G;
    const int64_t li = gidx[c] - goff;
    resid[g * 16 + c] = Q[li * 2 * G + 2 * g + 1] - mean[int64_t(g) * N + li];
  }
}

// ... and G[that row] = 1 when a candidate was found and every active GP
// certified it.
__global__ void k_mark_top_if(uint8_t* Gm, const int64_t* gidx, const int* nfound,
                              int64_t goff, const int32_t* flags, int G,
                              Vec8 fmin) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*nfound <= 0) return;
  bool ok = true, any = false;
  for (int g = 0; g < G; ++g) {
    if (fmin.v[g] == -INFINITY) continue;
    any = true;
    ok = ok && (flags[g] != 0);
  }
  if (ok && any) Gm[gidx[0] - goff] = 1;
}

__global__ void k_mark(uint8_t* Gm, const int64_t* lidx, int m, int value) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) Gm[lidx[j]] = uint8_t(value);
}

__global__ void k_import_points(const double* src, int64_t N, int d,
                                int64_t stride_row, int64_t stride_col,
                                double* dst) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = 0; k < d; ++k)
    dst[int64_t(k) * N + i] = src[i * stride_row + k * stride_col];
}

inline Vec8 vec8(const double* p, int n, double fill) {
  Vec8 v;
  for (int i = 0; i < SGP_MAX_GPS; ++i) v.v[i] = (p && i < n) ? p[i] : fill;
  return v;
}

// ---- fused single-rank chain of compute_sets (gp_opt.py:511-552, 611-615, 635-649)
// The passes below run on at most kFrontBlocks workgroups with grid-stride row
// loops, so a pass leaves <= 1024 partial results and the NEXT pass folds them
// itself (every workgroup re-reads 8 KB from L2) -- no reduction launches, no
// device-to-device copies between them.
constexpr int kFrontBlocks = 1024;

// M = S & (u0 >= max l0[S]); wpart[block] = max(u0 - l0) over the block's M.
// max l0[S] comes from (in this order) the sweep's per-wave partials, a resident
// device value, or the host.
__global__ __launch_bounds__(T) void k_maximizers_f(
    const double* Q, const uint8_t* S, int64_t N, int G, double max_l,
    const double* l0_part, int n_l0, const double* max_l_dev, uint8_t* M,
    double* wpart, double* max_l_out, double* max_l_out2) {
  __shared__ double sh[T / 64];
  if (l0_part) {
    double v = -INFINITY;
    for (int e = threadIdx.x; e < n_l0; e += T) v = fmax(v, l0_part[e]);
    max_l = block_max(v, sh);
  } else if (max_l_dev) {
    max_l = max_l_dev[0];
  }
  double v = -INFINITY;
  // four rows per thread and trip, all loads issued before the first use (the
  // pass is a chain of exposed memory latencies otherwise)
  const int64_t stride = int64_t(gridDim.x) * T;
  for (int64_t i0 = int64_t(blockIdx.x) * T + threadIdx.x; i0 < N; i0 += 4 * stride) {
    double2_t q[4];
    uint8_t s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = i0 + r * stride, ii = i < N ? i : i0;
      q[r] = *reinterpret_cast<const double2_t*>(Q + ii * G * 2);
      s[r] = S[ii];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = i0 + r * stride;
      if (i >= N) break;
      const bool m = s[r] && (q[r].y >= max_l);
      M[i] = m ? 1 : 0;
      if (m) v = fmax(v, q[r].y - q[r].x);
    }
  }
  const double mx = block_max(v, sh);
  if (threadIdx.x == 0) {
    wpart[blockIdx.x] = mx;
    if (blockIdx.x == 0) {
      if (max_l_out) *max_l_out = max_l;
      if (max_l_out2) *max_l_out2 = max_l;
    }
  }
}

// Candidate mask, widths, G = 0, per-block (candidates, unsafe) counts and the
// block's FIRST candidate in visiting order (largest width, ties: largest index).
__global__ __launch_bounds__(T) void k_candidates_f(
    const double* Q, const uint8_t* S, const uint8_t* M, int64_t N, int G,
    const double* wpart, int nwpart, Vec8 scaling, Vec8 thr_beta, int64_t goff,
    uint8_t* cand, double* w, uint8_t* Gm, unsigned* block_counts, double* best_w,
    int64_t* best_i, unsigned* best_ties, double* max_width_out) {
  __shared__ double sh[T / 64];
  __shared__ Pair shp[T / 64];
  __shared__ unsigned shc[2 * (T / 64)];
  double mw = -INFINITY;
  for (int e = threadIdx.x; e < nwpart; e += T) mw = fmax(mw, wpart[e]);
  mw = block_max(mw, sh);
  const double max_var = mw / scaling.v[0];
  unsigned nc = 0, nu = 0, nt = 0;
  Pair best{-INFINITY, -1};
  // four rows per thread and trip, the loads of the four rows in flight together;
  // nt = how many of this thread's candidates share the width of its best one
  const int64_t stride = int64_t(gridDim.x) * T;
  for (int64_t i0 = int64_t(blockIdx.x) * T + threadIdx.x; i0 < N; i0 += 4 * stride) {
    int64_t ii[4];
    uint8_t sv[4], mv[4];
    double wmax[4], smax[4];
    bool above[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ii[r] = (i0 + r * stride < N) ? i0 + r * stride : i0;
      sv[r] = S[ii[r]];
      mv[r] = M[ii[r]];
      wmax[r] = smax[r] = -INFINITY;
      above[r] = false;
    }
    for (int g = 0; g < G; ++g) {
      double2_t q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        q[r] = *reinterpret_cast<const double2_t*>(Q + (ii[r] * G + g) * 2);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double width = q[r].y - q[r].x;
        wmax[r] = fmax(wmax[r], width);
        smax[r] = fmax(smax[r], width / scaling.v[g]);
        above[r] = above[r] || (width > thr_beta.v[g]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = i0 + r * stride;
      if (i >= N) break;
      const bool sf = sv[r] != 0;
      const bool c = sf && !mv[r] && (smax[r] > max_var) && above[r];
      if (!sf) ++nu;
      cand[i] = c ? 1 : 0;
      w[i] = sf ? wmax[r] : -INFINITY;
      Gm[i] = 0;
      if (c) {
        ++nc;
        const Pair p{wmax[r], goff + i};
        if (best.i >= 0 && p.v == best.v) ++nt;
        if (best.i < 0 || p.v > best.v) nt = 1;
        if (best.i < 0 || before_desc(p, best)) best = p;
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    nc += __shfl_xor(nc, o, 64);
    nu += __shfl_xor(nu, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    shc[2 * wave] = nc;
    shc[2 * wave + 1] = nu;
  }
  const Pair win = block_best<false>(best, shp);     // (syncs)
  if (threadIdx.x < 2) {
    unsigned t = 0;
    for (int wv = 0; wv < T / 64; ++wv) t += shc[2 * wv + threadIdx.x];
    block_counts[2 * blockIdx.x + threadIdx.x] = t;
  }
  // how many of this block's candidates share the width of its first one (exact
  // ties decide the reference's visiting order, gp_opt.py:542-552): the threads
  // whose own best has that width counted theirs on the way
  if (!(win.i >= 0 && best.i >= 0 && best.v == win.v)) nt = 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nt += __shfl_xor(nt, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) shc[wave] = nt;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int wv = 0; wv < T / 64; ++wv) t += shc[wv];
    best_ties[blockIdx.x] = t;
    best_w[blockIdx.x] = win.v;
    best_i[blockIdx.x] = win.i;
    if (blockIdx.x == 0 && max_width_out) *max_width_out = mw;
  }
}

// One workgroup: total counts, the first candidate of the whole shard, its row
// (x, mean, Q) into the result block AND as the operand of the expander test
// (xc, resid[g * 16] = u_g - mu_g), flags / list counter zeroed: front_final_fold
// (sets_front.h) as a launch of its own -- N ranks, or no GP with a constraint; the
// one-rank chain runs it at the top of k_expkt instead.
__global__ __launch_bounds__(T) void k_front_final(FrontArgs a) {
  front_final_fold(a, true);
}

// M | G arg-max of max_i (u_i - l_i) / scaling_i with the conditional G mark of
// the fused path folded in: when a first candidate was found and every active GP
// certified it, the workgroup that owns its row sets G there and counts it.
__global__ __launch_bounds__(T) void k_argmax_marked(
    const double* Q, const uint8_t* M, uint8_t* Gm, int64_t N, int G,
    int64_t goff, Vec8 scaling, Vec8 fmin, const int32_t* flags,
    const int64_t* cand_gidx, const int* nfound, double* out_v, int64_t* out_i,
    int32_t* flags_out) {
  __shared__ Pair sh[T / 64];
  // (one-rank chain: the partial results go to mapped host memory, where the host takes
  // the last level -- no final launch, no read-back copy -- and the flags with them)
  if (flags_out && blockIdx.x == 0 && threadIdx.x < G) flags_out[threadIdx.x] = flags[threadIdx.x];
  int64_t lmark = -1;
  if (*nfound > 0) {
    bool ok = true, any = false;
    for (int g = 0; g < G; ++g) {
      if (fmin.v[g] == -INFINITY) continue;
      any = true;
      ok = ok && (flags[g] != 0);
    }
    if (ok && any) lmark = cand_gidx[0] - goff;
  }
  Pair best{-INFINITY, -1};
  const int64_t begin = int64_t(blockIdx.x) * (T * 4);
  for (int r = 0; r < 4; ++r) {
    const int64_t i = begin + r * T + threadIdx.x;
    if (i >= N) continue;
    const bool marked = i == lmark;
    if (marked) Gm[i] = 1;
    if (!(M[i] || Gm[i] || marked)) continue;
    double v = -INFINITY;
    for (int g = 0; g < G; ++g)
      v = fmax(v, (Q[(i * G + g) * 2 + 1] - Q[(i * G + g) * 2]) / scaling.v[g]);
    const Pair p{v, goff + i};
    if (before_first(p, best)) best = p;
  }
  const Pair win = block_best<true>(best, sh);
  if (threadIdx.x == 0) {
    out_v[blockIdx.x] = win.v;
    out_i[blockIdx.x] = win.i;
  }
}

// ... final level, and the expander flags into the result block
__global__ __launch_bounds__(T) void k_argmax_final_f(
    const double* in_v, const int64_t* in_i, int64_t n, const int32_t* flags,
    int G, int32_t* flags_out, double* out_v, int64_t* out_i) {
  __shared__ Pair sh[T / 64];
  Pair best{-INFINITY, -1};
  for (int64_t e = threadIdx.x; e < n; e += T) {
    const Pair p{in_v[e], in_i[e]};
    if (before_first(p, best)) best = p;
  }
  const Pair win = block_best<true>(best, sh);
  if (threadIdx.x == 0) {
    out_v[0] = win.v;
    out_i[0] = win.i;
  }
  if (threadIdx.x < G) flags_out[threadIdx.x] = flags[threadIdx.x];
}

// ---- N ranks: merges behind in-stream all-gathers ------------------------------------
// Every rank's front block ([0] max width | [1..2] counts (u64) | [3] w_top | [4]
// idx_top (i64) | [5] n_found, n_tied (int) | x[d] | mean[G] | q[2G]) gathered into
// `all` [world][nfront]: the first candidate of the WHOLE grid in visiting order
// (gp_opt.py:542-552: width descending, among equal widths the larger index first),
// total counts, the number of candidates tied with it over all shards -> `res` in the
// same layout, and the candidate staged as the operand of the expander test exactly
// as k_front_final stages it (xc, resid[g * 16] = u_g - mu_g; flags zeroed).  One
// wave; every rank computes the same block.
__global__ __launch_bounds__(64) void k_merge_front(
    const double* all, int world, int nfront, int d, int G, double* res, double* xc,
    int n_xc_resid, int32_t* flags, int n_flag_words) {
  const int lane = threadIdx.x;
  for (int e = lane; e < n_xc_resid; e += 64) xc[e] = 0.0;
  for (int e = lane; e < n_flag_words; e += 64) flags[e] = 0;
  __shared__ int win_rank;
  if (lane == 0) {
    unsigned long long ta = 0, tb = 0;
    Pair best{-INFINITY, -1};
    int wr = -1;
    for (int r = 0; r < world; ++r) {
      const double* b = all + size_t(r) * nfront;
      ta += reinterpret_cast<const unsigned long long*>(b)[1];
      tb += reinterpret_cast<const unsigned long long*>(b)[2];
      const int found = reinterpret_cast<const int*>(b + 5)[0];
      const Pair p{b[3], reinterpret_cast<const int64_t*>(b)[4]};
      if (found > 0 && p.i >= 0 && (best.i < 0 || before_desc(p, best))) {
        best = p;
        wr = r;
      }
    }
    int ties = 0;
    for (int r = 0; r < world; ++r) {
      const double* b = all + size_t(r) * nfront;
      if (reinterpret_cast<const int*>(b + 5)[0] > 0 && b[3] == best.v)
        ties += reinterpret_cast<const int*>(b + 5)[1];
    }
    res[0] = all[0];                      // (all-reduced before: the same everywhere)
    reinterpret_cast<unsigned long long*>(res)[1] = ta;
    reinterpret_cast<unsigned long long*>(res)[2] = tb;
    res[3] = best.v;
    reinterpret_cast<int64_t*>(res)[4] = best.i;
    reinterpret_cast<int*>(res + 5)[0] = wr >= 0 ? 1 : 0;
    reinterpret_cast<int*>(res + 5)[1] = wr >= 0 ? ties : 0;
    win_rank = wr;
  }
  __syncthreads();
  const int wr = win_rank;
  if (wr < 0) return;
  const double* b = all + size_t(wr) * nfront;
  double* resid = xc + (n_xc_resid - G * 16);
  for (int k = lane; k < d; k += 64) {
    res[6 + k] = b[6 + k];
    xc[k] = b[6 + k];
  }
  for (int g = lane; g < G; g += 64) {
    const double mu = b[6 + d + g];
    res[6 + d + g] = mu;
    resid[g * 16] = b[6 + d + G + 2 * g + 1] - mu;
  }
  for (int q = lane; q < 2 * G; q += 64) res[6 + d + G + q] = b[6 + d + G + q];
}

// Every rank's (value, global index) of the M | G arg-max gathered into `all`
// [world][2]: np.argmax over the whole grid (larger value first, among equals the
// smaller index) -- gp_opt.py:635, 642-644.
__global__ void k_merge_argmax(const double* all, int world, double* out_v,
                               int64_t* out_i) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Pair best{-INFINITY, -1};
  for (int r = 0; r < world; ++r) {
    const Pair p{all[2 * r], reinterpret_cast<const int64_t*>(all)[2 * r + 1]};
    if (before_first(p, best)) best = p;
  }
  *out_v = best.v;
  *out_i = best.i;
}

inline unsigned nblk(int64_t N, int per) { return unsigned((N + per - 1) / per); }

}  // namespace

int launch_reduce_max(sgp_ctx* ctx, const double* in, int64_t n, double* out) {
  hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(1024), 0, ctx->stream, in, n,
                     out);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_safe_set(sgp_grid* g, const double* fmin) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_safe_set, dim3(nblk(g->N, T)), dim3(T), 0, ctx->stream,
                     g->Q, g->N, g->G, vec8(fmin, g->G, -INFINITY), g->S,
                     g->partial);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_maximizers(sgp_grid* g, double max_l, const double* max_l_dev) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_maximizers, dim3(nblk(g->N, T)), dim3(T), 0, ctx->stream,
                     g->Q, g->S, g->N, g->G, max_l, max_l_dev, g->M, g->partial);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_candidates(sgp_grid* g, double max_var, const double* max_width_dev,
                      const double* scaling, const double* thr_beta,
                      int full_sets, unsigned long long* counts) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = nblk(g->N, T);
  unsigned* bc = static_cast<unsigned*>(
      sgp_scratch(ctx, 2, size_t(nb) * 2 * sizeof(unsigned)));
  if (!counts || !bc) return -1;
  hipLaunchKernelGGL(k_candidates, dim3(nb), dim3(T), 0, ctx->stream, g->Q,
                     g->S, g->M, g->N, g->G, max_var, max_width_dev,
                     vec8(scaling, g->G, 1.0), vec8(thr_beta, g->G, 0.0),
                     full_sets, g->cand, g->w, g->Gm, bc);
  hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(1024), 0, ctx->stream, bc,
                     int64_t(nb), counts);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_topk(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, int k,
                double* w_out_dev, int64_t* idx_out_dev, int* n_out_dev) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = nblk(g->N, TK_CHUNK);
  double* pw = static_cast<double*>(
      sgp_scratch(ctx, 2, size_t(nb) * k * (sizeof(double) + sizeof(int64_t))));
  if (!pw) return -1;
  int64_t* pi = reinterpret_cast<int64_t*>(pw + size_t(nb) * k);
  hipLaunchKernelGGL(k_topk, dim3(nb), dim3(T), 0, ctx->stream, g->cand, g->w,
                     static_cast<const int64_t*>(nullptr), g->N, g->goff, mode,
                     cut_w, cut_idx, k, pw, pi, static_cast<int*>(nullptr));
  hipLaunchKernelGGL(k_topk, dim3(1), dim3(T), 0, ctx->stream,
                     static_cast<const uint8_t*>(nullptr), pw, pi,
                     int64_t(nb) * k, int64_t(0), 0, INFINITY,
                     INT64_MAX, k, w_out_dev, idx_out_dev, n_out_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// Number of candidates whose width equals the first candidate's, bit for bit
// (gp_opt.py:542-552 visits the candidates by argsort()[::-1]: among exactly tied
// widths NumPy's sort decides, and the host has to settle the order only when
// this count exceeds one).  *w_top / *n_found are what launch_topk(k = 1) left;
// the count goes to ties[0] (zeroed here).
__global__ __launch_bounds__(256) void k_count_ties(const uint8_t* cand, const double* w,
                                                    int64_t N, const double* w_top,
                                                    const int* n_found, int* ties) {
  if (*n_found <= 0) return;
  const double wt = *w_top;
  int mine = 0;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < N;
       i += int64_t(gridDim.x) * blockDim.x)
    mine += (cand[i] && w[i] == wt) ? 1 : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(ties, mine);
}

int launch_count_ties(sgp_grid* g, const double* w_top_dev, const int* n_found_dev,
                      int* ties_dev) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipMemsetAsync(ties_dev, 0, sizeof(int), ctx->stream));
  const unsigned nb = unsigned(std::min<int64_t>(1024, (g->N + 1023) / 1024));
  hipLaunchKernelGGL(k_count_ties, dim3(nb ? nb : 1), dim3(256), 0, ctx->stream, g->cand,
                     g->w, g->N, w_top_dev, n_found_dev, ties_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_lipschitz(sgp_grid* g, int G, const double* fmin,
                     const double* lipschitz, int m, const double* xc_dev,
                     const double* uc_dev, int32_t* flags_dev) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_lipschitz, dim3(nblk(g->N, T)), dim3(T), 0, ctx->stream,
                     g->pts, g->S, g->N, g->d, G, vec8(fmin, G, -INFINITY),
                     vec8(lipschitz, G, 0.0), m, xc_dev, uc_dev, flags_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_argmax(sgp_grid* g, int mode, const double* scaling,
                  double* value_dev, int64_t* idx_dev) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = nblk(g->N, T * 4);
  double* pv = static_cast<double*>(
      sgp_scratch(ctx, 2, size_t(nb) * (sizeof(double) + sizeof(int64_t))));
  if (!pv) return -1;
  int64_t* pi = reinterpret_cast<int64_t*>(pv + nb);
  hipLaunchKernelGGL(k_argmax, dim3(nb), dim3(T), 0, ctx->stream, g->Q, g->S,
                     g->M, g->Gm, g->N, g->G, g->goff, mode,
                     vec8(scaling, g->G, 1.0), pv, pi);
  hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(T), 0, ctx->stream, pv, pi,
                     int64_t(nb), value_dev, idx_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_fill_cols(sgp_grid* g, const double* c, int nc) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_fill_cols, dim3(nblk(g->N, T)), dim3(T), 0, ctx->stream,
                     g->pts, g->N, g->d, nc, vec8(c, nc, 0.0));
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_gather_rows(sgp_grid* g, const int64_t* lidx_dev, int m, double* x,
                       double* mean, double* var, double* Q) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_gather_rows, dim3(m), dim3(64), 0, ctx->stream, g->pts,
                     g->mean, g->var, g->Q, g->N, g->d, g->G, lidx_dev, m, x,
                     mean, var, Q);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_gather_top(sgp_grid* g, const int64_t* gidx_dev, double* x,
                      double* mean, double* Q) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_gather_top, dim3(1), dim3(64), 0, ctx->stream, g->pts,
                     g->mean, g->Q, g->N, g->d, g->G, gidx_dev, g->goff, x,
                     mean, Q);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_mark_if(sgp_grid* g, int64_t li, const int32_t* flags_dev,
                   const double* fmin) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_mark_if, dim3(1), dim3(64), 0, ctx->stream, g->Gm, li,
                     flags_dev, g->G, vec8(fmin, g->G, -INFINITY));
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_stage_top(sgp_grid* g, const double* x_top, const double* mean_top,
                     const double* q_top, double* xc, double* resid) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_stage_top, dim3(1), dim3(64), 0, ctx->stream, x_top,
                     mean_top, q_top, g->d, g->G, xc, resid);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_stage_batch(sgp_grid* g, const int64_t* gidx_dev, const int* nfound_dev, int K,
                       double* xc, int n_xc_resid, int32_t* flags, int n_flag_words) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_stage_batch, dim3(1), dim3(256), 0, ctx->stream, gidx_dev, nfound_dev, K,
                     g->pts, g->mean, g->Q, g->N, g->d, g->G, g->goff, xc, n_xc_resid, flags,
                     n_flag_words);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_mark_top_if(sgp_grid* g, const int64_t* gidx_dev, const int* nfound_dev,
                       const int32_t* flags_dev, const double* fmin) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_mark_top_if, dim3(1), dim3(64), 0, ctx->stream, g->Gm,
                     gidx_dev, nfound_dev, g->goff, flags_dev, g->G,
                     vec8(fmin, g->G, -INFINITY));
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// ---- a pass of the expander loop over MANY candidates (sgp_grid_expander_pass) ---------
// The reference walks the candidates by width, widest first, until one certifies
// (gp_opt.py:542-557, 611-612) -- every one of them when none does, and in plotting mode
// (full_sets, :553-555) in any case.  A pass takes the next ~`want` candidates behind the cut
// WITHOUT sorting them: a histogram of their keys (width; -index in full_sets mode) gives the
// key `thr` above which about `want` of them lie, those are listed in whatever order the
// workgroups get to them, all of them are tested (k_expander_many), and the first expander in
// VISITING order is the listed hit with the largest (key, index) -- everything in front of it
// was tested in this pass or an earlier one.
namespace {
constexpr int kPassBins = 4096;

struct PassSel {
  double thr;      // the pass = candidates behind the cut with key >= thr
  int count;       // ... as k_pass_list counted them
  int est;         // ... as the histogram promised
};

__device__ __forceinline__ bool pass_key(const uint8_t* cand, const double* w, int64_t e,
                                         int64_t goff, int index_key, double cut_w,
                                         int64_t cut_idx, double* key) {
  if (!cand[e]) return false;
  const int64_t gi = goff + e;
  const double k = index_key ? -double(gi) : w[e];
  *key = k;
  return k < cut_w || (k == cut_w && gi < cut_idx);      // strictly behind the cut
}

__global__ __launch_bounds__(T) void k_pass_hist(const uint8_t* cand, const double* w,
                                                 int64_t N, int64_t goff, int index_key,
                                                 double cut_w, int64_t cut_idx, double lo,
                                                 double hi, unsigned* hist) {
  __shared__ unsigned sh[kPassBins];
  for (int b = threadIdx.x; b < kPassBins; b += T) sh[b] = 0;
  __syncthreads();
  const double scale = double(kPassBins) / (hi - lo);
  for (int64_t e = int64_t(blockIdx.x) * T + threadIdx.x; e < N; e += int64_t(gridDim.x) * T) {
    double key;
    if (!pass_key(cand, w, e, goff, index_key, cut_w, cut_idx, &key)) continue;
    int b = int((key - lo) * scale);
    b = b < 0 ? 0 : (b >= kPassBins ? kPassBins - 1 : b);
    atomicAdd(&sh[b], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kPassBins; b += T)
    if (sh[b]) atomicAdd(&hist[b], sh[b]);
}

// thr = lower edge of the highest bin at which the count from the top reaches `want`
// (-inf: fewer than that are left, the pass takes them all)
__global__ __launch_bounds__(1024) void k_pass_pick(const unsigned* hist, int want, double lo,
                                                    double hi, PassSel* sel) {
  __shared__ unsigned part[1024];
  constexpr int kPer = kPassBins / 1024;
  const int t = threadIdx.x;
  // thread t: the bins kPassBins - kPer (t + 1) .. kPassBins - kPer t - 1 (from the top)
  unsigned mine = 0;
  for (int q = 0; q < kPer; ++q) mine += hist[kPassBins - 1 - (kPer * t + q)];
  part[t] = mine;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {           // inclusive scan
    const unsigned v = (t >= o) ? part[t - o] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const unsigned incl = part[t], excl = incl - mine;
  if (t == 1023 && incl < unsigned(want)) {
    sel->thr = -INFINITY;
    sel->est = int(incl);
    sel->count = 0;
  }
  if (incl >= unsigned(want) && excl < unsigned(want)) {
    unsigned c = excl;
    for (int q = 0; q < kPer; ++q) {
      const int b = kPassBins - 1 - (kPer * t + q);
      c += hist[b];
      if (c >= unsigned(want)) {
        sel->thr = b == 0 ? -INFINITY : lo + (hi - lo) * (double(b) / double(kPassBins));
        sel->est = int(c);
        sel->count = 0;
        break;
      }
    }
  }
}

// The list is in ROW ORDER (a count per 256-row chunk, a prefix sum over the chunks, then the
// rows behind their chunk's offset): a group of 16 consecutive list entries -- the unit of the
// test, k_expander_many -- is a run of neighbours along a grid line, and its bounding box is
// what lets whole (rows x group) blocks skip the pre-filter.  Deterministic, too.
__device__ __forceinline__ bool pass_take(const uint8_t* cand, const double* w, int64_t e,
                                          int64_t N, int64_t goff, int index_key, double cut_w,
                                          int64_t cut_idx, double thr) {
  double key = 0.0;
  return e < N && pass_key(cand, w, e, goff, index_key, cut_w, cut_idx, &key) && key >= thr;
}

__global__ __launch_bounds__(T) void k_pass_count(const uint8_t* cand, const double* w,
                                                  int64_t N, int64_t goff, int index_key,
                                                  double cut_w, int64_t cut_idx,
                                                  const PassSel* sel, int* counts) {
  __shared__ int wc[T / 64];
  const double thr = sel->thr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t e0 = int64_t(blockIdx.x) * T; e0 < N; e0 += int64_t(gridDim.x) * T) {
    const bool take = pass_take(cand, w, e0 + threadIdx.x, N, goff, index_key, cut_w, cut_idx, thr);
    const unsigned long long b = __ballot(take);
    if (lane == 0) wc[wave] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
      int n = 0;
      for (int q = 0; q < T / 64; ++q) n += wc[q];
      counts[e0 / T] = n;
    }
    __syncthreads();
  }
}

// exclusive prefix sum of the chunk counts (in place), the total into sel->count
__global__ __launch_bounds__(1024) void k_pass_scan(int* counts, int nchunks, PassSel* sel) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nchunks; base += 1024) {
    const int i = base + t;
    const int v = i < nchunks ? counts[i] : 0;
    part[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int u = (t >= o) ? part[t - o] : 0;
      __syncthreads();
      part[t] += u;
      __syncthreads();
    }
    if (i < nchunks) counts[i] = carry + part[t] - v;
    __syncthreads();
    if (t == 1023) carry += part[1023];
    __syncthreads();
  }
  if (t == 0) sel->count = carry;
}

__global__ __launch_bounds__(T) void k_pass_list(const uint8_t* cand, const double* w,
                                                 int64_t N, int64_t goff, int index_key,
                                                 double cut_w, int64_t cut_idx,
                                                 const PassSel* sel, const int* offsets,
                                                 int* list) {
  __shared__ int wc[T / 64];
  const double thr = sel->thr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t e0 = int64_t(blockIdx.x) * T; e0 < N; e0 += int64_t(gridDim.x) * T) {
    const int64_t e = e0 + threadIdx.x;
    const bool take = pass_take(cand, w, e, N, goff, index_key, cut_w, cut_idx, thr);
    const unsigned long long b = __ballot(take);
    if (lane == 0) wc[wave] = __popcll(b);
    __syncthreads();
    int at = offsets[e0 / T];
    for (int q = 0; q < wave; ++q) at += wc[q];
    if (take) list[at + __popcll(b & ((1ull << lane) - 1ull))] = int(e);
    __syncthreads();
  }
}

// operand block of the listed candidates: xc[pos][d] = the row, resid[group][g][c] = u_g -
// mu_g (groups of 16 in list order; the tail of the last group stays zero)
__global__ __launch_bounds__(T) void k_pass_stage(const int* list, int count, const double* pts,
                                                  const double* mean, const double* Q, int64_t N,
                                                  int d, int G, double* xc, double* resid) {
  const int pos = blockIdx.x * T + threadIdx.x;
  if (pos >= count) return;
  const int64_t li = list[pos];
  for (int k = 0; k < d; ++k) xc[int64_t(pos) * d + k] = pts[int64_t(k) * N + li];
  for (int g = 0; g < G; ++g)
    resid[(int64_t(pos >> 4) * G + g) * 16 + (pos & 15)] =
        Q[li * 2 * G + 2 * g + 1] - mean[int64_t(g) * N + li];
}

// ---- the Lipschitz test of MANY candidates (sgp_grid_lipschitz_pass) ---------------------
// gp_opt.py:558-576 for every candidate of a pass: candidate c is an expander when for every
// GP i with a constraint SOME unsafe row x has u_i(x_c) - L_i |x_c - x| >= fmin_i -- the
// comparison is monotone in the distance, so that is: the unsafe row NEAREST to x_c passes
// for every such GP, and one flag per candidate is enough.  The pairs are pruned by boxes:
// per group of 16 listed candidates (neighbours along a grid line) the bounding box and the
// largest radius rho_c = min_i (u_i - fmin_i) / L_i (+ room for the rounding of the comparison)
// any of them has; a 16-row segment of the grid whose box is further than that from the
// group's box has no pair to test.  What is left gets the arithmetic of k_lipschitz.
__global__ __launch_bounds__(T) void k_lip_stage(const int* list, int count, const double* pts,
                                                 const double* Q, int64_t N, int d, int G,
                                                 double* xc, double* uc) {
  const int pos = blockIdx.x * T + threadIdx.x;
  if (pos >= count) return;
  const int64_t li = list[pos];
  for (int k = 0; k < d; ++k) xc[int64_t(pos) * d + k] = pts[int64_t(k) * N + li];
  for (int g = 0; g < G; ++g) uc[int64_t(pos) * G + g] = Q[li * 2 * G + 2 * g + 1];
}

__global__ __launch_bounds__(T) void k_lip_agg(int count, int d, int G, Vec8 fmin, Vec8 lips,
                                               const double* xc, const double* uc, double* box,
                                               double* rad, int ngroups) {
  const int z = blockIdx.x * T + threadIdx.x;
  if (z >= ngroups) return;
  const int m = min(16, count - 16 * z);
  double rmax = -INFINITY;
  for (int c = 0; c < m; ++c) {
    double rc = INFINITY;
    for (int g = 0; g < G; ++g) {
      if (fmin.v[g] == -INFINITY) continue;
      const double u = uc[(int64_t(z) * 16 + c) * G + g];
      const double room = (u - fmin.v[g]) + 1e-9 * (fabs(u) + fabs(fmin.v[g]) + 1.0);
      const double r = lips.v[g] > 0.0 ? room / lips.v[g] : (room >= 0.0 ? INFINITY : -INFINITY);
      rc = fmin2(rc, r);
    }
    rmax = fmax(rmax, rc);
  }
  rad[z] = rmax;
  for (int k = 0; k < d; ++k) {
    double lo = INFINITY, hi = -INFINITY;
    for (int c = 0; c < m; ++c) {
      const double v = xc[(int64_t(z) * 16 + c) * d + k];
      lo = fmin2(lo, v);
      hi = fmax(hi, v);
    }
    box[(int64_t(z) * 2 + 0) * d + k] = lo;
    box[(int64_t(z) * 2 + 1) * d + k] = hi;
  }
}

struct LipPass {
  const double* pts;
  const uint8_t* S;
  int64_t N;
  int d, G, m, ngroups;
  Vec8 fmin, lips;
  const double* xc;     // [m][d]
  const double* uc;     // [m][G]
  const double* box;    // [group][2][d]
  const double* rad;    // [group]
  int32_t* flags;       // [m][G]
  int* wcount;          // segments with a group in reach: count, list
  int* wlist;
};

// the wave's 16 rows (lane & 15), their box, and the block test of group zz
struct LipRows {
  double x[SGP_MAX_D], lo[SGP_MAX_D], hi[SGP_MAX_D];
  bool unsafe;
  __device__ __forceinline__ LipRows(const LipPass& a, int64_t wid, int lane) {
    const int64_t i = wid * 16 + (lane & 15);
    unsafe = (i < a.N) && (a.S[i] == 0);
    for (int k = 0; k < a.d; ++k) {
      x[k] = unsafe ? a.pts[int64_t(k) * a.N + i] : 0.0;
      double l = unsafe ? x[k] : INFINITY, h = unsafe ? x[k] : -INFINITY;
      for (int o = 8; o >= 1; o >>= 1) {
        l = fmin2(l, __shfl_xor(l, o, 64));
        h = fmax(h, __shfl_xor(h, o, 64));
      }
      lo[k] = l;
      hi[k] = h;
    }
  }
  __device__ __forceinline__ bool in_reach(const LipPass& a, int zz) const {
    const double r = a.rad[zz];
    if (!(r >= 0.0)) return false;
    const double* bx = a.box + int64_t(zz) * 2 * a.d;
    double d2 = 0.0;
    for (int k = 0; k < a.d; ++k) {
      const double gap = fmax(fmax(bx[k] - hi[k], lo[k] - bx[a.d + k]), 0.0);
      d2 += gap * gap;
    }
    return d2 <= r * r * (1.0 + 1e-9);        // (r = inf: in reach)
  }
};

// the scan of the grid: the segments with a group in reach (one group per lane)
__global__ __launch_bounds__(T) void k_lip_scan(LipPass a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = int64_t(blockIdx.x) * (T / 64) + (threadIdx.x >> 6);
  if (wid * 16 >= a.N) return;
  const LipRows rw(a, wid, lane);
  if (__ballot(rw.unsafe) == 0ull) return;
  bool some = false;
  for (int z0 = 0; z0 < a.ngroups && !some; z0 += 64) {
    const int zz = z0 + lane;
    some = __ballot(zz < a.ngroups && rw.in_reach(a, zz)) != 0ull;
  }
  if (some && lane == 0) a.wlist[atomicAdd(a.wcount, 1)] = int(wid);
}

// the listed segments, 64 groups at a time -- an item per wave of a fixed launch --: the pairs
// of the groups in reach with the arithmetic of k_lipschitz (lane = 16 (candidate % 4) + row,
// candidates (lane >> 4) + 4 r)
__global__ __launch_bounds__(T) void k_lip_items(LipPass a) {
  const int lane = threadIdx.x & 63;
  const int nch = (a.ngroups + 63) / 64;
  const int64_t total = int64_t(*a.wcount) * nch;
  for (int64_t item = int64_t(blockIdx.x) * (T / 64) + (threadIdx.x >> 6); item < total;
       item += int64_t(gridDim.x) * (T / 64)) {
    const int hw = int(item / nch), gc = int(item - int64_t(hw) * nch);
    const LipRows rw(a, a.wlist[hw], lane);
    const int zz = gc * 64 + lane;
    unsigned long long mask = __ballot(zz < a.ngroups && rw.in_reach(a, zz));
    while (mask != 0ull) {
      const int z = gc * 64 + __builtin_ctzll(mask);
      mask &= mask - 1ull;
      const int m = min(16, a.m - 16 * z);
      for (int r = 0; r < 4; ++r) {
        const int cand = (lane >> 4) + 4 * r;
        if (cand >= m || !rw.unsafe) continue;
        const int64_t c = int64_t(z) * 16 + cand;
        double s = 0.0;
        for (int k = 0; k < a.d; ++k) {
          const double df = a.xc[c * a.d + k] - rw.x[k];
          s += df * df;
        }
        const double dist = sqrt(s);
        bool hit = true;
        for (int g = 0; g < a.G; ++g) {
          if (a.fmin.v[g] == -INFINITY) continue;
          hit = hit && (a.uc[c * a.G + g] - a.lips.v[g] * dist >= a.fmin.v[g]);
        }
        if (hit && a.flags[c * a.G] == 0)
          for (int g = 0; g < a.G; ++g) atomicOr(&a.flags[c * a.G + g], 1);
      }
    }
  }
}

// N ranks: what the other ranks need of the listed candidates -- global row, key, the row
// itself, u_g - mu_g -- in list order.
__global__ __launch_bounds__(T) void k_pass_gather(const int* list, int count, const double* pts,
                                                   const double* mean, const double* Q,
                                                   const double* w, int64_t N, int d, int G,
                                                   int64_t goff, int index_key, int64_t* gidx,
                                                   double* key, double* x, double* resid) {
  const int pos = blockIdx.x * T + threadIdx.x;
  if (pos >= count) return;
  const int64_t li = list[pos];
  gidx[pos] = goff + li;
  key[pos] = (index_key & 1) ? -double(goff + li) : w[li];
  for (int k = 0; k < d; ++k) x[int64_t(pos) * d + k] = pts[int64_t(k) * N + li];
  for (int g = 0; g < G; ++g)
    resid[int64_t(pos) * G + g] =
        (index_key & 2) ? Q[li * 2 * G + 2 * g + 1]                       // (u_g itself)
                        : Q[li * 2 * G + 2 * g + 1] - mean[int64_t(g) * N + li];
}

// hits of the pass: a candidate is an expander when every GP with a constraint flagged it.
// mode 0: the first one in visiting order (largest key, then largest index); mode 1
// (full_sets): every one of them is marked in G.  res = { hits, key, index (i64 bits) }.
__global__ __launch_bounds__(1024) void k_pass_result(const int* list, int count,
                                                      const int32_t* flags, int G, Vec8 fmin,
                                                      const double* w, int64_t goff,
                                                      int index_key, uint8_t* Gm, double* res) {
  __shared__ Pair sh[1024 / 64];
  __shared__ unsigned shn[1024 / 64];
  Pair best{-INFINITY, -1};
  unsigned hits = 0;
  for (int pos = threadIdx.x; pos < count; pos += 1024) {
    bool ok = true;
    for (int g = 0; g < G; ++g)
      if (fmin.v[g] != -INFINITY && flags[int64_t(pos) * G + g] == 0) ok = false;
    if (!ok) continue;
    ++hits;
    const int64_t e = list[pos];
    if (index_key) {
      Gm[e] = 1;
    } else {
      const Pair p{w[e], goff + e};
      if (best.i < 0 || before_desc(p, best)) best = p;
    }
  }
  const Pair win = block_best<false>(best, sh);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) hits += __shfl_xor(hits, o, 64);
  if ((threadIdx.x & 63) == 0) shn[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (int wv = 0; wv < 1024 / 64; ++wv) tot += shn[wv];
    res[0] = double(tot);
    res[1] = win.v;
    memcpy(&res[2], &win.i, 8);
  }
}
}  // namespace

// count, scan, list behind the threshold in *sel (counts_dev: room for N / 256 + 1 ints)
static int pass_list_ordered(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, PassSel* sel,
                             int* list_dev, int* counts_dev) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nchunks = nblk(g->N, T);
  const unsigned nb = std::min<unsigned>(nchunks, 2048u);
  hipLaunchKernelGGL(k_pass_count, dim3(nb), dim3(T), 0, ctx->stream, g->cand, g->w, g->N,
                     g->goff, mode, cut_w, cut_idx, sel, counts_dev);
  hipLaunchKernelGGL(k_pass_scan, dim3(1), dim3(1024), 0, ctx->stream, counts_dev, int(nchunks),
                     sel);
  hipLaunchKernelGGL(k_pass_list, dim3(nb), dim3(T), 0, ctx->stream, g->cand, g->w, g->N,
                     g->goff, mode, cut_w, cut_idx, sel, counts_dev, list_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// Select the pass: list (device, room for N ints) and *sel (thr, count).
int launch_pass_select(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, double lo,
                       double hi, int want, void* sel_dev, int* list_dev, unsigned* hist_dev,
                       int* counts_dev) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = std::min<unsigned>(nblk(g->N, T), 2048u);
  SGP_HIP(ctx, hipMemsetAsync(hist_dev, 0, kPassBins * sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(k_pass_hist, dim3(nb), dim3(T), 0, ctx->stream, g->cand, g->w, g->N,
                     g->goff, mode, cut_w, cut_idx, lo, hi, hist_dev);
  hipLaunchKernelGGL(k_pass_pick, dim3(1), dim3(1024), 0, ctx->stream, hist_dev, want, lo, hi,
                     static_cast<PassSel*>(sel_dev));
  return pass_list_ordered(g, mode, cut_w, cut_idx, static_cast<PassSel*>(sel_dev), list_dev,
                           counts_dev);
}

// N ranks: the histogram alone (the ranks sum theirs and pick ONE threshold) ...
int launch_pass_hist(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, double lo, double hi,
                     unsigned* hist_dev) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = std::min<unsigned>(nblk(g->N, T), 2048u);
  SGP_HIP(ctx, hipMemsetAsync(hist_dev, 0, kPassBins * sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(k_pass_hist, dim3(nb), dim3(T), 0, ctx->stream, g->cand, g->w, g->N,
                     g->goff, mode, cut_w, cut_idx, lo, hi, hist_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// ... and the list behind a threshold the caller has put into *sel_dev ({ thr, 0, 0 })
int launch_pass_list(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, void* sel_dev,
                     int* list_dev, int* counts_dev) {
  return pass_list_ordered(g, mode, cut_w, cut_idx, static_cast<PassSel*>(sel_dev), list_dev,
                           counts_dev);
}

int launch_pass_gather(sgp_grid* g, const int* list_dev, int count, int mode, int64_t* gidx,
                       double* key, double* x, double* resid) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_pass_gather, dim3((count + T - 1) / T), dim3(T), 0, ctx->stream, list_dev,
                     count, g->pts, g->mean, g->Q, g->w, g->N, g->d, g->G, g->goff, mode, gidx,
                     key, x, resid);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_pass_stage(sgp_grid* g, const int* list_dev, int count, double* xc, double* resid) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_pass_stage, dim3((count + T - 1) / T), dim3(T), 0, ctx->stream, list_dev,
                     count, g->pts, g->mean, g->Q, g->N, g->d, g->G, xc, resid);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_pass_result(sgp_grid* g, const int* list_dev, int count, const int32_t* flags_dev,
                       const double* fmin, int mode, double* res_dev) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_pass_result, dim3(1), dim3(1024), 0, ctx->stream, list_dev, count,
                     flags_dev, g->G, vec8(fmin, g->G, -INFINITY), g->w, g->goff, mode, g->Gm,
                     res_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// the Lipschitz test of `count` listed candidates (list_dev: local rows): stage their rows and
// upper bounds (xc [count][d], uc [count][G]; box / rad per group behind them), test, flags
// [count][G] (zeroed here)
int launch_lipschitz_many(sgp_grid* g, int G, const double* fmin, const double* lipschitz,
                          const int* list_dev, int count, const double* xc_in, const double* uc_in,
                          double* work, int32_t* flags_dev) {
  sgp_ctx* ctx = g->ctx;
  const int d = g->d;
  const int ngroups = (count + 15) / 16;
  double* xc = work;
  double* uc = xc + size_t(count) * d;
  double* box = uc + size_t(count) * G;
  double* rad = box + size_t(ngroups) * 2 * d;
  if (list_dev) {
    hipLaunchKernelGGL(k_lip_stage, dim3((count + T - 1) / T), dim3(T), 0, ctx->stream, list_dev,
                       count, g->pts, g->Q, g->N, d, G, xc, uc);
  } else {
    SGP_HIP(ctx, hipMemcpyAsync(xc, xc_in, size_t(count) * d * 8, hipMemcpyHostToDevice, ctx->stream));
    SGP_HIP(ctx, hipMemcpyAsync(uc, uc_in, size_t(count) * G * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  const size_t nw = size_t((g->N + 15) >> 4);
  int* hot = static_cast<int*>(sgp_scratch(ctx, 12, (64 + nw) * sizeof(int)));
  SGP_CHECK(ctx, hot, "device allocation failed: %s", ctx->err.c_str());
  SGP_HIP(ctx, hipMemsetAsync(hot, 0, 64 * sizeof(int), ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(flags_dev, 0, size_t(count) * G * 4, ctx->stream));
  LipPass a{};
  a.pts = g->pts;
  a.S = g->S;
  a.N = g->N;
  a.d = d;
  a.G = G;
  a.m = count;
  a.ngroups = ngroups;
  a.fmin = vec8(fmin, G, -INFINITY);
  a.lips = vec8(lipschitz, G, 0.0);
  a.xc = xc;
  a.uc = uc;
  a.box = box;
  a.rad = rad;
  a.flags = flags_dev;
  a.wcount = hot;
  a.wlist = hot + 64;
  hipLaunchKernelGGL(k_lip_agg, dim3((ngroups + T - 1) / T), dim3(T), 0, ctx->stream, count, d, G,
                     a.fmin, a.lips, xc, uc, box, rad, ngroups);
  hipLaunchKernelGGL(k_lip_scan, dim3(unsigned((nw + T / 64 - 1) / (T / 64))), dim3(T), 0,
                     ctx->stream, a);
  hipLaunchKernelGGL(k_lip_items, dim3(1024), dim3(T), 0, ctx->stream, a);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_mark(sgp_grid* g, const int64_t* lidx_dev, int m, int value) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_mark, dim3((m + 63) / 64), dim3(64), 0, ctx->stream,
                     g->Gm, lidx_dev, m, value);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_import_points(sgp_ctx* ctx, const double* src, int64_t N, int d,
                         int64_t stride_row, int64_t stride_col, double* dst) {
  hipLaunchKernelGGL(k_import_points, dim3(nblk(N, T)), dim3(T), 0, ctx->stream,
                     src, N, d, stride_row, stride_col, dst);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}


// ---- launchers of the fused single-rank chain ---------------------------------------
static unsigned front_blocks(int64_t N) {
  const unsigned nb = nblk(N, T);
  return nb < unsigned(kFrontBlocks) ? nb : unsigned(kFrontBlocks);
}

int launch_sets_front_fused(sgp_grid* g, double max_l, const double* l0_part,
                            int n_l0, const double* max_l_dev,
                            const double* scaling, const double* thr_beta,
                            double* res, double* max_l_slot, double* xc,
                            int n_xc_resid, int32_t* flags, int n_flag_words,
                            FrontArgs* fold) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = front_blocks(g->N);
  // scratch: width partials | block counts | block bests
  char* sc = static_cast<char*>(sgp_scratch(
      ctx, 2, size_t(kFrontBlocks) * (8 + 8 + 8 + 8 + 8)));
  if (!sc) return -1;
  double* wpart = reinterpret_cast<double*>(sc);
  unsigned* bc = reinterpret_cast<unsigned*>(sc + size_t(kFrontBlocks) * 8);
  double* bw = reinterpret_cast<double*>(sc + size_t(kFrontBlocks) * 16);
  int64_t* bi = reinterpret_cast<int64_t*>(sc + size_t(kFrontBlocks) * 24);
  unsigned* bt = reinterpret_cast<unsigned*>(sc + size_t(kFrontBlocks) * 32);
  hipLaunchKernelGGL(k_maximizers_f, dim3(nb), dim3(T), 0, ctx->stream, g->Q,
                     g->S, g->N, g->G, max_l, l0_part, n_l0, max_l_dev, g->M,
                     wpart, g->scal, max_l_slot);
  hipLaunchKernelGGL(k_candidates_f, dim3(nb), dim3(T), 0, ctx->stream, g->Q,
                     g->S, g->M, g->N, g->G, wpart, int(nb),
                     vec8(scaling, g->G, 1.0), vec8(thr_beta, g->G, 0.0), g->goff,
                     g->cand, g->w, g->Gm, bc, bw, bi, bt, res);
  FrontArgs fa{};
  fa.block_counts = bc;
  fa.best_w = bw;
  fa.best_i = bi;
  fa.best_ties = bt;
  fa.nb = int(nb);
  fa.pts = g->pts;
  fa.mean = g->mean;
  fa.Q = g->Q;
  fa.N = g->N;
  fa.goff = g->goff;
  fa.d = g->d;
  fa.G = g->G;
  fa.res = res;
  fa.res_host = nullptr;
  fa.xc = xc;
  fa.n_xc_resid = n_xc_resid;
  fa.flags = flags;
  fa.n_flag_words = n_flag_words;
  if (fold)
    *fold = fa;      // (the caller's next kernel -- or launch_front_final -- takes it from here)
  else
    hipLaunchKernelGGL(k_front_final, dim3(1), dim3(T), 0, ctx->stream, fa);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_front_final(sgp_ctx* ctx, const FrontArgs& fa) {
  hipLaunchKernelGGL(k_front_final, dim3(1), dim3(T), 0, ctx->stream, fa);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int argmax_marked_blocks(int64_t N) { return int(nblk(N, T * 4)); }

// `host_part`: [nb] values | [nb] indices in mapped host memory and `flags_out` next to
// them -- the last level is the host's (sgp_grid_sets_fused); else a final launch leaves
// value / index / flags on the device.
int launch_argmax_marked(sgp_grid* g, const double* scaling, const double* fmin,
                         const int32_t* flags_dev, const int64_t* cand_gidx_dev,
                         const int* nfound_dev, int32_t* flags_out,
                         double* value_dev, int64_t* idx_dev, double* host_part) {
  sgp_ctx* ctx = g->ctx;
  const unsigned nb = nblk(g->N, T * 4);
  if (host_part) {
    hipLaunchKernelGGL(k_argmax_marked, dim3(nb), dim3(T), 0, ctx->stream, g->Q,
                       g->M, g->Gm, g->N, g->G, g->goff, vec8(scaling, g->G, 1.0),
                       vec8(fmin, g->G, -INFINITY), flags_dev, cand_gidx_dev,
                       nfound_dev, host_part, reinterpret_cast<int64_t*>(host_part + nb),
                       flags_out);
    SGP_HIP(ctx, hipGetLastError());
    return 0;
  }
  double* pv = static_cast<double*>(
      sgp_scratch(ctx, 2, size_t(nb) * (sizeof(double) + sizeof(int64_t))));
  if (!pv) return -1;
  int64_t* pi = reinterpret_cast<int64_t*>(pv + nb);
  hipLaunchKernelGGL(k_argmax_marked, dim3(nb), dim3(T), 0, ctx->stream, g->Q,
                     g->M, g->Gm, g->N, g->G, g->goff, vec8(scaling, g->G, 1.0),
                     vec8(fmin, g->G, -INFINITY), flags_dev, cand_gidx_dev,
                     nfound_dev, pv, pi, static_cast<int32_t*>(nullptr));
  hipLaunchKernelGGL(k_argmax_final_f, dim3(1), dim3(T), 0, ctx->stream, pv, pi,
                     int64_t(nb), flags_dev, g->G, flags_out, value_dev, idx_dev);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_merge_front(sgp_grid* g, const double* all, int world, int nfront, double* res,
                       double* xc, int n_xc_resid, int32_t* flags, int n_flag_words) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_merge_front, dim3(1), dim3(64), 0, ctx->stream, all, world, nfront,
                     g->d, g->G, res, xc, n_xc_resid, flags, n_flag_words);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_merge_argmax(sgp_ctx* ctx, const double* all, int world, double* out_v,
                        int64_t* out_i) {
  hipLaunchKernelGGL(k_merge_argmax, dim3(1), dim3(64), 0, ctx->stream, all, world, out_v,
                     out_i);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}
