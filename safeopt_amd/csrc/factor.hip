// GP "set_XY" on the device: kernel matrix, Cholesky, L^-1, alpha.
//
// Replaces what GPy's ExactGaussianInference does behind gp.set_XY
// (safeopt/gp_opt.py:227, 267, 275):  Ky = k(X,X) + (noise + 1e-8) I,
// L = jitchol(Ky), Ky^-1 via the triangular inverse, alpha = Ky^-1 y.
// n <= a few thousand, so n^3/3 <= ~3 GFLOP: a recursive blocked algorithm
// (32x32 leaves in LDS + one tiled fp64 GEMM kernel) is launch-bound, not
// flop-bound, and runs replicated on every GPU (no communication).
//
//   factor(A[0:s]):  s == 32 -> leaf (potf2 + triangular inverse in LDS)
//     else split h:  factor(A11); L21 = A21 Linv11^T; A22 -= L21 L21^T;
//                    factor(A22); Linv21 = -Linv22 (L21 Linv11)
#include <algorithm>
#include "small_path.h"

#include "kern_eval.h"
#include "sets_front.h"

namespace {

constexpr int NB = 32;  // leaf size

// ---- covariance matrix --------------------------------------------------------
// out[i*ld + j] = k(X1_i, X2_j) (+ diag_add on i == j when symmetric_diag).
// Rows/cols >= n_valid (padding of a square factorisation matrix) become the
// identity so the padded Cholesky stays well defined.
template <int D>
__global__ void k_kernel_matrix(KernDesc kd, const double* X1, int64_t n1,
                                const double* X2, int64_t n2, double* out,
                                int64_t ld, int symmetric_diag, double diag_add,
                                int64_t n_valid) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t i = int64_t(blockIdx.y) * blockDim.y + threadIdx.y;
  if (i >= n1 || j >= n2) return;
  double v;
  if (i >= n_valid || j >= n_valid) {
    v = (i == j) ? 1.0 : 0.0;
  } else {
    double a[D], b[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      a[k] = X1[i * D + k];
      b[k] = X2[j * D + k];
    }
    v = kern_eval<D>(kd, a, b);
    if (symmetric_diag && i == j) v += diag_add;
  }
  out[i * ld + j] = v;
}

// ---- leaf: 32x32 Cholesky + inverse of the factor, one workgroup --------------
// A (ld) holds the symmetric block (lower part used); on exit A's lower part is
// L, and Linv (ldi) receives L^-1 (lower).  info[0] = first bad pivot (1-based,
// offset by `row0`) if a pivot is not positive / not finite.
__global__ __launch_bounds__(256) void k_leaf(double* A, int64_t ld,
                                              double* Linv, int64_t ldi,
                                              int row0, int* info) {
  __shared__ double a[NB][NB + 1];
  __shared__ double inv[NB][NB + 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, j = e % NB;
    a[i][j] = (j <= i) ? A[i * ld + j] : 0.0;
    inv[i][j] = 0.0;
  }
  __syncthreads();
  for (int k = 0; k < NB; ++k) {
    const double piv = a[k][k];
    if (!(piv > 0.0) || !isfinite(piv)) {
      if (tid == 0 && bad == 0) bad = row0 + k + 1;
      __syncthreads();
      break;
    }
    const double dk = sqrt(piv);
    __syncthreads();
    if (tid == 0) a[k][k] = dk;
    if (tid > k && tid < NB) a[tid][k] /= dk;
    __syncthreads();
    // trailing update: a[i][j] -= a[i][k] a[j][k], k < j <= i
    for (int e = tid; e < NB * NB; e += 256) {
      const int i = e / NB, j = e % NB;
      if (j > k && j <= i) a[i][j] -= a[i][k] * a[j][k];
    }
    __syncthreads();
  }
  if (bad != 0) {
    if (tid == 0) atomicCAS(info, 0, bad);
    return;
  }
  // inverse by forward substitution, one column per thread
  if (tid < NB) {
    const int j = tid;
    for (int i = j; i < NB; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= a[i][k] * inv[k][j];
      inv[i][j] = s / a[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, j = e % NB;
    if (j <= i) {
      A[i * ld + j] = a[i][j];
      Linv[i * ldi + j] = inv[i][j];
    }
  }
}

// ---- tiled fp64 GEMM: C = alpha * A * op(B) + beta * C ------------------------
// Row-major, 64x64 tile per workgroup, 4x4 micro-tile per thread, K step 16.
// transB: op(B) = B^T with B given as (n x k).  lowerB: B (k x n) is lower
// triangular (entries with row < col are skipped = treated as 0); lowerA: same
// for A (m x k).
template <bool TRANSB>
__global__ __launch_bounds__(256) void k_gemm(int m, int n, int k, double alpha,
                                              const double* A, int64_t lda,
                                              const double* B, int64_t ldb,
                                              double beta, double* C,
                                              int64_t ldc) {
  __shared__ double As[16][64 + 1];  // As[kk][i]
  __shared__ double Bs[16][64 + 1];  // Bs[kk][j]
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  double acc[4][4] = {};
  for (int k0 = 0; k0 < k; k0 += 16) {
    for (int e = tid; e < 64 * 16; e += 256) {
      {  // A tile: rows i0..i0+63, cols k0..k0+15
        const int i = e >> 4, kk = e & 15;
        const int gi = i0 + i, gk = k0 + kk;
        As[kk][i] = (gi < m && gk < k) ? A[int64_t(gi) * lda + gk] : 0.0;
      }
      if (TRANSB) {  // B is (n x k): Bs[kk][j] = B[j0+j][k0+kk]
        const int j = e >> 4, kk = e & 15;
        const int gj = j0 + j, gk = k0 + kk;
        Bs[kk][j] = (gj < n && gk < k) ? B[int64_t(gj) * ldb + gk] : 0.0;
      } else {  // B is (k x n)
        const int kk = e >> 6, j = e & 63;
        const int gj = j0 + j, gk = k0 + kk;
        Bs[kk][j] = (gj < n && gk < k) ? B[int64_t(gk) * ldb + gj] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = As[kk][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = Bs[kk][tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = i0 + ty + 16 * r;
    if (gi >= m) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gj = j0 + tx + 16 * c;
      if (gj >= n) continue;
      double* p = C + int64_t(gi) * ldc + gj;
      const double old = (beta == 0.0) ? 0.0 : beta * (*p);
      *p = alpha * acc[r][c] + old;
    }
  }
}

int gemm(sgp_ctx* ctx, bool transB, int m, int n, int k, double alpha,
         const double* A, int64_t lda, const double* B, int64_t ldb,
         double beta, double* C, int64_t ldc) {
  if (m <= 0 || n <= 0) return 0;
  dim3 grid((n + 63) / 64, (m + 63) / 64);
  if (transB)
    hipLaunchKernelGGL(k_gemm<true>, grid, dim3(256), 0, ctx->stream, m, n, k,
                       alpha, A, lda, B, ldb, beta, C, ldc);
  else
    hipLaunchKernelGGL(k_gemm<false>, grid, dim3(256), 0, ctx->stream, m, n, k,
                       alpha, A, lda, B, ldb, beta, C, ldc);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// Recursive Cholesky + inverse on the n_f x n_f device matrices A (-> L) and
// Li (-> L^-1, must be zero-initialised); T is an n_f x n_f workspace.
int factor_rec(sgp_ctx* ctx, double* A, double* Li, double* T, int64_t ld,
               int off, int s, int* info_dev) {
  double* Ad = A + int64_t(off) * ld + off;
  double* Ld = Li + int64_t(off) * ld + off;
  if (s <= NB) {
    hipLaunchKernelGGL(k_leaf, dim3(1), dim3(256), 0, ctx->stream, Ad, ld, Ld,
                       ld, off, info_dev);
    SGP_HIP(ctx, hipGetLastError());
    return 0;
  }
  const int h = ((s / NB + 1) / 2) * NB;  // split at a leaf boundary
  const int r = s - h;
  SGP_TRY(factor_rec(ctx, A, Li, T, ld, off, h, info_dev));
  double* A21 = Ad + int64_t(h) * ld;
  double* A22 = Ad + int64_t(h) * ld + h;
  double* Li21 = Ld + int64_t(h) * ld;
  double* Li22 = Ld + int64_t(h) * ld + h;
  double* T21 = T + int64_t(off + h) * ld + off;
  // T21 = A21 * Linv11^T ; A21 <- T21  (L21)
  SGP_TRY(gemm(ctx, true, r, h, h, 1.0, A21, ld, Ld, ld, 0.0, T21, ld));
  SGP_HIP(ctx, hipMemcpy2DAsync(A21, ld * sizeof(double), T21,
                                ld * sizeof(double), h * sizeof(double), r,
                                hipMemcpyDeviceToDevice, ctx->stream));
  // A22 -= L21 L21^T
  SGP_TRY(gemm(ctx, true, r, r, h, -1.0, A21, ld, A21, ld, 1.0, A22, ld));
  SGP_TRY(factor_rec(ctx, A, Li, T, ld, off + h, r, info_dev));
  // Linv21 = -Linv22 * (L21 * Linv11)
  SGP_TRY(gemm(ctx, false, r, h, h, 1.0, A21, ld, Ld, ld, 0.0, T21, ld));
  SGP_TRY(gemm(ctx, false, r, h, r, -1.0, Li22, ld, T21, ld, 0.0, Li21, ld));
  return 0;
}

// ---- pack L^-1 into MFMA A-operand order ----------------------------------------
// Apack[(b * nsteps + s) * 64 + lane] = Linv[16 b + (lane & 15)][4 s + (lane >> 4)]
// (zero outside the n x n lower triangle): lane = 16 k + row, which is the A
// operand map of v_mfma_f64_16x16x4_f64 AND of v_mfma_f64_4x4x4_4b_f64 when its
// four blocks are four 4-row groups (A[blk][i][k] <- lane 16k + 4blk + i).
//
// Narrow last row block (n - 16 (nblk - 1) <= 4 real rows): its four 4-row
// groups all hold the SAME real rows 16 b + (lane & 3).  With
// v_mfma_f64_4x4x4_4b_f64 the four groups are the instruction's four blocks, so
// B may then hold a different point quad per block -- the plain covariance
// register, no broadcast -- and ONE instruction per k-step covers all 16 points
// of the wave (the padding rows would otherwise cost 3 of every 4 MFMAs of the
// row block that meets EVERY j-block).
__global__ void k_pack(const double* Li, int64_t ld, int n, int nblk,
                       int nsteps, int narrow, double* Apack) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = int64_t(nblk) * nsteps * 64;
  if (e >= total) return;
  const int lane = int(e & 63);
  const int64_t bs = e >> 6;
  const int s = int(bs % nsteps);
  const int b = int(bs / nsteps);
  const int i = 16 * b + ((narrow && b == nblk - 1) ? (lane & 3) : (lane & 15));
  const int j = 4 * s + (lane >> 4);
  double v = 0.0;
  if (i < n && j <= i) v = Li[int64_t(i) * ld + j];
  Apack[e] = v;
}

// Lower-triangular products with up to 16 right-hand sides, V and the results
// stored one vector per row (pitch ldv / ldo).  Both read L^-1 exactly once,
// coalesced along its rows, in a fixed summation order (every rank of a
// multi-GPU run computes bit-identical operands).
constexpr int kMaxRhs = 16;

// T[c][i] = sum_{j <= i} Li[i][j] V[c][j]: one wave per row i.
__global__ __launch_bounds__(256) void k_tri_mv(const double* Li, int64_t ld,
                                                int n, const double* V,
                                                int64_t ldv, int m, double* T,
                                                int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  double acc[kMaxRhs];
#pragma unroll
  for (int c = 0; c < kMaxRhs; ++c) acc[c] = 0.0;
  const double* row = Li + int64_t(i) * ld;
  for (int j = lane; j <= i; j += 64) {
    const double l = row[j];
#pragma unroll
    for (int c = 0; c < kMaxRhs; ++c)
      if (c < m) acc[c] = fma(l, V[c * ldv + j], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < kMaxRhs; ++c) {
    if (c < m) {
      double v = acc[c];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) T[c * ldo + i] = v;
    }
  }
}

// W[c][j] = sum_{i >= j} Li[i][j] T[c][i] for j < n, 0 for n <= j < n_out:
// one workgroup per 64 columns, its 8 waves take every 8th row.
__global__ __launch_bounds__(512) void k_tri_mtv(const double* Li, int64_t ld,
                                                 int n, const double* T,
                                                 int64_t ldt, int m, double* W,
                                                 int64_t ldo, int n_out) {
  __shared__ double sh[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * 64, j = j0 + lane;
  double acc[kMaxRhs];
#pragma unroll
  for (int c = 0; c < kMaxRhs; ++c) acc[c] = 0.0;
  if (j < n) {
    for (int i = j0 + wave; i < n; i += 8) {
      const double l = (i >= j) ? Li[int64_t(i) * ld + j] : 0.0;
#pragma unroll
      for (int c = 0; c < kMaxRhs; ++c)
        if (c < m) acc[c] = fma(l, T[c * ldt + i], acc[c]);
    }
  }
  for (int c = 0; c < m; ++c) {           // m is uniform: barriers are safe
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < kMaxRhs; ++k) v = (k == c) ? acc[k] : v;
    sh[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && j < n_out) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += sh[w][lane];
      W[c * ldo + j] = (j < n) ? tot : 0.0;
    }
    __syncthreads();
  }
}

int launch_tri_mv(sgp_ctx* ctx, const double* Li, int64_t ld, int n,
                  const double* V, int64_t ldv, int m, double* T, int64_t ldo) {
  hipLaunchKernelGGL(k_tri_mv, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, Li,
                     ld, n, V, ldv, m, T, ldo);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_tri_mtv(sgp_ctx* ctx, const double* Li, int64_t ld, int n,
                   const double* T, int64_t ldt, int m, double* W, int64_t ldo,
                   int n_out) {
  hipLaunchKernelGGL(k_tri_mtv, dim3((n_out + 63) / 64), dim3(512), 0,
                     ctx->stream, Li, ld, n, T, ldt, m, W, ldo, n_out);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// Xpad = zero-padded X; Xs = Xpad * scale0 per column (products of parts: copy);
// XA = the same rows and alpha interleaved per block of 16 training points:
// [16 d of Xs | 16 of alpha] (GpDev::XA)
__global__ void k_pad_rows(const double* X, const double* alpha, int n, int n_pad,
                           int d, KernDesc kd, double* Xpad, double* Xs,
                           double* XA) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_pad * d) return;
  const double v = (e < n * d) ? X[e] : 0.0;
  const double vs = (kd.n_parts == 1) ? v * kd.scale0[e % d] : v;
  Xpad[e] = v;
  Xs[e] = vs;
  const int row = e / d, col = e - row * d, jb = row >> 4, r = row & 15;
  double* blk = XA + size_t(jb) * (16 * d + 16);
  blk[r * d + col] = vs;
  if (col == 0) blk[16 * d + r] = alpha[row];     // zero beyond n (publish_gp)
}


}  // namespace

int launch_kernel_matrix(sgp_ctx* ctx, const KernDesc& kd, const double* X1,
                         int64_t n1, const double* X2, int64_t n2, double* out,
                         int64_t ld, int symmetric_diag, double diag_add,
                         int64_t n_valid) {
  if (n1 <= 0 || n2 <= 0) return 0;
  dim3 block(64, 4);
  dim3 grid(unsigned((n2 + 63) / 64), unsigned((n1 + 3) / 4));
#define KM_CASE(DD)                                                            \
  case DD:                                                                     \
    hipLaunchKernelGGL(k_kernel_matrix<DD>, grid, block, 0, ctx->stream, kd,   \
                       X1, n1, X2, n2, out, ld, symmetric_diag, diag_add,      \
                       n_valid);                                               \
    break;
  switch (kd.d) {
    KM_CASE(1) KM_CASE(2) KM_CASE(3) KM_CASE(4)
    KM_CASE(5) KM_CASE(6) KM_CASE(7) KM_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", kd.d, SGP_MAX_D);
      return -2;
  }
#undef KM_CASE
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// Everything the sweeps read, derived from the dense L^-1 (ld = gp->ld),
// gp->X and gp->alpha (first n entries valid): packed operand matrix, padded /
// pre-scaled inputs, zero-padded alpha, device descriptor.
int publish_gp(sgp_gp* gp) {
  sgp_ctx* ctx = gp->ctx;
  const int n = int(gp->n), np = gp->n_pad, d = gp->kern.d;
  double* Li = static_cast<double*>(gp->Linv.p);
  const int nblk = np / 16, nsteps = np / 4;
  const int64_t total = int64_t(nblk) * nsteps * 64;
  // (SGP_NO_NARROW=1: the A/B switch of profiles/)
  static const bool no_narrow = getenv("SGP_NO_NARROW") != nullptr;
  const int narrow = (!no_narrow && n - 16 * (nblk - 1) <= 4) ? 1 : 0;
  // (capacity for every n up to the pitch of L^-1: one-row appends then never
  // reallocate -- a reallocation is a hipMalloc and a stream sync)
  const size_t cap_rows = size_t(std::max(gp->ld, np));
  SGP_TRY(sgp_reserve(ctx, &gp->Apack,
                      (cap_rows / 16) * (cap_rows / 4) * 64 * sizeof(double)));
  hipLaunchKernelGGL(k_pack, dim3(unsigned((total + 255) / 256)), dim3(256), 0,
                     ctx->stream, Li, int64_t(gp->ld), n, nblk, nsteps, narrow,
                     static_cast<double*>(gp->Apack.p));
  SGP_TRY(sgp_reserve(ctx, &gp->Xpad, cap_rows * d * sizeof(double)));
  SGP_TRY(sgp_reserve(ctx, &gp->Xs, cap_rows * d * sizeof(double)));
  SGP_TRY(sgp_reserve(ctx, &gp->XA, (cap_rows / 16 + 1) * (16 * d + 16) * sizeof(double)));
  hipLaunchKernelGGL(k_pad_rows, dim3((np * d + 255) / 256), dim3(256), 0,
                     ctx->stream, static_cast<double*>(gp->X.p),
                     static_cast<double*>(gp->alpha.p), n, np, d,
                     gp->kern, static_cast<double*>(gp->Xpad.p),
                     static_cast<double*>(gp->Xs.p), static_cast<double*>(gp->XA.p));
  SGP_HIP(ctx, hipGetLastError());
  gp->dev.Apack = static_cast<double*>(gp->Apack.p);
  gp->dev.Xpad = static_cast<double*>(gp->Xpad.p);
  gp->dev.Xs = static_cast<double*>(gp->Xs.p);
  gp->dev.alpha = static_cast<double*>(gp->alpha.p);
  gp->dev.XA = static_cast<double*>(gp->XA.p);
  gp->dev.upd_w = static_cast<double*>(gp->updw.p);
  gp->dev.upd = static_cast<double*>(gp->upd.p);
  gp->dev.n = n;
  gp->dev.n_pad = np;
  gp->dev.nblk = nblk;
  gp->dev.narrow = narrow;
  gp->dev.last_rows = n - 16 * (nblk - 1);
  gp->dev.share = -1;       // (only collect_gps, which sees the other GPs, may set it)
  gp->dev.Linv = Li;
  gp->dev.ld = gp->ld;
  gp->dev.prior = gp->kern.kdiag + gp->noise_var + 1e-8 + gp->jitter;
  gp->dev.kern = gp->kern;
  return 0;
}

// Build Ky (with gp->jitter), factor, invert, pack, alpha.  *info = 0 or the
// 1-based index of the first non-positive pivot.  Buffers are sized for
// gp->ld >= n_f rows so later one-row appends need no reallocation.
int factor_gp(sgp_gp* gp, int* info) {
  sgp_ctx* ctx = gp->ctx;
  const int n = int(gp->n), nf = gp->n_f, np = gp->n_pad, ld = gp->ld;
  const size_t mat = size_t(ld) * ld * sizeof(double);
  SGP_TRY(sgp_reserve(ctx, &gp->Kmat, mat));
  SGP_TRY(sgp_reserve(ctx, &gp->Linv, mat));
  SGP_TRY(sgp_reserve(ctx, &gp->work, mat));
  SGP_TRY(sgp_reserve(ctx, &gp->tvec, size_t(ld) * sizeof(double) + 64));
  SGP_TRY(sgp_reserve(ctx, &gp->alpha, size_t(ld + 16) * sizeof(double)));
  SGP_TRY(sgp_reserve(ctx, &gp->updw, size_t(ld + 16) * sizeof(double)));
  SGP_TRY(sgp_reserve(ctx, &gp->upd, size_t(SGP_MAX_D + 2) * sizeof(double)));
  double* K = static_cast<double*>(gp->Kmat.p);
  double* Li = static_cast<double*>(gp->Linv.p);
  double* T = static_cast<double*>(gp->work.p);
  double* tv = static_cast<double*>(gp->tvec.p);
  int* info_dev = reinterpret_cast<int*>(tv + ld);
  gp->upd_valid = false;

  // padded rows of X are never read: n_valid = n turns them into identity
  SGP_TRY(launch_kernel_matrix(ctx, gp->kern, static_cast<double*>(gp->X.p),
                               nf, static_cast<double*>(gp->X.p), nf, K, ld, 1,
                               gp->noise_var + 1e-8 + gp->jitter, n));
  SGP_HIP(ctx, hipMemsetAsync(Li, 0, mat, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(info_dev, 0, sizeof(int), ctx->stream));
  SGP_TRY(factor_rec(ctx, K, Li, T, ld, 0, nf, info_dev));
  SGP_TRY(sgp_d2h(ctx, info, info_dev, sizeof(int)));
  if (*info != 0) return 0;

  SGP_TRY(launch_tri_mv(ctx, Li, ld, n, static_cast<double*>(gp->Y.p), 0, 1, tv, 0));
  SGP_TRY(launch_tri_mtv(ctx, Li, ld, n, tv, 0, 1,
                         static_cast<double*>(gp->alpha.p), 0, np));
  return publish_gp(gp);
}

// ---- one-row updates --------------------------------------------------------------
// Bordered Cholesky: with k = k(X, x*), t = L^-1 k, w = L^-T t = Ky^-1 k,
// s2 = k(x*,x*) + noise + 1e-8 + jitter - |t|^2:
//   new row of L^-1 = [ -w^T / s , 1 / s ],
//   alpha <- [ alpha - w r / s2 ; r / s2 ],  r = y* - k^T alpha  (= y* - mu(x*)).
// The record {w, r/s2, 1/s2, x*} is kept for the rank-1 update of the resident
// posterior (k_rank1).  One workgroup; n <= a few thousand.
__global__ __launch_bounds__(1024) void k_append_finish(
    double* Li, int64_t ld, int n, int d, const double* kc, const double* Tt,
    const double* Wt, double prior, double y, const double* xnew, double* alpha,
    double* updw, double* upd, int n_pad_new, int* info) {
  __shared__ double sh[2][1024 / 64];
  double a = 0.0, b = 0.0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    a = fma(Tt[j], Tt[j], a);
    b = fma(kc[j], alpha[j], b);
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
  double tn2 = 0.0, mu = 0.0;
  for (int w = 0; w < int(blockDim.x >> 6); ++w) {
    tn2 += sh[0][w];
    mu += sh[1][w];
  }
  const double s2 = prior - tn2;
  // GPy would retry with jitter here; report and let the host refit instead
  if (!(s2 > 1e-12 * prior) || !isfinite(s2)) {
    if (threadIdx.x == 0) info[0] = n + 1;
    return;
  }
  const double sd = sqrt(s2);
  const double r = y - mu;
  for (int j = threadIdx.x; j < n_pad_new; j += blockDim.x) {
    if (j < n) {
      const double w = Wt[j];
      Li[int64_t(n) * ld + j] = -w / sd;
      alpha[j] -= w * r / s2;
      updw[j] = w;
    } else {
      if (j == n) {
        Li[int64_t(n) * ld + n] = 1.0 / sd;
        alpha[n] = r / s2;
      } else {
        alpha[j] = 0.0;
      }
      updw[j] = 0.0;
    }
  }
  if (threadIdx.x == 0) {
    upd[0] = r / s2;
    upd[1] = 1.0 / s2;
    info[0] = 0;
  }
  if (threadIdx.x < d) upd[2 + threadIdx.x] = xnew[threadIdx.x];
}

// gp->X already holds the new row at index gp->n (uploaded by the caller);
// gp->Y gets y.  On success n grows by one; *info != 0 leaves the GP untouched.
int append_gp(sgp_gp* gp, double y, int* info) {
  sgp_ctx* ctx = gp->ctx;
  const int n = int(gp->n), ld = gp->ld, d = gp->kern.d;
  double* Li = static_cast<double*>(gp->Linv.p);
  double* X = static_cast<double*>(gp->X.p);
  double* buf = static_cast<double*>(sgp_scratch(ctx, 3, size_t(3) * ld * 8 + 64));
  if (!buf) return -1;
  double* Kc = buf;
  double* Tt = buf + ld;
  double* Wt = buf + 2 * size_t(ld);
  int* info_dev = reinterpret_cast<int*>(buf + 3 * size_t(ld));
  const double* xnew = X + size_t(n) * d;
  SGP_TRY(launch_kernel_matrix(ctx, gp->kern, xnew, 1, X, n, Kc, ld, 0, 0.0,
                               INT64_MAX));
  SGP_TRY(launch_tri_mv(ctx, Li, ld, n, Kc, ld, 1, Tt, ld));
  SGP_TRY(launch_tri_mtv(ctx, Li, ld, n, Tt, ld, 1, Wt, ld, n));
  const int np_new = (n + 1 + 15) / 16 * 16;
  const double prior = gp->kern.kdiag + gp->noise_var + 1e-8 + gp->jitter;
  hipLaunchKernelGGL(k_append_finish, dim3(1), dim3(1024), 0, ctx->stream, Li,
                     int64_t(ld), n, d, Kc, Tt, Wt, prior, y, xnew,
                     static_cast<double*>(gp->alpha.p),
                     static_cast<double*>(gp->updw.p),
                     static_cast<double*>(gp->upd.p), np_new, info_dev);
  SGP_HIP(ctx, hipGetLastError());
  SGP_TRY(sgp_d2h(ctx, info, info_dev, sizeof(int)));
  if (*info != 0) return 0;
  gp->n = n + 1;
  gp->n_pad = np_new;
  gp->n_f = (n + 1 + 31) / 32 * 32;
  gp->upd_valid = true;
  return publish_gp(gp);
}

// Drop the last row: the leading principal block of a lower-triangular inverse
// is the inverse of the leading block, so only alpha has to be recomputed.
int pop_gp(sgp_gp* gp) {
  sgp_ctx* ctx = gp->ctx;
  const int n = int(gp->n) - 1, ld = gp->ld;
  double* Li = static_cast<double*>(gp->Linv.p);
  double* tv = static_cast<double*>(gp->tvec.p);
  gp->n = n;
  gp->n_pad = (n + 15) / 16 * 16;
  gp->n_f = (n + 31) / 32 * 32;
  gp->upd_valid = false;
  SGP_TRY(launch_tri_mv(ctx, Li, ld, n, static_cast<double*>(gp->Y.p), 0, 1, tv, 0));
  SGP_TRY(launch_tri_mtv(ctx, Li, ld, n, tv, 0, 1,
                         static_cast<double*>(gp->alpha.p), 0, gp->n_pad));
  return publish_gp(gp);
}

// ---- posterior of a FEW points -------------------------------------------------------
// SafeOptSwarm's default swarm has 20 particles and SafeOpt asks for the
// posterior of single points (gp_opt.py:1117, 1132): one 64-row tile of the
// sweep kernel would walk through all n^2/512 block products on ONE compute
// unit.  For P <= kSmallPoints the work is spread over the chip the other way
// round -- one workgroup per 16-row block of L^-1, 16 points per pass:
//   k_small_kb   : Kb = k(X, pts) in MFMA B-operand order, zero padded
//   k_small_mfma : |L^-1 Kb|^2 per (row block, point) on the matrix cores (the
//                  packed A operands of the sweep), 16 waves splitting the
//                  k-steps; the last row block also forms alpha . Kb
//   k_small_post : var = k(x,x) - sum over row blocks, GPy clip
// Three launches for ALL GPs (blockIdx.z = GP) whose time does not depend on
// n^2 per compute unit.
template <int D>
__global__ void k_small_kb(const GpDev* gps, const double* pts, int P, SmallBufs sb) {
  const GpDev& gp = gps[blockIdx.z];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int pass = blockIdx.y;
  if (e >= (gp.n_pad / 4) * 64) return;
  const int lane = e & 63, s = e >> 6;
  const int pt = pass * 16 + (lane & 15), j = 4 * s + (lane >> 4);
  double v = 0.0;
  if (pt < P && j < gp.n) {
    double a[D], b[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      a[k] = pts[int64_t(pt) * D + k];
      b[k] = gp.Xpad[int64_t(j) * D + k];
    }
    v = kern_eval<D>(gp.kern, a, b);
  }
  sb.Kb[blockIdx.z * sb.kb_stride + int64_t(pass) * sb.nsteps_max * 64 + e] = v;
}

// One workgroup per 16-row block of L^-1 and pass of 16 points; its 16 waves
// split the k-steps (the last row block has n / 4 of them: with four waves it was
// a chain of n / 16 exposed load latencies) and fold their partial products
// through LDS.
constexpr int kSmallWaves = 16;

__global__ __launch_bounds__(64 * kSmallWaves) void k_small_mfma(const GpDev* gps,
                                                                 SmallBufs sb) {
  __shared__ double4_t sh[kSmallWaves][64];
  __shared__ double shm[kSmallWaves][16];
  const GpDev& gp = gps[blockIdx.z];
  const int nblk = gp.nblk, nsteps = gp.n_pad / 4;
  const int blk = blockIdx.x, pass = blockIdx.y;
  if (blk >= nblk) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* A = gp.Apack + int64_t(blk) * nsteps * 64 + lane;
  const double* B = sb.Kb + blockIdx.z * sb.kb_stride +
                    int64_t(pass) * sb.nsteps_max * 64 + lane;
  const bool last = blk == nblk - 1;            // covers every k-step: the mean
  // narrow packing of the last row block (k_pack): its rows 4..15 repeat rows 0..3
  const bool dup = last && gp.narrow && (lane & 15) >= 4;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  double m = 0.0;
  const int send = (blk + 1) * 4;               // up to the diagonal block
  constexpr int kInFlight = 8;                  // k-steps of a wave per batch (16: slower)
  for (int s = wave; s < send; s += kInFlight * kSmallWaves) {
    // all loads of the batch are issued before the first use
    double av[kInFlight], bv[kInFlight], al[kInFlight];
#pragma unroll
    for (int i = 0; i < kInFlight; ++i) {
      const int si = s + i * kSmallWaves;
      const bool ok = si < send;
      av[i] = (ok && !dup) ? A[si * 64] : 0.0;
      bv[i] = ok ? B[si * 64] : 0.0;
      al[i] = (ok && last) ? gp.alpha[4 * si + (lane >> 4)] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < kInFlight; ++i) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc, 0, 0, 0);
      m = fma(al[i], bv[i], m);
    }
  }
  sh[wave][lane] = acc;
  if (last) {
    m += __shfl_xor(m, 16, 64);
    m += __shfl_xor(m, 32, 64);
    if (lane < 16) shm[wave][lane] = m;
  }
  __syncthreads();
  if (wave == 0) {
    // D layout: column (point) = lane & 15, rows (lane >> 4) + 4 r
    double4_t t = sh[0][lane];
#pragma unroll
    for (int w = 1; w < kSmallWaves; ++w) t += sh[w][lane];
    double ss = (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) {
      sb.part[blockIdx.z * sb.part_stride + (int64_t(pass) * sb.nblk_max + blk) * 16 + lane] = ss;
      if (last) {
        double mm = shm[0][lane];
#pragma unroll
        for (int w = 1; w < kSmallWaves; ++w) mm += shm[w][lane];
        sb.mtmp[(blockIdx.z * sb.passes + pass) * 16 + lane] = mm;
      }
    }
  }
}

// var = k(x,x) - sum over the row blocks (small_block_sum), GPy clip.  One
// workgroup per pass of 16 points and GP; mean / var are [G][P].
__global__ __launch_bounds__(256) void k_small_post(const GpDev* gps, SmallBufs sb,
                                                    int P, double* mean, double* var) {
  __shared__ double sh[16][16];
  const int g = blockIdx.y, pass = blockIdx.x;
  const GpDev& gp = gps[g];
  const double tot = small_block_sum(sb.part + g * sb.part_stride, sb.nblk_max,
                                     gp.nblk, pass, sh);
  const int p = pass * 16 + (threadIdx.x & 15);
  if ((threadIdx.x >> 4) == 0 && p < P) {
    mean[int64_t(g) * P + p] = sb.mtmp[g * sb.passes * 16 + p];
    var[int64_t(g) * P + p] = fmax(gp.kern.kdiag - tot, 1e-15);      // GPy clip
  }
}

// ---- operands of the rank-1 expander test, all GPs per launch ----------------------
// For candidate c and GP g:  k_c = k(X, x_c),  t = L^-1 k_c,  w = L^-T t = Ky^-1 k_c
// (packed as an MFMA A operand: cand x j),  s2 = prior - |t|^2,  delta = resid / s2.
namespace {
struct ExpGpSel {
  int active[SGP_MAX_GPS];
};

// 2 .. thousands of candidates (sgp_grid_expander_batch, sgp_grid_expander_pass) in groups of
// 16; every per-candidate array is laid out [group][GP slot][...] (ExpanderOps::Gs slots), the
// group's candidate count is what is left of `m` behind the groups in front.  The two
// triangular products are matrix products of 16-candidate panels on v_mfma_f64_16x16x4_f64:
// per (group, GP) the panels Kc and T are [j][16 candidates] -- 64 consecutive doubles are a
// B operand (k_expt_mm: lane 16 k + c <- [4 s + k][c]) or an A operand (k_expw_mm) as they lie.
constexpr int kOpGroups = 2;       // groups of a wave: one operand of L^-1 feeds that many products

// Kc[group][g][j][c] = k(x_c, X_j); zero for j >= n (the padding meets zeros of L^-1) and c >= m
template <int D>
__global__ __launch_bounds__(256) void k_expk(const GpDev* gps, ExpGpSel sel,
                                              const double* xc, int m, double* Kc,
                                              int64_t ldk, int Gs) {
  const int g = blockIdx.y;
  if (!sel.active[g]) return;
  const int z = blockIdx.z;
  xc += int64_t(z) * kMaxRhs * D;
  m = min(kMaxRhs, m - kMaxRhs * z);
  const GpDev& gp = gps[g];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= gp.n_pad) return;
  double* out = Kc + ((int64_t(z) * Gs + g) * ldk + j) * kMaxRhs;
  double xj[D];
#pragma unroll
  for (int k = 0; k < D; ++k) xj[k] = gp.Xpad[int64_t(j) * D + k];
  for (int c = 0; c < kMaxRhs; ++c)
    out[c] = (j < gp.n && c < m) ? kern_eval<D>(gp.kern, xc + c * D, xj) : 0.0;
}

// T[group][g][i][c] = sum_{j <= i} Li[i][j] Kc[..][j][c]: a wave per (row block of 16, kOpGroups
// groups); A = the packed L^-1 of the sweeps (GpDev::Apack: zero above the diagonal and in the
// padding), B = the panel.  D: column (candidate) = lane & 15, rows (lane >> 4) + 4 r.
__global__ __launch_bounds__(256) void k_expt_mm(const GpDev* gps, ExpGpSel sel,
                                                 const double* Kc, double* Tt,
                                                 int64_t ldk, int Gs, int ngroups) {
  const int g = blockIdx.y;
  if (!sel.active[g]) return;
  const GpDev& gp = gps[g];
  const int b = blockIdx.x;
  if (b >= gp.nblk) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int z0 = (blockIdx.z * 4 + wave) * kOpGroups;
  if (z0 >= ngroups) return;
  const int nsteps = gp.n_pad >> 2;
  const double* A = gp.Apack + int64_t(b) * nsteps * 64 + lane;
  // narrow packing of the last row block (k_pack): its rows 4..15 repeat rows 0..3
  const bool dup = b == gp.nblk - 1 && gp.narrow && (lane & 15) >= 4;
  const double* B[kOpGroups];
#pragma unroll
  for (int q = 0; q < kOpGroups; ++q)
    B[q] = Kc + (int64_t(min(z0 + q, ngroups - 1)) * Gs + g) * ldk * kMaxRhs + lane;
  double4_t acc[kOpGroups];
#pragma unroll
  for (int q = 0; q < kOpGroups; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int send = (b + 1) * 4;                 // up to the diagonal block
  constexpr int kInFlight = 4;
#pragma unroll 1
  for (int s = 0; s < send; s += kInFlight) {   // (send is a multiple of 4)
    double av[kInFlight], bv[kOpGroups][kInFlight];
#pragma unroll
    for (int i = 0; i < kInFlight; ++i) {
      av[i] = A[(s + i) * 64];
#pragma unroll
      for (int q = 0; q < kOpGroups; ++q) bv[q][i] = B[q][(s + i) * 64];
    }
#pragma unroll
    for (int i = 0; i < kInFlight; ++i) {
      const double a = dup ? 0.0 : av[i];
#pragma unroll
      for (int q = 0; q < kOpGroups; ++q)
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[q][i], acc[q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < kOpGroups; ++q) {
    if (z0 + q < ngroups) {
      double* T = Tt + (int64_t(z0 + q) * Gs + g) * ldk * kMaxRhs;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        T[(16 * b + (lane >> 4) + 4 * r) * kMaxRhs + (lane & 15)] = acc[q][r];
    }
  }
}

// W[c][j] = sum_{i >= j} Li[i][j] T[..][i][c] straight into the packed operand
// (Wpack[(j / 4) * 64 + (j % 4) * 16 + c], zero for j >= n and c >= m): a wave per (column block
// of 16, kOpGroups groups); A = the panel, B = rows of the dense L^-1 (16 consecutive columns).
// D: column j = lane & 15, rows (candidates) (lane >> 4) + 4 r.  The wave of column block 0 walks
// over every row: it also leaves s2 / delta / |t|^2 of its candidates.
__global__ __launch_bounds__(256) void k_expw_mm(const GpDev* gps, ExpGpSel sel,
                                                 const double* Tt, int64_t ldk, ExpanderOps ops,
                                                 int ngroups) {
  const int g = blockIdx.y;
  if (!sel.active[g]) return;
  const GpDev& gp = gps[g];
  const int jb = blockIdx.x;
  if (jb >= gp.nblk) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int z0 = (blockIdx.z * 4 + wave) * kOpGroups;
  if (z0 >= ngroups) return;
  const int n = gp.n;
  const int j = 16 * jb + (lane & 15);
  const double* A[kOpGroups];
#pragma unroll
  for (int q = 0; q < kOpGroups; ++q)
    A[q] = Tt + (int64_t(min(z0 + q, ngroups - 1)) * ops.Gs + g) * ldk * kMaxRhs + lane;
  const double* Bp = gp.Linv + int64_t(lane >> 4) * gp.ld + min(j, n - 1);
  double4_t acc[kOpGroups];
  double ss[kOpGroups];
#pragma unroll
  for (int q = 0; q < kOpGroups; ++q) {
    acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
    ss[q] = 0.0;
  }
  const int send = (n + 3) >> 2;
  constexpr int kInFlight = 4;
#pragma unroll 1
  for (int s = 4 * jb; s < send; s += kInFlight) {
    double av[kOpGroups][kInFlight], bv[kInFlight];
#pragma unroll
    for (int i = 0; i < kInFlight; ++i) {
      const int row = 4 * (s + i) + (lane >> 4);
      bv[i] = Bp[int64_t(min(4 * (s + i), n - 1 - (lane >> 4))) * gp.ld];
      bv[i] = (row < n && row >= j && j < n) ? bv[i] : 0.0;
      // (rows >= n of the panel are zero: k_expt_mm, the padding of Apack)
#pragma unroll
      for (int q = 0; q < kOpGroups; ++q) av[q][i] = A[q][min(s + i, (gp.n_pad >> 2) - 1) * 64];
    }
#pragma unroll
    for (int i = 0; i < kInFlight; ++i) {
#pragma unroll
      for (int q = 0; q < kOpGroups; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][i], bv[i], acc[q], 0, 0, 0);
        ss[q] = (s + i < send) ? fma(av[q][i], av[q][i], ss[q]) : ss[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kOpGroups; ++q) {
    const int z = z0 + q;
    if (z >= ngroups) break;
    const int m = min(kMaxRhs, ops.m - kMaxRhs * z);
    if (j < gp.n_pad) {
      double* Wp = ops.Wpack + (int64_t(z) * ops.Gs + g) * ops.wstride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = (lane >> 4) + 4 * r;
        Wp[(j >> 2) * 64 + (j & 3) * 16 + c] = (j < n && c < m) ? acc[q][r] : 0.0;
      }
    }
    if (jb == 0) {
      double t2 = ss[q];
      t2 += __shfl_xor(t2, 16, 64);
      t2 += __shfl_xor(t2, 32, 64);
      const int c = lane;
      if (c < m) {
        const int64_t o = (int64_t(z) * ops.Gs + g) * 16 + c;
        const double s2 = gp.prior - t2;
        ops.inv_s2[o] = 1.0 / s2;
        ops.delta[o] = ops.resid[o] / s2;
        ops.tn2[o] = t2;
      }
    }
  }
}

// One candidate (the fused single-GPU step and the probe of the first candidate):
// T[g][0][i] with k_c evaluated where it is needed -- lane l of the wave that owns
// row i evaluates k(x_c, X_j) for its j = l, l + 64, ... <= i -- instead of a
// separate launch that writes k_c out first.
template <int D>
__global__ __launch_bounds__(256) void k_expkt(const GpDev* gps, ExpGpSel sel,
                                               const double* xc, double* Tt,
                                               int64_t ldk, FrontArgs fa, int g_writer) {
  const int g = blockIdx.y;
  if (!sel.active[g]) return;
  const GpDev& gp = gps[g];
  const int lane = threadIdx.x & 63;
  double x[D];
  if (fa.nb > 0) {
    // one-rank chain: the candidate is not staged yet -- the last step of the front half
    // runs here (sets_front.h); one workgroup leaves the result block and the staged
    // operand for the kernels behind this one
    const int64_t top = front_final_fold(fa, blockIdx.x == 0 && g == g_writer);
#pragma unroll
    for (int k = 0; k < D; ++k)
      x[k] = top >= 0 ? fa.pts[int64_t(k) * fa.N + (top - fa.goff)] : 0.0;
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = xc[k];
  }
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= gp.n) return;
  const double* row = gp.Linv + int64_t(i) * gp.ld;
  double acc = 0.0;
  for (int j = lane; j <= i; j += 64) {
    double xj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xj[k] = gp.Xpad[int64_t(j) * D + k];
    acc = fma(row[j], kern_eval<D>(gp.kern, x, xj), acc);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) Tt[int64_t(g) * kMaxRhs * ldk + i] = acc;
}

// ... and W[0][j] for it: 64 columns per workgroup, the 16 waves split the rows
// i >= j0 (four loads in flight per lane), one fold through LDS.
__global__ __launch_bounds__(1024) void k_expw1(const GpDev* gps, ExpGpSel sel,
                                                const double* Tt, int64_t ldk,
                                                ExpanderOps ops) {
  __shared__ double sh[16][64];
  const int g = blockIdx.y;
  if (!sel.active[g]) return;
  const GpDev& gp = gps[g];
  const int n = gp.n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * 64, j = j0 + lane;
  if (j0 >= gp.n_pad) return;
  const double* T = Tt + int64_t(g) * kMaxRhs * ldk;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (j < n) {
    const double* col = gp.Linv + j;
    int i = j0 + wave;
    for (; i + 48 < n; i += 64) {
      const double l0 = (i >= j) ? col[int64_t(i) * gp.ld] : 0.0;
      const double l1 = (i + 16 >= j) ? col[int64_t(i + 16) * gp.ld] : 0.0;
      const double l2 = (i + 32 >= j) ? col[int64_t(i + 32) * gp.ld] : 0.0;
      const double l3 = (i + 48 >= j) ? col[int64_t(i + 48) * gp.ld] : 0.0;
      a0 = fma(l0, T[i], a0);
      a1 = fma(l1, T[i + 16], a1);
      a2 = fma(l2, T[i + 32], a2);
      a3 = fma(l3, T[i + 48], a3);
    }
    for (; i < n; i += 16)
      if (i >= j) a0 = fma(col[int64_t(i) * gp.ld], T[i], a0);
  }
  sh[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && j < gp.n_pad) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sh[w][lane];
    double* Wp = ops.Wpack + int64_t(g) * ops.wstride + (j >> 2) * 64 + (j & 3) * 16;
    Wp[0] = (j < n) ? tot : 0.0;
#pragma unroll
    for (int c = 1; c < kMaxRhs; ++c) Wp[c] = 0.0;
  }
  if (blockIdx.x == 0 && wave == 1) {
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s = fma(T[i], T[i], s);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
      const double s2 = gp.prior - s;
      ops.inv_s2[g * 16] = 1.0 / s2;
      ops.delta[g * 16] = ops.resid[g * 16] / s2;
      ops.tn2[g * 16] = s;
    }
  }
}

}  // namespace

int expander_operands_all(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                          int G, int d, const ExpanderOps& ops, const FrontArgs* fold) {
  int n_max = 0, np_max = 0, g_writer = -1;
  ExpGpSel sel{};
  FrontArgs fa{};
  if (fold) fa = *fold;
  for (int g = 0; g < G; ++g) {
    sel.active[g] = ops.active[g];
    if (!ops.active[g]) continue;
    if (g_writer < 0) g_writer = g;
    n_max = std::max(n_max, gps_host[g].n);
    np_max = std::max(np_max, gps_host[g].n_pad);
  }
  if (n_max == 0) {
    // no GP with a constraint: nothing evaluates the candidate, the fold has no host
    if (fold) return launch_front_final(ctx, *fold);
    return 0;
  }
  SGP_CHECK(ctx, !fold || ops.m == 1, "the front fold goes with one candidate");
  const int64_t ldk = (np_max + 31) / 32 * 32;
  const int ngroups = (ops.m + kMaxRhs - 1) / kMaxRhs;
  SGP_CHECK(ctx, ngroups == 1 || ops.Gs == G, "ExpanderOps::Gs = %d for %d GPs", ops.Gs, G);
  double* buf = static_cast<double*>(
      sgp_scratch(ctx, 3, size_t(2) * ngroups * G * kMaxRhs * ldk * sizeof(double)));
  SGP_CHECK(ctx, buf, "device allocation failed: %s", ctx->err.c_str());
  double* Kc = buf;
  double* Tt = buf + size_t(ngroups) * G * kMaxRhs * ldk;
  if (ops.m == 1) {
#define EXPKT_CASE(DD)                                                        \
  case DD:                                                                    \
    hipLaunchKernelGGL(k_expkt<DD>, dim3((n_max + 3) / 4, G), dim3(256), 0,   \
                       ctx->stream, gps_dev, sel, ops.xc, Tt, ldk, fa,        \
                       g_writer);                                             \
    break;
    switch (d) {
      EXPKT_CASE(1) EXPKT_CASE(2) EXPKT_CASE(3) EXPKT_CASE(4)
      EXPKT_CASE(5) EXPKT_CASE(6) EXPKT_CASE(7) EXPKT_CASE(8)
      default:
        sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
        return -2;
    }
#undef EXPKT_CASE
    hipLaunchKernelGGL(k_expw1, dim3((np_max + 63) / 64, G), dim3(1024), 0,
                       ctx->stream, gps_dev, sel, Tt, ldk, ops);
    SGP_HIP(ctx, hipGetLastError());
    return 0;
  }
#define EXPK_CASE(DD)                                                         \
  case DD:                                                                    \
    hipLaunchKernelGGL(k_expk<DD>, dim3((np_max + 255) / 256, G, ngroups), dim3(256), 0,\
                       ctx->stream, gps_dev, sel, ops.xc, ops.m, Kc, ldk, G); \
    break;
  switch (d) {
    EXPK_CASE(1) EXPK_CASE(2) EXPK_CASE(3) EXPK_CASE(4)
    EXPK_CASE(5) EXPK_CASE(6) EXPK_CASE(7) EXPK_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
#undef EXPK_CASE
  const int zblocks = (ngroups + 4 * kOpGroups - 1) / (4 * kOpGroups);
  hipLaunchKernelGGL(k_expt_mm, dim3(np_max / 16, G, zblocks), dim3(256), 0, ctx->stream,
                     gps_dev, sel, Kc, Tt, ldk, G, ngroups);
  hipLaunchKernelGGL(k_expw_mm, dim3(np_max / 16, G, zblocks), dim3(256), 0, ctx->stream,
                     gps_dev, sel, Tt, ldk, ops, ngroups);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

bool small_path_pays(const sgp_gp* gp, int64_t P) {
  // The sweep needs >= 256 tiles of 64 rows to fill the chip; a tile walks
  // through all n^2 / 512 block products on one compute unit (n = 2000: 2.8 ms
  // whatever the number of tiles up to 512).  The few-points path spreads
  // (P / 16) x (n / 16) workgroups of a few microseconds each over the chip:
  // n = 2000, P = 2000: 0.3 ms.  Below n ~ 128 a tile is as short as the three
  // launches (measured crossover, scripts/dev/swarm_small.py: n between 50 and 200).
  return P >= 1 && P <= kSmallPoints && gp->n >= 128;
}

int small_reserve(sgp_ctx* ctx, const GpDev* gps_host, int G, int P, SmallBufs* sb) {
  int np_max = 0;
  for (int g = 0; g < G; ++g) np_max = std::max(np_max, gps_host[g].n_pad);
  sb->passes = (P + 15) / 16;
  sb->nsteps_max = np_max / 4;
  sb->nblk_max = np_max / 16;
  sb->kb_stride = int64_t(sb->passes) * sb->nsteps_max * 64;
  sb->part_stride = int64_t(sb->passes) * sb->nblk_max * 16;
  double* buf = static_cast<double*>(sgp_scratch(
      ctx, 6, size_t(G) * (sb->kb_stride + sb->part_stride + sb->passes * 16) * sizeof(double)));
  if (!buf) return -1;
  sb->Kb = buf;
  sb->part = buf + size_t(G) * sb->kb_stride;
  sb->mtmp = sb->part + size_t(G) * sb->part_stride;
  return 0;
}

// |L^-1 k(X, pts)|^2 per row block and alpha . k(X, pts) of all G GPs (two
// launches); `post` adds the block sums -> mean / var [G][P].
int posterior_small_all(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                        int G, const double* pts_rowmajor, int P, const SmallBufs& sb,
                        double* mean, double* var) {
  const int d = gps_host[0].kern.d;
  ctx->last_sweep = 4;
#define SMALL_CASE(DD)                                                         \
  case DD:                                                                     \
    hipLaunchKernelGGL(k_small_kb<DD>,                                         \
                       dim3((sb.nsteps_max * 64 + 255) / 256, sb.passes, G),   \
                       dim3(256), 0, ctx->stream, gps_dev, pts_rowmajor, P, sb);\
    break;
  switch (d) {
    SMALL_CASE(1) SMALL_CASE(2) SMALL_CASE(3) SMALL_CASE(4)
    SMALL_CASE(5) SMALL_CASE(6) SMALL_CASE(7) SMALL_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
#undef SMALL_CASE
  hipLaunchKernelGGL(k_small_mfma, dim3(sb.nblk_max, sb.passes, G),
                     dim3(64 * kSmallWaves), 0, ctx->stream, gps_dev, sb);
  if (mean && var)
    hipLaunchKernelGGL(k_small_post, dim3(sb.passes, G), dim3(256), 0, ctx->stream,
                       gps_dev, sb, P, mean, var);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}
