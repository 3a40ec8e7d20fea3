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
