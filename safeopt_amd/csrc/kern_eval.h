// Device-side covariance-function evaluation (GPy Stationary.K_of_r restated
// for one lane): RBF / Matern-3/2 / Matern-5/2, ARD lengthscales, products of
// parts on arbitrary column subsets.  Reference call sites: gp.kern.K reached
// through gp.predict_noiseless (safeopt/gp_opt.py:469, 591, 929, 973).
#pragma once

#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double k_of_r2(int kind, double r2) {
  if (kind == SGP_RBF) return exp(-0.5 * r2);
  const double r = sqrt(r2);
  if (kind == SGP_MATERN32) {
    const double a = 1.7320508075688772 * r;  // sqrt(3) r
    return (1.0 + a) * exp(-a);
  }
  const double a = 2.23606797749979 * r;  // sqrt(5) r
  return (1.0 + a + (5.0 / 3.0) * r2) * exp(-a);
}

// k(x, y) for the product kernel `kd`; x and y are D-vectors in registers/LDS.
template <int D>
__device__ __forceinline__ double kern_eval(const KernDesc& kd, const double* x,
                                            const double* y) {
  double diff[D];
#pragma unroll
  for (int k = 0; k < D; ++k) diff[k] = x[k] - y[k];
  double out = 1.0;
  for (int p = 0; p < kd.n_parts; ++p) {  // wave-uniform trip count
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double t = diff[k] * kd.inv_ls[p][k];
      r2 = fma(t, t, r2);
    }
    out *= kd.variance[p] * k_of_r2(kd.kind[p], r2);
  }
  return out;
}

// Hyper-parameters of one GP hoisted out of the inner loops.  The common case
// (one stationary part) keeps everything in registers / SGPRs; products of
// parts fall back to the descriptor loop.
template <int D>
struct KernFast {
  const KernDesc* kd;
  bool single;
  int kind0;
  double var0;
  double il0[D];

  __device__ __forceinline__ explicit KernFast(const KernDesc& k) : kd(&k) {
    single = k.n_parts == 1;
    kind0 = k.kind[0];
    var0 = k.variance[0];
#pragma unroll
    for (int i = 0; i < D; ++i) il0[i] = k.inv_ls[0][i];
  }

  __device__ __forceinline__ double operator()(const double* x,
                                               const double* y) const {
    if (single) {
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double t = (x[i] - y[i]) * il0[i];
        r2 = fma(t, t, r2);
      }
      return var0 * k_of_r2(kind0, r2);
    }
    return kern_eval<D>(*kd, x, y);
  }
};

// Sum over the four 16-lane groups of a wave: lanes l, l^16, l^32, l^48.
__device__ __forceinline__ double sum_lane_groups(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
