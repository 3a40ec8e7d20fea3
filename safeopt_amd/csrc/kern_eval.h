// Device-side covariance-function evaluation (GPy Stationary.K_of_r restated
// for one lane): RBF / Matern-3/2 / Matern-5/2, ARD lengthscales, products of
// parts on arbitrary column subsets.  Reference call sites: gp.kern.K reached
// through gp.predict_noiseless (safeopt/gp_opt.py:469, 591, 929, 973).
//
// fp64 VALU work is not free next to the fp64 matrix pipe on gfx950 (they
// share the FP64 units: MFMA-only 49 TF/s, v_fma_f64-only 66, interleaved sum
// ~52 -- scripts/microbench.py), so the per-element cost of the covariance is
// trimmed: inputs pre-scaled by 1/lengthscale, and exp() through a 32-entry
// 2^(j/32) table (one LDS bank row, conflict free) + a degree-6 polynomial:
// ~13 fp64 ops instead of the library's ~27, accurate to ~1.5 ulp.
#pragma once

#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kExpTabSize = 32;

// Fill the 2^(j/32) table (call by all threads, then barrier).
__device__ __forceinline__ void exp_tab_init(double* tab) {
  if (threadIdx.x < kExpTabSize) tab[threadIdx.x] = exp2(threadIdx.x * (1.0 / 32.0));
}

// exp(x) for x <= 0 (any x works; large negative x underflows to 0).
__device__ __forceinline__ double exp_tab(double x, const double* tab) {
  x = fmax(x, -745.2);
  const double kf = rint(x * 46.16624130844683);       // 32 / ln 2
  double r = fma(kf, -0.02166084937925916, x);         // ln2/32, high part
  r = fma(kf, -1.3239129268154012e-11, r);             //         low part
  const int k = int(kf);
  const double t = tab[k & 31];
  // exp(r), |r| <= ln2/64: truncation r^7/5040 < 4e-18
  double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
  p = fma(r, p, 1.0 / 24.0);
  p = fma(r, p, 1.0 / 6.0);
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  return ldexp(t * p, k >> 5);
}

// Four exp() at once, written step-major with scheduling fences: the compiler
// otherwise runs the four ~20-deep dependent chains one after the other (it
// minimises live registers at the 256-VGPR limit), which leaves the kernel
// latency bound; step-major order keeps 4 independent instructions in flight.
#define SGP_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void exp_tab4(const double (&xin)[4],
                                         const double* tab, double (&out)[4]) {
  double x[4], kf[4], r[4], t[4], p[4];
  int k[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = fmax(xin[q], -745.2);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) kf[q] = rint(x[q] * 46.16624130844683);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    k[q] = int(kf[q]);
    r[q] = fma(kf[q], -0.02166084937925916, x[q]);
  }
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    t[q] = tab[k[q] & 31];
    r[q] = fma(kf[q], -1.3239129268154012e-11, r[q]);
  }
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = fma(r[q], 1.0 / 720.0, 1.0 / 120.0);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = fma(r[q], p[q], 1.0 / 24.0);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = fma(r[q], p[q], 1.0 / 6.0);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = fma(r[q], p[q], 0.5);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = fma(r[q], p[q], 1.0);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = fma(r[q], p[q], 1.0);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = ldexp(t[q] * p[q], k[q] >> 5);
  SGP_FENCE();
}

// Four square roots, step-major (rsq seed + two Goldschmidt steps + a final
// correction: <= 1 ulp for normal inputs; 0 maps to ~1e-150, which is 0 for
// the Matern factors).
__device__ __forceinline__ void sqrt4(const double (&xin)[4], double (&out)[4]) {
  double x[4], y[4], g[4], h[4], r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = fmax(xin[q], 1e-300);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) y[q] = __builtin_amdgcn_rsq(x[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    g[q] = x[q] * y[q];
    h[q] = 0.5 * y[q];
  }
  SGP_FENCE();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = fma(-h[q], g[q], 0.5);
    SGP_FENCE();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g[q] = fma(g[q], r[q], g[q]);
      h[q] = fma(h[q], r[q], h[q]);
    }
    SGP_FENCE();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) r[q] = fma(-g[q], g[q], x[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = fma(r[q], h[q], g[q]);
  SGP_FENCE();
}

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

__device__ __forceinline__ double k_of_r2_tab(int kind, double r2,
                                              const double* tab) {
  if (kind == SGP_RBF) return exp_tab(-0.5 * r2, tab);
  const double r = sqrt(r2);
  if (kind == SGP_MATERN32) {
    const double a = 1.7320508075688772 * r;
    return (1.0 + a) * exp_tab(-a, tab);
  }
  const double a = 2.23606797749979 * r;
  return (1.0 + a + (5.0 / 3.0) * r2) * exp_tab(-a, tab);
}

// k(x, y) for the product kernel `kd`; x and y are raw D-vectors.
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
// (one stationary part) works on inputs pre-scaled by 1/lengthscale and keeps
// everything in registers; products of parts fall back to the descriptor loop
// on raw inputs.
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

  // candidate row -> the form operator() expects (scaled when `single`)
  __device__ __forceinline__ void prep(const double* x, double* xs) const {
#pragma unroll
    for (int i = 0; i < D; ++i) xs[i] = single ? x[i] * il0[i] : x[i];
  }

  // xs from prep(); ys = row of GpDev::Xs (pre-scaled when `single`)
  __device__ __forceinline__ double operator()(const double* xs,
                                               const double* ys,
                                               const double* tab) const {
    if (single) {
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double t = xs[i] - ys[i];
        r2 = fma(t, t, r2);
      }
      return var0 * k_of_r2_tab(kind0, r2, tab);
    }
    return kern_eval<D>(*kd, xs, ys);
  }

  // NV evaluations in one go: ys[q] = rows q*stride of the (pre-scaled when
  // `single`) training inputs.  The kind switch sits OUTSIDE the loop so the NV
  // dependent chains (distance, exp polynomial) are interleaved by the
  // scheduler instead of running one after the other.
  template <int NV>
  __device__ __forceinline__ void many(const double* xs, const double* ys,
                                       int stride, const double* tab,
                                       double (&out)[NV]) const {
    static_assert(NV == 4, "the batched evaluation is written for 4 values");
    if (single) {
      double r2[4], arg[4], e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r2[q] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const double t = xs[i] - ys[q * stride + i];
          r2[q] = fma(t, t, r2[q]);
        }
      }
      if (kind0 == SGP_RBF) {
#pragma unroll
        for (int q = 0; q < 4; ++q) arg[q] = -0.5 * r2[q];
        exp_tab4(arg, tab, e);
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = var0 * e[q];
      } else {
        double rr[4];
        sqrt4(r2, rr);
        const double c = (kind0 == SGP_MATERN32) ? 1.7320508075688772
                                                 : 2.23606797749979;
#pragma unroll
        for (int q = 0; q < 4; ++q) arg[q] = -c * rr[q];
        exp_tab4(arg, tab, e);
        if (kind0 == SGP_MATERN32) {
#pragma unroll
          for (int q = 0; q < 4; ++q) out[q] = var0 * (1.0 - arg[q]) * e[q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            out[q] = var0 * (1.0 - arg[q] + (5.0 / 3.0) * r2[q]) * e[q];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
        out[q] = kern_eval<D>(*kd, xs, ys + q * stride);
    }
  }

  // both arguments raw (unscaled) rows
  __device__ __forceinline__ double raw(const double* x, const double* y,
                                        const double* tab) const {
    if (single) {
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double t = (x[i] - y[i]) * il0[i];
        r2 = fma(t, t, r2);
      }
      return var0 * k_of_r2_tab(kind0, r2, tab);
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
