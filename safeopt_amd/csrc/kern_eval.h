// Device-side covariance-function evaluation (GPy Stationary.K_of_r restated
// for one lane): RBF / Matern-3/2 / Matern-5/2, ARD lengthscales, products of
// parts on arbitrary column subsets.  Reference call sites: gp.kern.K reached
// through gp.predict_noiseless (safeopt/gp_opt.py:469, 591, 929, 973).
//
// fp64 VALU work is not free next to the fp64 matrix pipe on gfx950: the two
// do not overlap (SQ_VALU_MFMA_COEXEC_CYCLES = 0; profiles/r01/ablation.txt:
// removing the evaluation shortens the sweep by exactly its own VALU time), so
// the per-element cost of the covariance is trimmed to the bone:
//   * inputs are stored pre-multiplied by 1/lengthscale AND by a per-kind unit
//     (common.h, kern_unit) chosen so that the covariance is
//         RBF     var * 2^(u/32),                   u = -|dz|^2
//         Matern  var * poly(u) * 2^(u/32),         u = -|dz|
//     -- no argument scaling, no clamp, and the range reduction w = u - rint(u)
//     is exact (no two-word ln2 constant);
//   * 2^(w/32), |w| <= 1/2, is a degree-6 polynomial (truncation < 4e-18);
//     2^(j/32) comes from a 32-entry LDS table (every entry has its own banks:
//     conflict free), the integer part goes through v_ldexp_f64.
// RBF: 16 fp64 ops per value (d = 2), Matern-5/2: ~30, against ~27 + distance
// for the library exp alone; accurate to ~2 ulp.
#pragma once

#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kExpTabSize = 32;

// Fill the 2^(j/32) table (call by all threads, then barrier).
__device__ __forceinline__ void exp_tab_init(double* tab) {
  if (threadIdx.x < kExpTabSize) tab[threadIdx.x] = exp2(threadIdx.x * (1.0 / 32.0));
}

// (ln2/32)^i / i!
#define SGP_E1 0.02166084939249829
#define SGP_E2 0.0002345961982022468
#define SGP_E3 1.693850972437182e-06
#define SGP_E4 9.172562701824643e-09
#define SGP_E5 3.973709984549416e-11
#define SGP_E6 1.4345655584131934e-13

// 2^(u/32) for u <= 0 (any finite u works; very negative u underflows to 0:
// the int conversion saturates and v_ldexp_f64 flushes).
__device__ __forceinline__ double exp2_32(double u, const double* tab) {
  const double kf = rint(u);
  const double w = u - kf;                 // exact, |w| <= 1/2
  const int k = int(kf);
  const double t = tab[k & 31];
  double p = fma(w, SGP_E6, SGP_E5);
  p = fma(w, p, SGP_E4);
  p = fma(w, p, SGP_E3);
  p = fma(w, p, SGP_E2);
  p = fma(w, p, SGP_E1);
  p = fma(w, p, 1.0);
  return ldexp(t * p, k >> 5);
}

// Four 2^(u/32) at once, written step-major with scheduling fences: the
// compiler otherwise runs the four dependent chains one after the other (it
// minimises live registers at the 256-VGPR limit), which leaves the kernel
// latency bound; step-major order keeps 4 independent instructions in flight.
#define SGP_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int NV>
__device__ __forceinline__ void exp2_32xn(const double (&u)[NV], const double* tab,
                                          double (&out)[NV]) {
  double kf[NV], w[NV], t[NV], p[NV];
  int k[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) kf[q] = rint(u[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    k[q] = int(kf[q]);
    w[q] = u[q] - kf[q];
  }
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    t[q] = tab[k[q] & 31];
    p[q] = fma(w[q], SGP_E6, SGP_E5);
  }
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) p[q] = fma(w[q], p[q], SGP_E4);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) p[q] = fma(w[q], p[q], SGP_E3);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) p[q] = fma(w[q], p[q], SGP_E2);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) p[q] = fma(w[q], p[q], SGP_E1);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) p[q] = fma(w[q], p[q], 1.0);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) out[q] = ldexp(t[q] * p[q], k[q] >> 5);
  SGP_FENCE();
}
__device__ __forceinline__ void exp2_32x4(const double (&u)[4], const double* tab,
                                          double (&out)[4]) {
  exp2_32xn<4>(u, tab, out);
}

// Four square roots, step-major: v_rsq_f64 seed (relative error 5e-8 on
// gfx950, scripts/dev/sqrt_accuracy.hip) and ONE Newton step
//   g = x y,  out = g + (x - g^2) (y / 2)
// which leaves a relative error of ~1.3e-15 -- far below what the covariance
// needs (it enters 2^(u/32) as an absolute error of |u| 1.3e-15 ln2/32 < 1e-13
// wherever the covariance is not already below 1e-16 of its variance).
// 0 maps to ~1e-150, which is 0 for the Matern factors.
__device__ __forceinline__ void sqrt4(const double (&xin)[4], double (&out)[4]) {
  double x[4], y[4], g[4], r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = fmax(xin[q], 1e-300);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) y[q] = __builtin_amdgcn_rsq(x[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    g[q] = x[q] * y[q];
    y[q] = 0.5 * y[q];
  }
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) r[q] = fma(-g[q], g[q], x[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = fma(r[q], y[q], g[q]);
  SGP_FENCE();
}

// The same for strictly positive arguments (no clamp).
template <int NV>
__device__ __forceinline__ void sqrtn_pos(const double (&x)[NV], double (&out)[NV]) {
  double y[NV], g[NV], r[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) y[q] = __builtin_amdgcn_rsq(x[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    g[q] = x[q] * y[q];
    y[q] = 0.5 * y[q];
  }
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) r[q] = fma(-g[q], g[q], x[q]);
  SGP_FENCE();
#pragma unroll
  for (int q = 0; q < NV; ++q) out[q] = fma(r[q], y[q], g[q]);
  SGP_FENCE();
}
__device__ __forceinline__ void sqrt4_pos(const double (&x)[4], double (&out)[4]) {
  sqrtn_pos<4>(x, out);
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
// (one stationary part) works on inputs pre-multiplied by KernDesc::scale0 and
// keeps everything in registers; products of parts fall back to the descriptor
// loop on raw inputs.
template <int D>
struct KernFast {
  const KernDesc* kd;
  bool single;
  int kind0;
  double var0;
  double m1, m2;   // Matern polynomial in u = -|dz|: var0 + u (m1 + u m2)
  double sc[D];

  // wave-uniform hyper-parameters, pinned to SGPRs (the sweep runs at the
  // 256-VGPR limit)
  static __device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
  }

  __device__ __forceinline__ explicit KernFast(const KernDesc& k) : kd(&k) {
    single = __builtin_amdgcn_readfirstlane(int(k.n_parts == 1)) != 0;
    kind0 = __builtin_amdgcn_readfirstlane(k.kind[0]);
    var0 = uni(k.variance[0]);
    // 1 + a (+ a^2/3), a = -u ln2/32, times var0
    m1 = uni(-var0 * SGP_E1);
    m2 = uni((kind0 == SGP_MATERN52) ? var0 * 0.00015639746546816451 : 0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) sc[i] = uni(k.scale0[i]);
  }

  // The same from a descriptor in CONSTANT address space (read-only for the whole
  // launch): every field is a scalar load -- no vector load + readfirstlane, and
  // no vmcnt wait that would sit out the LDS-DMA traffic in flight.
  typedef const __attribute__((address_space(4))) KernDesc* const_desc_t;
  __device__ __forceinline__ KernFast() : kd(nullptr) {}
  __device__ __forceinline__ void load_const(const KernDesc* generic) {
    kd = generic;
    const const_desc_t k = (const_desc_t)generic;
    single = k->n_parts == 1;
    kind0 = k->kind[0];
    var0 = k->variance[0];
    m1 = uni(-var0 * SGP_E1);
    m2 = uni((kind0 == SGP_MATERN52) ? var0 * 0.00015639746546816451 : 0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) sc[i] = k->scale0[i];
  }

  // candidate row -> the form operator() expects (scaled when `single`)
  __device__ __forceinline__ void prep(const double* x, double* xs) const {
#pragma unroll
    for (int i = 0; i < D; ++i) xs[i] = single ? x[i] * sc[i] : x[i];
  }

  // covariance from the squared distance in scaled units
  __device__ __forceinline__ double of_r2s(double r2, const double* tab) const {
    if (kind0 == SGP_RBF) return var0 * exp2_32(-r2, tab);
    const double u = -sqrt(r2);
    return fma(u, fma(u, m2, m1), var0) * exp2_32(u, tab);
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
      return of_r2s(r2, tab);
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
      double r2[4], u[4], e[4];
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
        for (int q = 0; q < 4; ++q) u[q] = -r2[q];
        exp2_32x4(u, tab, e);
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = var0 * e[q];
      } else {
        double rr[4];
        sqrt4(r2, rr);
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = -rr[q];
        exp2_32x4(u, tab, e);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          out[q] = fma(u[q], fma(u[q], m2, m1), var0) * e[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
        out[q] = kern_eval<D>(*kd, xs, ys + q * stride);
    }
  }

  // The sweep's variants with the single-part test resolved at compile time
  // (no generic product-kernel code, registers or branches in that instance).
  template <bool SINGLE>
  __device__ __forceinline__ void prep_t(const double* x, double* xs) const {
#pragma unroll
    for (int i = 0; i < D; ++i) xs[i] = (SINGLE || single) ? x[i] * sc[i] : x[i];
  }

  template <int NV, bool SINGLE>
  __device__ __forceinline__ void manyn_t(const double* xs, const double* ys,
                                          int stride, const double* tab,
                                          double (&out)[NV]) const {
    if (SINGLE) {
      double r2[NV], u[NV], e[NV], y[NV][D];
      // all the training rows first (one LDS round trip, not NV)
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int i = 0; i < D; ++i) y[q][i] = ys[q * stride + i];
      SGP_FENCE();
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        // 1e-300 instead of 0: the square root below needs no clamp, and it is
        // far below the last bit of any distance that matters
        r2[q] = 1e-300;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const double t = xs[i] - y[q][i];
          r2[q] = fma(t, t, r2[q]);
        }
      }
      if (kind0 == SGP_RBF) {
#pragma unroll
        for (int q = 0; q < NV; ++q) u[q] = -r2[q];
        exp2_32xn<NV>(u, tab, e);
#pragma unroll
        for (int q = 0; q < NV; ++q) out[q] = var0 * e[q];
      } else {
        double rr[NV];
        sqrtn_pos<NV>(r2, rr);
#pragma unroll
        for (int q = 0; q < NV; ++q) u[q] = -rr[q];
        exp2_32xn<NV>(u, tab, e);
#pragma unroll
        for (int q = 0; q < NV; ++q)
          out[q] = fma(u[q], fma(u[q], m2, m1), var0) * e[q];
      }
    } else {
      // (two values at a time: the 4-wave sweep runs at the register limit; d >= 7:
      // one -- the 8 squared differences of a value are 16 registers)
      constexpr int kStep = NV > 2 ? (D >= 7 ? 1 : 2) : NV;
#pragma unroll
      for (int q0 = 0; q0 < NV; q0 += kStep) {
        double o[kStep];
        product_n<kStep>(xs, ys + q0 * stride, stride, tab, o);
#pragma unroll
        for (int q = 0; q < kStep; ++q) out[q0 + q] = o[q];
      }
    }
  }

  // Product of stationary parts, NV values: with the weights of KernDesc::wsq the
  // part p contributes u_p = -sum_k wsq[p][k] dx_k^2 (RBF) or minus its square root
  // (Matern) in units where the part is poly_p(u_p) 2^(u_p / 32): the exponents add,
  // ONE 2^(U/32) per covariance, a square root and a polynomial factor per Matern
  // part.  The weights are wave-uniform scalar loads from the descriptor (constant
  // address space: no vector registers, no vmcnt wait); those of the first two parts
  // are fetched up front (a part the kernel does not have weighs zero).  GPy's Prod
  // kernel: K = prod_p K_p (GPy kern/src/prod.py).  xs, ys: RAW rows (products) or
  // the pre-scaled ones of a single-part GP.
  template <int NV>
  __device__ __forceinline__ void product_n(const double* xs, const double* ys,
                                            int stride, const double* tab,
                                            double (&out)[NV]) const {
    const const_desc_t k = (const_desc_t)kd;
    const int P = k->n_parts;
    // (a single-part GP in a launch that also has products: its training rows are
    // stored pre-scaled, GpDev::Xs, and prep_t scaled the candidate row -- weights 1)
    double w0[D], w1[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      w0[i] = single ? 1.0 : k->wsq[0][i];
      w1[i] = k->wsq[1][i];
    }
    const int kind0p = k->kind[0], kind1p = k->kind[1];
    const double vtot = k->kdiag;
    double d2[NV][D];
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double t = xs[i] - ys[q * stride + i];
        d2[q][i] = t * t;
      }
    double U[NV], F[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      U[q] = 0.0;
      F[q] = vtot;
    }
    auto part = [&](const double (&w)[D], int kind) {
      double r2[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        r2[q] = 1e-300;
#pragma unroll
        for (int i = 0; i < D; ++i) r2[q] = fma(w[i], d2[q][i], r2[q]);
      }
      if (kind == SGP_RBF) {
#pragma unroll
        for (int q = 0; q < NV; ++q) U[q] -= r2[q];
      } else {
        double rr[NV];
        sqrtn_pos<NV>(r2, rr);
        // 1 + a (+ a^2 / 3), a = -u ln2 / 32
        const double c2 = kind == SGP_MATERN52 ? 0.00015639746546816451 : 0.0;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const double u = -rr[q];
          U[q] += u;
          F[q] *= fma(u, fma(u, c2, -SGP_E1), 1.0);
        }
      }
    };
    part(w0, kind0p);
    if (P > 1) part(w1, kind1p);
    for (int p = 2; p < P; ++p) {        // (wave-uniform, rare)
      double w[D];
#pragma unroll
      for (int i = 0; i < D; ++i) w[i] = k->wsq[p][i];
      part(w, k->kind[p]);
    }
    double e[NV];
    exp2_32xn<NV>(U, tab, e);
#pragma unroll
    for (int q = 0; q < NV; ++q) out[q] = F[q] * e[q];
  }

  template <bool SINGLE>
  __device__ __forceinline__ void many4_t(const double* xs, const double* ys,
                                          int stride, const double* tab,
                                          double (&out)[4]) const {
    manyn_t<4, SINGLE>(xs, ys, stride, tab, out);
  }

  // ... with the single-part test resolved at compile time
  template <bool SINGLE>
  __device__ __forceinline__ double raw_t(const double* x, const double* y,
                                          const double* tab) const {
    if (SINGLE) {
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double t = (x[i] - y[i]) * sc[i];
        r2 = fma(t, t, r2);
      }
      return of_r2s(r2, tab);
    }
    return kern_eval<D>(*kd, x, y);
  }

  // both arguments raw (unscaled) rows
  __device__ __forceinline__ double raw(const double* x, const double* y,
                                        const double* tab) const {
    if (single) {
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double t = (x[i] - y[i]) * sc[i];
        r2 = fma(t, t, r2);
      }
      return of_r2s(r2, tab);
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
