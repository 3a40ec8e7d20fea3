// One candidate row through the posterior of every GP of a launch, on the fp64 VALU,
// for GPs with at most NP <= 48 observations (see sweep_tiny.hip for the why):
//   v = L^-1 k(X, x),  var = k(x,x) - |v|^2,  mean = alpha . k(X, x),
//   Q row = mean -+ beta sqrt(var), safe = all(l_i > fmin_i)
//   (gp.predict_noiseless + update_confidence_intervals + compute_safe_set,
//    safeopt/gp_opt.py:453-481)
// Shared by k_sweep_tiny (sweep_tiny.hip: one thread per row of a large grid) and
// k_step_small (step_small.hip: a whole SafeOpt.optimize() of a small grid in one
// workgroup): ONE copy of the arithmetic, the same bits in both.
#pragma once

#include "kern_eval.h"
#include "sweep_shared.h"

typedef const __attribute__((address_space(4))) double* cdbl_t;
typedef const __attribute__((address_space(4))) GpDev* gpdev_c_t;

// One GP of the row.  XP / LP: where the wave-uniform operands come from -- pointers into
// constant address space (scalar loads: k_sweep_tiny, whose many waves per SIMD hide their
// latency) or into LDS (k_step_small: one pass of one workgroup, nothing to hide a chain
// of scalar-load round trips behind).  The arithmetic and its order are the same.
template <int D, int NP, bool SINGLE, typename XP, typename LP>
__device__ __forceinline__ void tiny_row_gp(const GpDev* gps, int g, int G, int n, XP X, XP al,
                                            LP Li, int64_t ld, const ConfOut& conf, int64_t N,
                                            const double (&x)[D], int64_t row, bool valid,
                                            const double* tab, bool& safe, double& l0) {
  const gpdev_c_t gpc = (gpdev_c_t)(gps);
  KernFast<D> kf;
  kf.load_const(&gps[g].kern);
  double xs[D];
  kf.template prep_t<SINGLE>(x, xs);
  // the n covariances of this row, four at a time (training rows: scalar loads)
  double k[NP];
  double mean = 0.0;
#pragma unroll
  for (int j0 = 0; j0 < NP; j0 += 4) {
    if (j0 < n) {                      // (uniform)
      double y[4][D];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < D; ++i) y[q][i] = (j0 + q < n) ? X[(j0 + q) * D + i] : 0.0;
      double kv[4];
      kf.template manyn_t<4, SINGLE>(xs, &y[0][0], D, tab, kv);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        k[j0 + q] = (j0 + q < n) ? kv[q] : 0.0;
        mean = fma(al[j0 + q], k[j0 + q], mean);      // (alpha: zero padded to 16)
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) k[j0 + q] = 0.0;
    }
  }
  // |L^-1 k|^2 (the entries of L^-1 are scalar operands), four rows at a time: four
  // independent chains, so that a launch with few rows (a swarm of 20 particles: one
  // wave, nothing else to hide the FMA latency behind) is not a single dependent chain
  // of n^2 / 2 instructions.  Every row is summed in the order j = 0 .. i.
  double ssq = 0.0;
#pragma unroll
  for (int i0 = 0; i0 < NP; i0 += 4) {
    if (i0 < n) {                      // (uniform)
      double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j <= i0 + 3; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j <= i0 + q) v[q] = fma(Li[int64_t(i0 + q) * ld + j], k[j], v[q]);
      }
      // (rows n .. of the last group are NOT summed: after a pop / in a buffer with room
      // for appends they hold whatever the factor left there -- only their reads are
      // harmless)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + q < n) ssq = fma(v[q], v[q], ssq);
    }
  }
  {
    // (no contraction: mu -+ beta sd is rounded as the reference rounds it -- multiply,
    // then add)
#pragma clang fp contract(off)
    const double var = fmax(gpc[g].kern.kdiag - ssq, 1e-15);   // GPy clip
    const double sd = sqrt(var);
    const double lo = mean - conf.beta * sd;
    const double up = mean + conf.beta * sd;
    if (g == 0) l0 = lo;
    safe = safe && (lo > conf.fmin[g]);
    if (valid) {
      __builtin_nontemporal_store(mean, conf.mean + int64_t(g) * N + row);
      __builtin_nontemporal_store(var, conf.var + int64_t(g) * N + row);
      if (conf.Q)
        *reinterpret_cast<double2_t*>(conf.Q + (row * G + g) * 2) = double2_t{lo, up};
    }
  }
}

template <int D, int NP, bool SINGLE>
__device__ __forceinline__ void tiny_row(const GpDev* gps, int G, const ConfOut& conf,
                                         int64_t N, const double (&x)[D], int64_t row,
                                         bool valid, const double* tab, bool& safe,
                                         double& l0) {
  const gpdev_c_t gpc = (gpdev_c_t)(gps);
  for (int g = 0; g < G; ++g)          // (wave-uniform)
    tiny_row_gp<D, NP, SINGLE>(gps, g, G, gpc[g].n, (cdbl_t)(gpc[g].Xs), (cdbl_t)(gpc[g].alpha),
                               (cdbl_t)(gpc[g].Linv), gpc[g].ld, conf, N, x, row, valid, tab,
                               safe, l0);
}
