// One SafeOpt.optimize() of a SMALL grid in ONE launch of ONE workgroup: the regime of
// the reference's own examples and tests (safeopt/gp_opt.py:651-675 on a 1000-point or
// 100 x 100 grid with n <= 20 observations: examples/1d_example.ipynb cells 2 / 6,
// safeopt/tests/test_gps.py:21-24 -- BASELINE.json config 1).
//
// There the step of the large-grid path -- the posterior sweep plus nine set-pass
// launches, each a chip-wide reduction the next one depends on -- costs 90-120 us
// whatever the grid (profiles/r04/experiments.txt section 15): ~7 us per dependent
// launch, and the kernels themselves are 2-4 us each.  For up to kStepRows rows and GPs
// with at most 48 observations the whole chain fits one workgroup of 512 threads, with
// a workgroup barrier where the large-grid path has a kernel boundary:
//   A  posterior of every row and GP, Q, S          (tiny_row.h: the arithmetic of
//                                                     k_sweep_tiny, the same bits)
//   B  max l0[S]                                     gp_opt.py:511
//   C  M = S & (u0 >= max l0[S]), max width over M   gp_opt.py:511-513
//   D  candidate mask, widths, counts, the first candidate in visiting order and the
//      number of candidates tied with it             gp_opt.py:519-552
//   E  rank-1 expander test of that candidate over ALL unsafe rows (exact, no probe):
//      t = L^-1 k_c, w = L^-T t, s^2, c(x) = k(x, x_c) - w . k(X, x), l_2(x) >= fmin
//      (sweep.hip:k_expander_list, factor.hip:k_expkt / k_expw1 -- the same summation
//      orders, so the flags are those of the large-grid path)      gp_opt.py:579-606
//   F  G mark when every active GP certified it, arg-max over M | G   gp_opt.py:611-649
// and one result block in the layout of sgp_grid_sets_fused comes back.  Everything the
// uncommon branches of the host driver read afterwards (Q, mean, var, S, M, G, the
// candidate mask and widths, max l0[S]) is resident exactly as the large-grid path
// leaves it.
#include <algorithm>

#include "set_order.h"
#include "tiny_row.h"

namespace {

// Workgroup size (template parameter TH): 512 threads -- two waves per SIMD, 256 registers
// each: a row's 48 covariances and the four FMA chains of tiny_row_gp need up to ~190.
// (1024 threads with 128 registers for n <= 24 spill 64-320 B and were no faster: the step
// is a chain of latencies, not of issue slots -- profiles/r05/experiments.txt.)
constexpr int kExpwWaves = 16;    // row groups of k_expw1 (factor.hip), whose order is kept
constexpr int kStepNP = 48;       // observations per GP, at most (tiny_row)

struct StepParams {
  const GpDev* gps;
  int G;
  SweepPoints pts;
  ConfOut conf;            // Q, mean, var, S; beta, fmin
  uint8_t* M;
  uint8_t* Gm;
  uint8_t* cand;
  double* w;
  int64_t goff;
  Vec8 scaling, thr_beta;
  double* res;             // result block (layout of sgp_grid_sets_fused)
  double* scal;            // [0] = max l0[S]
  int nfront, nfl;
  unsigned long long seq;  // written behind the results (system scope): the host spins on it
};

// -DSTEP_STAMPS: s_memtime at the phase boundaries into words 40.. of the result block
// (timing experiments: scripts/dev/small_step_time.py prints them)
#ifdef STEP_STAMPS
#define STEP_STAMP(i)                                                                   \
  do {                                                                                  \
    if (threadIdx.x == 0)                                                               \
      reinterpret_cast<unsigned long long*>(p.res)[40 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define STEP_STAMP(i) do {} while (0)
#endif

template <int D, int NP, bool SINGLE, int TH>
__global__ __launch_bounds__(TH) void k_step_small(StepParams p) {
  constexpr int kStepThreads = TH;
  constexpr int kStepWaves = TH / 64;
  __shared__ double tab[kExpTabSize];
  __shared__ double shd[kStepWaves];
  __shared__ Pair shp[kStepWaves];
  __shared__ unsigned shc[3][kStepWaves];
  __shared__ double sh_w[kExpwWaves][64];
  __shared__ double kc[64], tt[64], wv[64], xcs[SGP_MAX_D];
  // the GP being worked on, staged: dense L^-1 (rows < n, pitch NP), its scaled and raw
  // training rows (n_pad <= 64 of them), alpha.  One pass of ONE workgroup has nothing to
  // hide a chain of scalar- or vector-load round trips behind, so the wave-uniform operands
  // of tiny_row_gp and of the expander test come from LDS here (same values, same order).
  __shared__ double gLi[NP * NP], gX[64 * D], gXp[64 * D], gAl[NP];
  __shared__ double s_ops[3];       // delta, 1 / s^2 of the GP being scanned
  __shared__ int s_flags[SGP_MAX_GPS];
  constexpr int kListCap = 2048;    // rows the pre-filter of the expander test may pass on
  __shared__ int s_list[kListCap];
  __shared__ int s_cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t N = p.pts.N;
  const int G = p.G;
  exp_tab_init(tab);
  if (tid < SGP_MAX_GPS) s_flags[tid] = 0;
  __syncthreads();

  const gpdev_c_t gpc = (gpdev_c_t)(p.gps);
  int staged = -1;
  auto stage_gp = [&](int g) {            // (workgroup-uniform; syncs)
    if (staged == g) return;
    __syncthreads();                      // the readers of the GP staged before
    const GpDev& gp = p.gps[g];
    const int n = gpc[g].n, np = gpc[g].n_pad;
    const int64_t ld = gpc[g].ld;
    for (int e = tid; e < NP * NP; e += kStepThreads) {
      const int i = e / NP, j = e - i * NP;
      gLi[e] = (i < n && j <= i) ? gp.Linv[int64_t(i) * ld + j] : 0.0;
    }
    for (int e = tid; e < 64 * D; e += kStepThreads) {
      gX[e] = e < np * D ? gp.Xs[e] : 0.0;
      gXp[e] = e < np * D ? gp.Xpad[e] : 0.0;
    }
    for (int e = tid; e < NP; e += kStepThreads) gAl[e] = e < np ? gp.alpha[e] : 0.0;
    __syncthreads();
    staged = g;
  };

  STEP_STAMP(0);
  // ---- A: posterior, intervals, S; B: max l0 over the safe rows.  GP by GP (its operands
  // staged), GP 0 last: its lower bound then meets the row's final `safe`.
  double lmax = -INFINITY;
  uint32_t safe_bits = 0xffffffffu;       // bit r: this thread's r-th row (<= 32 of them)
  const int rounds = int((N + kStepThreads - 1) / kStepThreads);
  for (int gi = 0; gi < G; ++gi) {
    const int g = (gi + 1 < G) ? gi + 1 : 0;
    stage_gp(g);
    const int n = gpc[g].n;
    for (int r = 0; r < rounds; ++r) {
      const int64_t row = int64_t(r) * kStepThreads + tid;
      const bool valid = row < N;
      const int64_t rr = valid ? row : N - 1;
      double x[D];
#pragma unroll
      for (int k = 0; k < D; ++k)
        x[k] = p.pts.base[rr * p.pts.stride_row + k * p.pts.stride_col];
      bool safe = ((safe_bits >> r) & 1u) != 0;
      double l0 = 0.0;
      tiny_row_gp<D, NP, SINGLE>(p.gps, g, G, n, (const double*)gX, (const double*)gAl,
                                 (const double*)gLi, int64_t(NP), p.conf, N, x, row, valid, tab,
                                 safe, l0);
      safe_bits = safe ? (safe_bits | (1u << r)) : (safe_bits & ~(1u << r));
      if (g == 0 && valid) {
        p.conf.S[row] = safe ? 1 : 0;
        if (safe) lmax = fmax(lmax, l0);
      }
    }
  }
  STEP_STAMP(1);
  const double max_l = block_max(lmax, shd);        // (syncs: the rows are visible)
  __syncthreads();
  STEP_STAMP(2);

  // ---- C: maximisers.  One read of S and of the intervals serves C and D for the first
  // kKeep rows of a thread (a 1000-point grid: all of them): every further pass over the
  // rows is another global round trip on the critical path of a single workgroup.
  constexpr int kKeep = 2;
  struct RowKeep {
    bool sf, above;
    double u0, w0, wmax, smax;
  };
  RowKeep keep[kKeep];
  auto read_row = [&](int64_t i, RowKeep& k) {
    k.sf = p.conf.S[i] != 0;
    k.wmax = k.smax = -INFINITY;
    k.above = false;
    for (int g = 0; g < G; ++g) {
      const double2_t q = *reinterpret_cast<const double2_t*>(p.conf.Q + (i * G + g) * 2);
      const double width = q.y - q.x;
      if (g == 0) {
        k.u0 = q.y;
        k.w0 = width;
      }
      k.wmax = fmax(k.wmax, width);
      k.smax = fmax(k.smax, width / p.scaling.v[g]);
      k.above = k.above || (width > p.thr_beta.v[g]);
    }
  };
  double v = -INFINITY;
#pragma unroll
  for (int r = 0; r < kKeep; ++r) {
    const int64_t i = int64_t(r) * kStepThreads + tid;
    if (i < N) {
      read_row(i, keep[r]);
      const bool m = keep[r].sf && (keep[r].u0 >= max_l);
      p.M[i] = m ? 1 : 0;
      if (m) v = fmax(v, keep[r].w0);
    }
  }
  for (int64_t i = int64_t(kKeep) * kStepThreads + tid; i < N; i += kStepThreads) {
    const double2_t q = *reinterpret_cast<const double2_t*>(p.conf.Q + i * G * 2);
    const bool m = p.conf.S[i] && (q.y >= max_l);
    p.M[i] = m ? 1 : 0;
    if (m) v = fmax(v, q.y - q.x);
  }
  const double mw = block_max(v, shd);
  const double max_var = mw / p.scaling.v[0];

  STEP_STAMP(3);
  // ---- D: candidates, their widths, counts, the first one in visiting order
  unsigned nc = 0, nu = 0;
  Pair best{-INFINITY, -1};
  bool ckeep[kKeep];
  auto candidate = [&](int64_t i, const RowKeep& k) {
    const bool mv = k.sf && (k.u0 >= max_l);          // (= M[i])
    const bool c = k.sf && !mv && (k.smax > max_var) && k.above;
    if (!k.sf) ++nu;
    p.cand[i] = c ? 1 : 0;
    p.w[i] = k.sf ? k.wmax : -INFINITY;
    p.Gm[i] = 0;
    if (c) {
      ++nc;
      const Pair pr{k.wmax, p.goff + i};
      if (best.i < 0 || before_desc(pr, best)) best = pr;
    }
    return c;
  };
#pragma unroll
  for (int r = 0; r < kKeep; ++r) {
    const int64_t i = int64_t(r) * kStepThreads + tid;
    ckeep[r] = (i < N) && candidate(i, keep[r]);
  }
  for (int64_t i = int64_t(kKeep) * kStepThreads + tid; i < N; i += kStepThreads) {
    RowKeep k;
    read_row(i, k);
    candidate(i, k);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    nc += __shfl_xor(nc, o, 64);
    nu += __shfl_xor(nu, o, 64);
  }
  if (lane == 0) {
    shc[0][wave] = nc;
    shc[1][wave] = nu;
  }
  const Pair win = block_best<false>(best, shp);     // (syncs)
  unsigned ncand = 0, nunsafe = 0;
  for (int wv_ = 0; wv_ < kStepWaves; ++wv_) {
    ncand += shc[0][wv_];
    nunsafe += shc[1][wv_];
  }
  // candidates that share the first one's width, bit for bit (gp_opt.py:542-552: NumPy's
  // sort decides among them -- the host settles the order when there is more than one)
  unsigned nt = 0;
  if (win.i >= 0) {
#pragma unroll
    for (int r = 0; r < kKeep; ++r)
      if (ckeep[r] && keep[r].wmax == win.v) ++nt;
    for (int64_t i = int64_t(kKeep) * kStepThreads + tid; i < N; i += kStepThreads)
      if (p.cand[i] && p.w[i] == win.v) ++nt;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nt += __shfl_xor(nt, o, 64);
  __syncthreads();
  if (lane == 0) shc[2][wave] = nt;
  __syncthreads();
  unsigned ntied = 0;
  for (int wv_ = 0; wv_ < kStepWaves; ++wv_) ntied += shc[2][wv_];

  const int d = D;
  const int64_t li = win.i - p.goff;
  if (tid == 0) {
    p.scal[0] = max_l;
    p.res[0] = mw;
    reinterpret_cast<unsigned long long*>(p.res)[1] = ncand;
    reinterpret_cast<unsigned long long*>(p.res)[2] = nunsafe;
    p.res[3] = win.v;
    reinterpret_cast<int64_t*>(p.res)[4] = win.i;
    reinterpret_cast<int*>(p.res + 5)[0] = win.i >= 0 ? 1 : 0;
    reinterpret_cast<int*>(p.res + 5)[1] = win.i >= 0 ? int(ntied) : 0;
    p.res[p.nfront + p.nfl + 2] = max_l;
  }
  if (win.i >= 0) {
    if (tid < d) {
      const double xv = p.pts.base[li * p.pts.stride_row + tid * p.pts.stride_col];
      p.res[6 + tid] = xv;
      xcs[tid] = xv;
    }
    if (tid >= 64 && tid < 64 + G) p.res[6 + d + (tid - 64)] = p.conf.mean[int64_t(tid - 64) * N + li];
    if (tid >= 128 && tid < 128 + 2 * G)
      p.res[6 + d + G + (tid - 128)] = p.conf.Q[li * 2 * G + (tid - 128)];
  }
  __syncthreads();

  STEP_STAMP(4);
  // ---- E: is the first candidate an expander?  (gp_opt.py:579-606 with the rank-1
  // update of the posterior instead of two refits per GP)
  if (win.i >= 0 && nunsafe > 0) {
    for (int g = 0; g < G; ++g) {          // (workgroup-uniform)
      if (p.conf.fmin[g] == -INFINITY) continue;
      const GpDev& gp = p.gps[g];
      const int n = gpc[g].n, np = gpc[g].n_pad;
      constexpr int64_t ld = NP;            // (the staged copy)
      stage_gp(g);
      // k_c = k(X, x_c) (the generic evaluation of k_expkt), t = L^-1 k_c
      if (tid < 64) {
        double kcv = 0.0;
        if (tid < n) {
          double xj[D], xc[D];
#pragma unroll
          for (int k = 0; k < D; ++k) {
            xj[k] = gXp[tid * D + k];
            xc[k] = xcs[k];
          }
          kcv = kern_eval<D>(gp.kern, xc, xj);
        }
        kc[tid] = kcv;
        tt[tid] = 0.0;
      }
      __syncthreads();
      for (int i = wave; i < n; i += kStepWaves) {
        double acc = 0.0;
        if (lane <= i) acc = fma(gLi[int64_t(i) * ld + lane], kc[lane], acc);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) tt[i] = acc;
      }
      __syncthreads();
      // w = L^-T t (k_expw1: 16 row groups i = q, q + 16, ..., summed over the groups in
      // order), |t|^2, s^2 = k(x_c, x_c) + noise - |t|^2
      for (int wq = wave; wq < kExpwWaves; wq += kStepWaves) {
        double a0 = 0.0;
        if (lane < n)
          for (int i = wq; i < n; i += kExpwWaves)
            if (i >= lane) a0 = fma(gLi[int64_t(i) * ld + lane], tt[i], a0);
        sh_w[wq][lane] = a0;
      }
      __syncthreads();
      if (wave == 0) {
        double tot = 0.0;
#pragma unroll
        for (int wq = 0; wq < kExpwWaves; ++wq) tot += sh_w[wq][lane];
        wv[lane] = (lane < n) ? tot : 0.0;
      }
      if (wave == 1) {
        double s = 0.0;
        if (lane < n) s = fma(tt[lane], tt[lane], s);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
          const double s2 = gp.prior - s;
          const double resid = p.conf.Q[li * 2 * G + 2 * g + 1] - p.conf.mean[int64_t(g) * N + li];
          s_ops[0] = resid / s2;      // delta
          s_ops[1] = 1.0 / s2;
          s_ops[2] = s;               // |t|^2
        }
      }
      __syncthreads();
      STEP_STAMP(5);
      const double delta = s_ops[0], inv_s2 = s_ops[1];
      // scan: 16 rows per wave (lane & 15), the four 16-lane groups split the training
      // points (k_expander_list); a row counts when it is unsafe
      const KernFast<D> kf(gp.kern);
      const double kdiag = gp.kern.kdiag;
      const int ph = lane >> 4;
      // exact update of one row (all four 16-lane groups call it with the same row in the
      // same lane & 15: they split the training points and fold -- k_expander_list)
      auto scan_row = [&](int64_t rrow) -> bool {
        double x[D], xs[D], xc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          x[k] = p.pts.base[rrow * p.pts.stride_row + k * p.pts.stride_col];
          xc[k] = xcs[k];
        }
        kf.prep(x, xs);
        const double mu = p.conf.mean[int64_t(g) * N + rrow];
        const double var = p.conf.var[int64_t(g) * N + rrow];
        double dot = 0.0;
#pragma unroll 1
        for (int s0 = 0; s0 < (np >> 2); s0 += 4) {   // 16 training points / step
          double kq[4], wq[4];
          kf.template many<4>(xs, gX + (s0 * 4 + ph) * D, 4 * D, tab, kq);
#pragma unroll
          for (int q = 0; q < 4; ++q) wq[q] = wv[(s0 + q) * 4 + ph];
#pragma unroll
          for (int q = 0; q < 4; ++q) dot = fma(wq[q], kq[q], dot);
        }
        dot = sum_lane_groups(dot);
        const double kxc = kf.raw(x, xc, tab);
        const double cx = kxc - dot;
        const double mu2 = mu + cx * delta;
        const double var2 = fmax(var - cx * cx * inv_s2, 1e-15);
        return mu2 - p.conf.beta * sqrt(var2) >= p.conf.fmin[g];
      };
      // Pre-filter (k_expander_filter, sweep.hip): |c(x)| <= |k(x, x_c)| + |L^-1 k_x| |t| with
      // |L^-1 k_x|^2 = k(x,x) - var(x) resident -- ONE covariance per unsafe row instead of
      // n decides for the rows that cannot be lifted to fmin whatever c(x) is; the others
      // (a few per cent) go on a list in LDS and get the exact n-term dot product below.
      // A conservative bound: the flags are those of the scan over all unsafe rows.  This
      // is what keeps the test off the critical path of ONE compute unit (1000 rows x 32
      // covariances were 10 of the step's 37 us).
      const double tn2 = s_ops[2];
      if (tid == 0) s_cnt = 0;
      __syncthreads();
      for (int64_t row = tid; row < N; row += kStepThreads) {
        if (p.conf.S[row] != 0) continue;
        double x[D], xc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          x[k] = p.pts.base[row * p.pts.stride_row + k * p.pts.stride_col];
          xc[k] = xcs[k];
        }
        const double mu = p.conf.mean[int64_t(g) * N + row];
        const double var = p.conf.var[int64_t(g) * N + row];
        const double qx = fmax(kdiag - var, 0.0);
        const double kxc = kf.raw(x, xc, tab);
        const double cmax = (fabs(kxc) + sqrt(qx * tn2)) * (1.0 + 1e-9);
        const double mu2 = mu + fabs(delta) * cmax;
        const double var2 = fmax(var - cmax * cmax * inv_s2, 1e-15);
        const double l2max = mu2 - p.conf.beta * sqrt(var2);
        if (l2max + 1e-9 * (fabs(mu2) + 1.0) >= p.conf.fmin[g]) {
          const int at = atomicAdd(&s_cnt, 1);
          if (at < kListCap) s_list[at] = int(row);
        }
      }
      __syncthreads();
      const int nlist = s_cnt;
      bool hit = false;
      if (nlist > kListCap) {
        // (more rows than the list holds: scan them all -- rare, and still exact)
        for (int64_t i0 = int64_t(wave) * 16; i0 < N; i0 += kStepWaves * 16) {
          const int64_t row = i0 + (lane & 15);
          const bool valid = row < N;
          const int64_t rrow = valid ? row : N - 1;
          const bool unsafe = valid && p.conf.S[rrow] == 0;
          if (__ballot(unsafe) == 0ull) continue;
          hit = hit || (unsafe && scan_row(rrow));
        }
      } else {
        for (int i0 = wave * 16; i0 < nlist; i0 += kStepWaves * 16) {
          const bool valid = i0 + (lane & 15) < nlist;
          const int64_t rrow = s_list[valid ? i0 + (lane & 15) : i0];
          hit = hit || (scan_row(rrow) && valid);
        }
      }
      if (__ballot(hit) != 0ull && lane == 0) atomicOr(&s_flags[g], 1);
      __syncthreads();
    }
  }
  __syncthreads();

  STEP_STAMP(6);
  // ---- F: G mark, arg-max over M | G of max_i (u_i - l_i) / scaling_i (first index)
  int64_t lmark = -1;
  if (win.i >= 0) {
    bool ok = true, any = false;
    for (int g = 0; g < G; ++g) {
      if (p.conf.fmin[g] == -INFINITY) continue;
      any = true;
      ok = ok && (s_flags[g] != 0);
    }
    if (ok && any) lmark = li;
  }
  Pair bm{-INFINITY, -1};
  for (int64_t i = tid; i < N; i += kStepThreads) {
    const bool marked = i == lmark;
    if (marked) p.Gm[i] = 1;
    if (!(p.M[i] || marked)) continue;
    double vv = -INFINITY;
    for (int g = 0; g < G; ++g)
      vv = fmax(vv, (p.conf.Q[(i * G + g) * 2 + 1] - p.conf.Q[(i * G + g) * 2]) / p.scaling.v[g]);
    const Pair pr{vv, p.goff + i};
    if (before_first(pr, bm)) bm = pr;
  }
  const Pair top = block_best<true>(bm, shp);
  if (tid == 0) {
    p.res[p.nfront + p.nfl] = top.v;
    reinterpret_cast<int64_t*>(p.res)[p.nfront + p.nfl + 1] = top.i;
  }
  if (tid < G) reinterpret_cast<int32_t*>(p.res + p.nfront)[tid] = s_flags[tid];
  // the block is host memory: everything above is out before the completion word
  __syncthreads();
  STEP_STAMP(7);
  if (tid == 0) {
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.res) + (kStepResWords - 1), p.seq,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int D, int NP, int TH>
void launch_step_np(sgp_ctx* ctx, const StepParams& p, bool single) {
  if (single)
    hipLaunchKernelGGL((k_step_small<D, NP, true, TH>), dim3(1), dim3(TH), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL((k_step_small<D, NP, false, TH>), dim3(1), dim3(TH), 0, ctx->stream, p);
}

template <int D>
void launch_step_d(sgp_ctx* ctx, const StepParams& p, int np, bool single) {
  if (np <= 24) return launch_step_np<D, 24, 512>(ctx, p, single);
  if (np <= 32) return launch_step_np<D, 32, 512>(ctx, p, single);
  return launch_step_np<D, 48, 512>(ctx, p, single);
}

// ---- every candidate of a small grid at once ------------------------------------------
// When the first candidate is no expander the reference walks through ALL candidates in
// visiting order until one is (gp_opt.py:557-612) -- and in a converged run none is, every
// iteration.  The large-grid path takes them 16 per device round trip (a scan of the unsafe
// rows per pass is what it can afford at 1e6 rows).  On a small grid the whole product
// candidates x unsafe rows is tiny: two launches test EVERY candidate of the list --
//   k_cand_ops:  one wave per candidate: t = L^-1 k_c, w = L^-T t, s^2, delta (rank-1
//                update of the posterior by the observation (x_c, u_i(x_c)))
//   k_cand_scan: one thread per unsafe row: its n covariances with the training points
//                once (registers), then for every candidate of the workgroup's chunk
//                c(x) = k(x, x_c) - w_c . k(X, x), l_2(x) >= fmin -> flags[c][g]
// and the host picks the first expander in the reference's own order (argsort()[::-1] of the
// candidate widths: exact ties included).
constexpr int kCandChunk = 32;          // candidates per workgroup of the scan
constexpr int kCandOps = 64 + 2;        // doubles per (candidate, GP): w[64] | delta | 1 / s^2

template <int D>
__global__ __launch_bounds__(64) void k_cand_ops(const GpDev* gps, int G, SweepPoints pts,
                                                 const double* Q, const double* mean, Vec8 fmin,
                                                 const int64_t* clist, int64_t goff, double* ops,
                                                 const int* count_dev) {
  __shared__ double kc[64], tt[64];
  const int lane = threadIdx.x, c = blockIdx.x;
  if (count_dev && c >= *count_dev) return;      // (the list was formed on the device: launched for its room)
  const gpdev_c_t gpc = (gpdev_c_t)(gps);
  const int64_t li = clist[c] - goff;
  double xc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) xc[k] = pts.base[li * pts.stride_row + k * pts.stride_col];
  for (int g = 0; g < G; ++g) {
    double* out = ops + (size_t(c) * G + g) * kCandOps;
    if (fmin.v[g] == -INFINITY) continue;
    const GpDev& gp = gps[g];
    const int n = gpc[g].n;
    const int64_t ld = gpc[g].ld;
    __syncthreads();
    double kcv = 0.0;
    if (lane < n) {
      double xj[D];
#pragma unroll
      for (int k = 0; k < D; ++k) xj[k] = gp.Xpad[int64_t(lane) * D + k];
      kcv = kern_eval<D>(gp.kern, xc, xj);
    }
    kc[lane] = kcv;
    __syncthreads();
    double t = 0.0;                       // t_i, i = lane
    if (lane < n)
      for (int j = 0; j <= lane; ++j) t = fma(gp.Linv[int64_t(lane) * ld + j], kc[j], t);
    tt[lane] = t;
    double s = lane < n ? t * t : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    __syncthreads();
    double w = 0.0;                       // w_j, j = lane
    if (lane < n)
      for (int i = lane; i < n; ++i) w = fma(gp.Linv[int64_t(i) * ld + lane], tt[i], w);
    out[lane] = w;
    if (lane == 0) {
      const double s2 = gp.prior - s;
      const double resid = Q[li * 2 * G + 2 * g + 1] - mean[int64_t(g) * pts.N + li];
      out[64] = resid / s2;
      out[65] = 1.0 / s2;
    }
  }
}

template <int D, int NP>
__global__ __launch_bounds__(256) void k_cand_scan(const GpDev* gps, int G, SweepPoints pts,
                                                   const uint8_t* S, const double* mean,
                                                   const double* var, double beta, Vec8 fmin,
                                                   const int64_t* clist, int m, int64_t goff,
                                                   const double* ops, int32_t* flags,
                                                   const int* count_dev) {
  __shared__ double tab[kExpTabSize];
  __shared__ double sw[kCandChunk][kCandOps];
  __shared__ double sxc[kCandChunk][D];
  __shared__ int shit[kCandChunk];
  const int c0 = blockIdx.y * kCandChunk;
  if (count_dev) m = min(m, *count_dev);
  if (c0 >= m) return;
  exp_tab_init(tab);
  const int tid = threadIdx.x;
  const int mc = min(kCandChunk, m - c0);
  const gpdev_c_t gpc = (gpdev_c_t)(gps);
  const int64_t row = int64_t(blockIdx.x) * 256 + tid;
  const bool valid = row < pts.N;
  const int64_t rr = valid ? row : pts.N - 1;
  const bool unsafe = valid && S[rr] == 0;
  for (int e = tid; e < mc * D; e += 256) {
    const int c = e / D, k = e - c * D;
    sxc[c][k] = pts.base[(clist[c0 + c] - goff) * pts.stride_row + k * pts.stride_col];
  }
  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k) x[k] = pts.base[rr * pts.stride_row + k * pts.stride_col];
  const bool any_unsafe = __syncthreads_or(unsafe ? 1 : 0) != 0;      // (also: tab, sxc)
  if (!any_unsafe) return;
  for (int g = 0; g < G; ++g) {
    if (fmin.v[g] == -INFINITY) continue;
    __syncthreads();
    for (int e = tid; e < mc * kCandOps; e += 256)
      sw[e / kCandOps][e % kCandOps] = ops[(size_t(c0 + e / kCandOps) * G + g) * kCandOps + e % kCandOps];
    if (tid < kCandChunk) shit[tid] = 0;
    __syncthreads();
    const GpDev& gp = gps[g];
    const int n = gpc[g].n;
    const KernFast<D> kf(gp.kern);
    double xs[D];
    kf.prep(x, xs);
    // this row's covariances with the training points, once for all candidates
    double kx[NP];
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += 4) {
      if (j0 < n) {
        double kv[4];
        kf.template many<4>(xs, gp.Xs + j0 * D, D, tab, kv);
#pragma unroll
        for (int q = 0; q < 4; ++q) kx[j0 + q] = (j0 + q < n) ? kv[q] : 0.0;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) kx[j0 + q] = 0.0;
      }
    }
    const double mu = mean[int64_t(g) * pts.N + rr];
    const double vr = var[int64_t(g) * pts.N + rr];
    for (int c = 0; c < mc; ++c) {
      double dot = 0.0;
#pragma unroll
      for (int j = 0; j < NP; ++j) dot = fma(sw[c][j], kx[j], dot);
      double xc[D];
#pragma unroll
      for (int k = 0; k < D; ++k) xc[k] = sxc[c][k];
      const double cx = kf.raw(x, xc, tab) - dot;
      const double mu2 = mu + cx * sw[c][64];
      const double var2 = fmax(vr - cx * cx * sw[c][65], 1e-15);
      const bool hit = unsafe && (mu2 - beta * sqrt(var2) >= fmin.v[g]);
      if (__ballot(hit) != 0ull && (tid & 63) == 0) shit[c] = 1;
    }
    __syncthreads();
    if (tid < mc && shit[tid]) atomicOr(&flags[(c0 + tid) * G + g], 1);
  }
}

}  // namespace

int launch_cand_all(sgp_grid* g, const GpDev* gps_dev, const GpDev* gh, int G, double beta,
                    const double* fmin, const int64_t* clist_dev, int m, double* ops,
                    int32_t* flags, const int* count_dev) {
  sgp_ctx* ctx = g->ctx;
  SweepPoints pts{g->pts, g->N, 1, g->N};
  Vec8 fm;
  int np = 1;
  for (int i = 0; i < SGP_MAX_GPS; ++i) fm.v[i] = i < G ? fmin[i] : -INFINITY;
  for (int i = 0; i < G; ++i) np = std::max(np, gh[i].n);
  const dim3 sg(unsigned((g->N + 255) / 256), unsigned((m + kCandChunk - 1) / kCandChunk));
#define CAND_CASE(DD)                                                                          \
  case DD:                                                                                     \
    hipLaunchKernelGGL(k_cand_ops<DD>, dim3(m), dim3(64), 0, ctx->stream, gps_dev, G, pts,     \
                       g->Q, g->mean, fm, clist_dev, g->goff, ops, count_dev);                 \
    if (np <= 24)                                                                              \
      hipLaunchKernelGGL((k_cand_scan<DD, 24>), sg, dim3(256), 0, ctx->stream, gps_dev, G,     \
                         pts, g->S, g->mean, g->var, beta, fm, clist_dev, m, g->goff, ops,     \
                         flags, count_dev);                                                    \
    else                                                                                       \
      hipLaunchKernelGGL((k_cand_scan<DD, 48>), sg, dim3(256), 0, ctx->stream, gps_dev, G,     \
                         pts, g->S, g->mean, g->var, beta, fm, clist_dev, m, g->goff, ops,     \
                         flags, count_dev);                                                    \
    break;
  switch (g->d) {
    CAND_CASE(1) CAND_CASE(2) CAND_CASE(3) CAND_CASE(4)
    CAND_CASE(5) CAND_CASE(6) CAND_CASE(7) CAND_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", g->d, SGP_MAX_D);
      return -2;
  }
#undef CAND_CASE
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// The candidates a device-side selection listed (local rows, count in sel[2] as PassSel has it):
// their global rows and widths next to each other for ONE read-back, flags zeroed.
__global__ __launch_bounds__(256) void k_small_pack(const int* list, const int* count_dev, int cap,
                                                    const double* w, int64_t goff, int G,
                                                    int64_t* hdr, int64_t* clist, double* wout,
                                                    int32_t* flags) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  const int count = *count_dev;
  if (pos == 0) hdr[0] = count;
  if (pos >= cap) return;
  const bool in = pos < count;
  const int li = in ? list[pos] : 0;
  clist[pos] = in ? goff + li : goff;
  wout[pos] = in ? w[li] : 0.0;
  for (int g = 0; g < G; ++g) flags[pos * G + g] = 0;
}

int launch_small_pack(sgp_grid* g, const int* list_dev, const int* count_dev, int cap,
                      int64_t* hdr, int64_t* clist, double* wout, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  hipLaunchKernelGGL(k_small_pack, dim3((cap + 255) / 256), dim3(256), 0, ctx->stream, list_dev,
                     count_dev, cap, g->w, g->goff, g->G, hdr, clist, wout, flags);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

size_t cand_ops_doubles(int m, int G) { return size_t(m) * G * kCandOps; }

bool step_small_eligible(const sgp_ctx* ctx, const GpDev* gh, int G, int64_t N) {
  static const bool off = getenv("SGP_NO_STEP_SMALL") != nullptr;
  if (off || N > kStepSmallRows || (ctx->sweep_choice & 3) != 0) return false;
  for (int g = 0; g < G; ++g)
    if (gh[g].n > kStepNP || gh[g].n_pad > 64) return false;
  return true;
}

int launch_step_small(sgp_grid* g, const GpDev* gps_dev, const GpDev* gh, int G, double beta,
                      const double* fmin, const double* scaling, const double* thr_beta,
                      double* res, int nfront, int nfl, uint64_t seq) {
  sgp_ctx* ctx = g->ctx;
  StepParams p{};
  p.gps = gps_dev;
  p.G = G;
  p.pts = SweepPoints{g->pts, g->N, 1, g->N};
  p.conf.Q = g->Q;
  p.conf.mean = g->mean;
  p.conf.var = g->var;
  p.conf.S = g->S;
  p.conf.partial = g->partial;
  p.conf.beta = beta;
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    p.conf.fmin[i] = i < G ? fmin[i] : -INFINITY;
    p.scaling.v[i] = i < G ? scaling[i] : 1.0;
    p.thr_beta.v[i] = i < G ? thr_beta[i] : 0.0;
  }
  p.M = g->M;
  p.Gm = g->Gm;
  p.cand = g->cand;
  p.w = g->w;
  p.goff = g->goff;
  p.res = res;
  p.scal = g->scal;
  p.nfront = nfront;
  p.nfl = nfl;
  p.seq = seq;
  int np = 1;
  bool single = true;
  for (int i = 0; i < G; ++i) {
    np = std::max(np, gh[i].n);
    single = single && gh[i].kern.n_parts == 1;
  }
  switch (g->d) {
    case 1: launch_step_d<1>(ctx, p, np, single); break;
    case 2: launch_step_d<2>(ctx, p, np, single); break;
    case 3: launch_step_d<3>(ctx, p, np, single); break;
    case 4: launch_step_d<4>(ctx, p, np, single); break;
    case 5: launch_step_d<5>(ctx, p, np, single); break;
    case 6: launch_step_d<6>(ctx, p, np, single); break;
    case 7: launch_step_d<7>(ctx, p, np, single); break;
    case 8: launch_step_d<8>(ctx, p, np, single); break;
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", g->d, SGP_MAX_D);
      return -2;
  }
  SGP_HIP(ctx, hipGetLastError());
  ctx->last_sweep = 5;
  return 0;
}
