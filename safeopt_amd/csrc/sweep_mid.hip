// posterior_sweep for 49 .. 128 observations: the matrix-core kernel between the VALU
// kernel (sweep_tiny.hip, n <= 48) and the stage-driven 4-wave kernel (sweep.hip).
//
// Same mathematics (gp.predict_noiseless + the Q update of
// SafeOpt.update_confidence_intervals, safeopt/gp_opt.py:453-481; the posterior half of
// SafeOptSwarm._compute_particle_fitness, :901-1013):
//     v = L^-1 k(X, x),  var = k(x,x) - |v|^2,  mean = alpha . k(X, x)
// on v_mfma_f64_4x4x4_4b_f64 with the operand maps of sweep.hip.  What the factor's
// size allows here and the general kernels cannot do:
//   * the whole L^-1 of every GP of the launch (lower-triangular 16 x 16 blocks in
//     A-operand order, 72 KB at n = 128), the training inputs and alpha stay in LDS for
//     the launch: one copy per workgroup at the start, then no staging, no LDS-DMA, no
//     stage table, no barrier -- the waves of a workgroup run free;
//   * the (j-block, row block) nest of a tile is straight-line code (NB = n_pad / 16 is
//     a template parameter): no slot guards, no computed entry, the compiler schedules
//     the evaluation of a j-block, the LDS transpose of its covariances and the matrix
//     instructions of the row blocks below it against each other;
//   * 32 .. 64 accumulator registers instead of 128: three waves per SIMD (12-wave
//     workgroups) -- a wave's row epilogue and its global loads pass under the matrix
//     instructions of two others.
// The 4-wave kernel spends 17 % of a tile at n = 64 on stage bookkeeping, 27 % on the
// tile epilogue and reaches 24 cycles per MFMA in its short slot runs with two waves
// per SIMD (profiles/r04/stamps_cfg2.txt): 0.24-0.31 of the fp64 matrix peak at n = 64,
// 0.38-0.44 at n = 128.
//
// Single-part kernels, d <= 4, every GP of the launch within 128 padded rows and all of
// them together within the LDS (mid_sweep_wanted); chosen by the GPs alone, never by the
// number of rows: a row's posterior does not depend on which rows are swept with it, and
// every rank of a sharded grid takes the same kernel.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "small_path.h"
#include "sweep_shared.h"

namespace {

#ifndef MID_NOBAR
#define MID_BARRIER __builtin_amdgcn_sched_barrier(0)
#else
#define MID_BARRIER
#endif

constexpr int kMidMaxNB = 8;          // row blocks of L^-1 (128 padded rows)
static_assert(kMidMaxNB <= kSepMinBlocks, "a launch reads every GP's tables up to its largest factor's blocks");
// waves per workgroup: three per SIMD up to 80 padded rows, two beyond (the instances for
// 6 .. 8 row blocks need 180 .. 210 registers)
constexpr int mid_waves(int nb, int d, int sep = 0) {
  if (sep > 0) return (sep == 1 ? nb <= 7 : sep == 2 ? nb <= 6 : nb <= 5) ? 12 : 8;     // (factor tables: fewer live registers)
  return (nb <= 4 || (nb == 5 && d <= 3)) ? 12 : 8;
}
constexpr int kMidWavesMax = 12;
constexpr int kMidKbRow = 80;         // doubles between the k-rows of a wave's transpose buffer
constexpr int kMidKbBuf = 4 * kMidKbRow;
constexpr int kMidLds = 160 * 1024;

struct MidParams {
  const GpDev* gps;
  int G;                      // GPs of this launch: g0 .. g0 + G - 1
  int g0;
  int Gtot;                   // GPs of the sweep (row pitch of Q)
  SweepPoints pts;
  ConfOut conf;
  // LDS (doubles): exp table | [waves] transpose buffers | per GP: A blocks, Xs, alpha
  int kb_off;
  int a_off[SGP_MAX_GPS];     // (a follower of a shared factor: its leader's)
  int x_off[SGP_MAX_GPS];
  int al_off[SGP_MAX_GPS];
  int lead[SGP_MAX_GPS];      // -1: a factor of its own; else the GP whose |L^-1 k|^2 it takes
  SepLaunch sep;              // tensor grid + factor tables (instances with SEP > 0)
};

typedef const __attribute__((address_space(4))) GpDev* mid_gpdev_t;

// SEP > 0: the rows are a tensor grid and the kernels products of RBF parts -- a covariance
// is the product of SEP per-axis table entries (SepLaunch, api.hip:sep_launch) instead of an
// evaluation: at n = 64 the evaluation is as many cycles of the shared fp64 pipe as the
// matrix instructions (profiles/r05/mid_kernel.txt).  D is not used then (instances: D = 1).
// Factors beyond 128 rows (tensor grids with factor tables, launch_sweep_mid below) go
// through in PASSES of row blocks [B0, B1): a pass keeps its blocks (b, jb <= b) resident,
// forms the covariances of j-blocks 0 .. B1 - 1 and this pass's share of |L^-1 k|^2; the share
// of the passes in front comes in through ConfOut::var, and only the FINAL pass of a GP
// (B1 = all its row blocks) forms alpha . k and runs the row epilogue.
template <int D, int B0, int B1, int WAVES, int SEP, bool FINAL>
__global__ __launch_bounds__(64 * WAVES) void k_sweep_mid(MidParams p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int T = 64 * WAVES;
  constexpr int NB = B1;                                   // j-blocks whose covariances are needed
  constexpr int NR = B1 - B0;                              // row blocks (accumulators) of the pass
  constexpr int kTri0 = B0 * (B0 + 1) / 2;
  constexpr int kTri = B1 * (B1 + 1) / 2 - kTri0;          // resident blocks
  const mid_gpdev_t gpc = (mid_gpdev_t)(p.gps);
  exp_tab_init(lds);
  // ---- the factors, inputs and weights of every GP -> LDS, once.  Block (b, jb <= b),
  // k-step q, lane (k, row): L^-1[16 b + row][16 jb + 4 q + k], zero outside the n x n
  // lower triangle (rows from n on hold whatever a pop or an append buffer left there).
  for (int g = p.g0; g < p.g0 + p.G; ++g) {
    const int n = gpc[g].n;
    if (p.lead[g] < 0) {
      const double* Li = gpc[g].Linv;
      const int64_t ld = gpc[g].ld;
      double* A = lds + p.a_off[g];
      for (int e = threadIdx.x; e < kTri * 256; e += T) {
        const int blk = e >> 8, q = (e >> 6) & 3, l = e & 63;
        int b = B0;
        while ((b + 1) * (b + 2) / 2 - kTri0 <= blk) ++b;
        const int jb = blk - (b * (b + 1) / 2 - kTri0);
        const int i = 16 * b + (l & 15), j = 16 * jb + 4 * q + (l >> 4);
        A[e] = (i < n && j <= i) ? Li[int64_t(i) * ld + j] : 0.0;
      }
    }
    if constexpr (SEP == 0) {
      const double* Xs = gpc[g].Xs;
      double* Xl = lds + p.x_off[g];
      for (int e = threadIdx.x; e < NB * 16 * D; e += T) Xl[e] = (e / D < n) ? Xs[e] : 0.0;
    }
    if constexpr (FINAL) {
      const double* al = gpc[g].alpha;
      double* all = lds + p.al_off[g];
      for (int e = threadIdx.x; e < NB * 16; e += T) all[e] = (e < n) ? al[e] : 0.0;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k4 = lane >> 4, c16 = lane & 15;
  const double* tab = lds;
  double* kbw = lds + p.kb_off + wave * kMidKbBuf;
  const int64_t N = p.pts.N;
  const int64_t ntiles = (N + 15) / 16;
  const int64_t tstep = int64_t(gridDim.x) * WAVES;
  auto load_x = [&](int64_t t, double (&xo)[D]) {
    int64_t r = t * 16 + c16;
    r = r < N ? r : N - 1;
#pragma unroll
    for (int k = 0; k < D; ++k)
      xo[k] = __builtin_nontemporal_load(p.pts.base + r * p.pts.stride_row + k * p.pts.stride_col);
  };
  // SEP: byte offsets of this lane's row (c16) and training points (4 q + k4 of a block of
  // 16: the four values of a lane side by side) in the tables; axis a's index =
  // (global row / stride_a) % count_a with stride_a = count_0 .. count_{a-1}
  constexpr int kAx = SEP > 0 ? SEP : 1;
  uint32_t soff[kAx];
  auto sep_offsets = [&](int64_t t) {
    int64_t r = t * 16 + c16;
    r = r < N ? r : N - 1;
    uint32_t q = uint32_t(p.sep.goff + r);
#pragma unroll
    for (int a = 0; a < kAx; ++a) {
      uint32_t idx = q;
      if (a + 1 < kAx) {
        const uint32_t c = p.sep.count[a];
        const uint32_t qn = q / c;
        idx = q - qn * c;
        q = qn;
      }
      soff[a] = idx * 128u + uint32_t(k4) * 32u;
    }
  };
  typedef const __attribute__((address_space(1))) double4_t* gvec_t;
  double lmax = -INFINITY;          // max l0 over the safe rows this wave has seen
  int64_t tile = int64_t(blockIdx.x) * WAVES + wave;
  double x[D];
  if (SEP == 0 && tile < ntiles) load_x(tile, x);
#pragma unroll 1
  for (; tile < ntiles; tile += tstep) {
    double xn[D];
    if constexpr (SEP == 0)
      load_x(tile + tstep < ntiles ? tile + tstep : tile, xn);     // the next tile's rows
    else
      sep_offsets(tile);
    const int64_t row = tile * 16 + c16;
    const bool writer = row < N && lane < 16;
    bool safe = true;
    double l0 = 0.0, ssq_lead = 0.0;
#pragma unroll 1
    for (int g = p.g0; g < p.g0 + p.G; ++g) {
      KernFast<D> kf;
      double xs[D];
      const double* Xg = lds + p.x_off[g] + k4 * D;
      const char* stab[kAx];
      uint32_t spitch[kAx];
      if constexpr (SEP == 0) {
        kf.load_const(&p.gps[g].kern);
        kf.template prep_t<true>(x, xs);
      } else {
#pragma unroll
        for (int a = 0; a < kAx; ++a) {
          stab[a] = reinterpret_cast<const char*>(uniform_ptr(p.sep.tab[g][a]));
          spitch[a] = __builtin_amdgcn_readfirstlane(p.sep.count[a] * 128u);
        }
      }
      // lane (k4, c16): kv[q] = k(X_{16 jb + 4 q + k4}, x_c16)
      auto factors = [&](int jb, double4_t (&f)[kAx]) {
#pragma unroll
        for (int a = 0; a < kAx; ++a)
          f[a] = *(gvec_t)(reinterpret_cast<const double4_t*>(stab[a] + jb * spitch[a] + soff[a]));
      };
      auto covariances = [&](int jb, const double4_t (&f)[kAx], double (&kv)[4]) {
        if constexpr (SEP == 0) {
          kf.template many4_t<true>(xs, Xg + jb * 16 * D, 4 * D, tab, kv);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            kv[q] = f[0][q];
#pragma unroll
            for (int a = 1; a < kAx; ++a) kv[q] *= f[a][q];
          }
        }
      };
      const double* alg = lds + p.al_off[g] + k4;
      double mean = 0.0, ssq;
      double4_t fcur[kAx], fnxt[kAx];
      if constexpr (SEP > 0) factors(0, fcur);
      if (p.lead[g] < 0) {
        const double* A = lds + p.a_off[g] + lane;
        double acc[NR][4];
#pragma unroll
        for (int b = 0; b < NR; ++b)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[b][m] = 0.0;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          double kv[4];
          covariances(jb, fcur, kv);
          if constexpr (SEP > 0) {
            // the next j-block's factors: their latency passes under this block's slots
            if (jb + 1 < NB) factors(jb + 1, fnxt);
          }
          if constexpr (FINAL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mean = fma(alg[jb * 16 + 4 * q], kv[q], mean);
            // (pinned here: left alone, the compiler sinks the whole chain to the end of the
            // tile and keeps every j-block's covariances and weights alive until then)
            asm volatile("" : "+v"(mean));
          }
          double kb[4][4];
          broadcast_quads<kMidKbRow>(kv, kbw, lane, kb);
          // the row blocks below (and on) the diagonal: the A operands of block b + 1 are
          // read while block b multiplies; the barriers keep the scheduler from hoisting
          // every read of the tile to its top (400 bytes of scratch per lane without them)
          const int bfirst = jb > B0 ? jb : B0;
          double a[2][4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a[0][q] = A[((bfirst * (bfirst + 1) / 2 - kTri0 + jb) * 4 + q) * 64];
#pragma unroll
          for (int b = (jb > B0 ? jb : B0); b < B1; ++b) {
            const int cur = (b - (jb > B0 ? jb : B0)) & 1;
            if (b + 1 < B1) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                a[cur ^ 1][q] = A[(((b + 1) * (b + 2) / 2 - kTri0 + jb) * 4 + q) * 64];
            }
            MID_BARRIER;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int m = 0; m < 4; ++m)
                acc[b - B0][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[cur][q], kb[m][q], acc[b - B0][m], 0, 0, 0);
            MID_BARRIER;
          }
          if constexpr (SEP > 0) {
#pragma unroll
            for (int a = 0; a < kAx; ++a) fcur[a] = fnxt[a];
          }
        }
        // squares: component m of a row block holds rows 4 ((l >> 2) & 3) + (l >> 4),
        // column 4 m + (l & 3); transposing fold over the lanes that share (l & 3), then
        // over the four 16-lane groups: every lane ends with |L^-1 k|^2 of point l & 15
        double sq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < NR; ++b)
#pragma unroll
          for (int m = 0; m < 4; ++m) sq[m] = fma(acc[b][m], acc[b][m], sq[m]);
        // (exchanges by DPP / permlane swaps, not through the LDS: sweep_shared.h)
        const bool a0 = (lane & 4) != 0, a1 = (lane & 8) != 0;
        const double v0 = (a0 ? sq[1] : sq[0]) + take_xor4(a0 ? sq[0] : sq[1], a0);
        const double v1 = (a0 ? sq[3] : sq[2]) + take_xor4(a0 ? sq[2] : sq[3], a0);
        const double t = (a1 ? v1 : v0) + take_xor8(a1 ? v0 : v1);
        ssq = sum_lane_groups_valu(t);
        if constexpr (B0 > 0) {
          // the share of the passes in front (rows beyond N: the last row's, unused)
          const int64_t rr = row < N ? row : N - 1;
          ssq = p.conf.var[int64_t(g) * N + rr] + ssq;
        }
        ssq_lead = ssq;
      } else {
        // the factor of the GP in front (GpDev::share): its |L^-1 k|^2, only alpha . k
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          double kv[4];
          if constexpr (SEP > 0) factors(jb, fcur);
          covariances(jb, fcur, kv);
#pragma unroll
          for (int q = 0; q < 4; ++q) mean = fma(alg[jb * 16 + 4 * q], kv[q], mean);
        }
        ssq = ssq_lead;
      }
      if constexpr (!FINAL) {
        if (writer) p.conf.var[int64_t(g) * N + row] = ssq;
        continue;
      }
      const double mu = sum_lane_groups_valu(mean);
      {
        // update_confidence_intervals + compute_safe_set (gp_opt.py:453-481); no
        // contraction: mu -+ beta sd is rounded as the reference rounds it
#pragma clang fp contract(off)
        const double var = fmax(gpc[g].kern.kdiag - ssq, 1e-15);   // GPy clip
        const double sd = sqrt(var);
        const double lo = mu - p.conf.beta * sd;
        const double up = mu + p.conf.beta * sd;
        if (g == 0) l0 = lo;
        safe = safe && (lo > p.conf.fmin[g]);
        if (writer) {
          __builtin_nontemporal_store(mu, p.conf.mean + int64_t(g) * N + row);
          __builtin_nontemporal_store(var, p.conf.var + int64_t(g) * N + row);
          if (p.conf.Q)
            *reinterpret_cast<double2_t*>(p.conf.Q + (row * p.Gtot + g) * 2) = double2_t{lo, up};
        }
      }
    }
    if (p.conf.S) {
      if (writer) p.conf.S[row] = safe ? 1 : 0;
      lmax = fmax(lmax, (writer && safe) ? l0 : -INFINITY);
    }
    if constexpr (SEP == 0) {
#pragma unroll
      for (int k = 0; k < D; ++k) x[k] = xn[k];
    }
  }
  if (p.conf.S && p.conf.partial) {
    lmax = wave_max(lmax);
    if (lane == 0) p.conf.partial[int(blockIdx.x) * WAVES + wave] = lmax;
  }
}

// One launch per GP (the passes below): the safe set and max l0[S] of several GPs from the
// intervals the launches left in Q (compute_safe_set, gp_opt.py:478-481).
struct MidFmin {
  double v[SGP_MAX_GPS];
};
__global__ __launch_bounds__(256) void k_mid_safe(const double* Q, uint8_t* S, double* partial,
                                                  MidFmin fmin, int G, int64_t N) {
  __shared__ double sh[4];
  const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
  double l0 = -INFINITY;
  if (row < N) {
    bool safe = true;
#pragma unroll
    for (int g = 0; g < SGP_MAX_GPS; ++g)
      if (g < G) safe = safe && (Q[(row * G + g) * 2] > fmin.v[g]);
    S[row] = safe ? 1 : 0;
    if (safe) l0 = Q[row * G * 2];
  }
  l0 = wave_max(l0);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = l0;
  __syncthreads();
  if (threadIdx.x == 0 && partial)
    partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

struct MidLayout {
  int nb = 0;
  size_t doubles = 0;
  MidParams p{};
};

// the LDS layout of a launch, or nb = 0 when the GPs do not fit
MidLayout mid_layout(const GpDev* gh, int Geff, int d) {
  MidLayout L;
  int nb = 0;
  for (int g = 0; g < Geff; ++g) nb = std::max(nb, gh[g].nblk);
  if (nb < 1 || nb > kMidMaxNB) return L;
  nb = std::max(nb, 4);
  const int tri = nb * (nb + 1) / 2;
  size_t off = kExpTabSize;
  L.p.kb_off = int(off);
  off += size_t(kMidWavesMax) * kMidKbBuf;
  int leader = -1;
  for (int g = 0; g < Geff; ++g) {
    const bool follows = g > 0 && gh[g].share >= 0 && leader >= 0;
    if (!follows) {
      leader = g;
      L.p.lead[g] = -1;
      L.p.a_off[g] = int(off);
      off += size_t(tri) * 256;
    } else {
      L.p.lead[g] = leader;
      L.p.a_off[g] = L.p.a_off[leader];
    }
    L.p.x_off[g] = int(off);
    off += size_t(nb) * 16 * d;
    L.p.al_off[g] = int(off);
    off += size_t(nb) * 16;
    off = (off + 1) & ~size_t(1);
  }
  if (off * 8 > size_t(kMidLds)) return L;
  L.nb = nb;
  L.doubles = off;
  return L;
}

// waves of a pass instance (129 .. 256 observations): the passes behind the first one hold at
// most 4 x 4 accumulators (101-159 VGPRs: three waves per SIMD), the first holds 9 x 4
constexpr int mid_pass_waves(int b0) { return b0 > 0 ? 12 : 8; }

// `nblocks` = 0: as many workgroups as the instance's waves ask for (the passes); the launch
// that leaves the partials of max l0[S] also says how many there are
template <int D, int B0, int B1, int SEP, bool FINAL>
int launch_mid_v(sgp_ctx* ctx, const MidParams& p, size_t lds_bytes, unsigned nblocks) {
  constexpr int kW = B1 > kMidMaxNB ? mid_pass_waves(B0) : mid_waves(B1, D, SEP);
  static bool attr_set = false;
  if (!attr_set) {
    SGP_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(&k_sweep_mid<D, B0, B1, kW, SEP, FINAL>),
                     hipFuncAttributeMaxDynamicSharedMemorySize, kMidLds));
    attr_set = true;
  }
  if (nblocks == 0) {
    const int64_t ntiles = (p.pts.N + 15) / 16;
    nblocks = unsigned(std::max<int64_t>(1, std::min<int64_t>(ctx->num_cu, (ntiles + kW - 1) / kW)));
    if (p.conf.S && p.conf.partial) ctx->sweep_partials = int(nblocks) * kW;
  }
  hipLaunchKernelGGL((k_sweep_mid<D, B0, B1, kW, SEP, FINAL>), dim3(nblocks), dim3(64 * kW),
                     lds_bytes, ctx->stream, p);
  return 0;
}

template <int D, int SEP>
int launch_mid_d(sgp_ctx* ctx, const MidParams& p, int nb, size_t lds_bytes, unsigned nblocks) {
  switch (nb) {
    case 4: return launch_mid_v<D, 0, 4, SEP, true>(ctx, p, lds_bytes, nblocks);
    case 5: return launch_mid_v<D, 0, 5, SEP, true>(ctx, p, lds_bytes, nblocks);
    case 6: return launch_mid_v<D, 0, 6, SEP, true>(ctx, p, lds_bytes, nblocks);
    case 7: return launch_mid_v<D, 0, 7, SEP, true>(ctx, p, lds_bytes, nblocks);
    case 8: return launch_mid_v<D, 0, 8, SEP, true>(ctx, p, lds_bytes, nblocks);
  }
  return -2;
}

// ---- factors of 129 .. 256 rows on tensor grids with factor tables: passes --------------
// Row blocks [0, 9), [9, 13), [13, 16) -- at most 46 resident blocks (92 KB) each; a pass
// re-forms the covariances of every j-block up to its last row block, which costs next to
// nothing from tables (and is why evaluated kernels stay with the 4-wave kernel here).
struct MidPass {
  int b0, b1;
};
int mid_passes(int nb, MidPass (&ps)[3]) {
  int n = 0;
  ps[n++] = MidPass{0, nb < 9 ? nb : 9};
  if (nb > 9) ps[n++] = MidPass{9, nb < 13 ? nb : 13};
  if (nb > 13) ps[n++] = MidPass{13, nb};
  return n;
}

template <int SEP>
int launch_mid_pass(sgp_ctx* ctx, const MidParams& p, MidPass ps, bool final, size_t lds_bytes,
                    unsigned nblocks) {
#define MID_PASS(B0, B1, F) \
  if (ps.b0 == B0 && ps.b1 == B1 && final == F)  \
    return launch_mid_v<1, B0, B1, SEP, F>(ctx, p, lds_bytes, nblocks);
  // (a smaller GP next to one with more than 128 observations: one pass, at least 4 blocks)
  MID_PASS(0, 4, true) MID_PASS(0, 5, true) MID_PASS(0, 6, true) MID_PASS(0, 7, true)
  MID_PASS(0, 8, true)
  MID_PASS(0, 9, true) MID_PASS(0, 9, false)
  MID_PASS(9, 10, true) MID_PASS(9, 11, true) MID_PASS(9, 12, true) MID_PASS(9, 13, true)
  MID_PASS(9, 13, false)
  MID_PASS(13, 14, true) MID_PASS(13, 15, true) MID_PASS(13, 16, true)
#undef MID_PASS
  return -2;
}

}  // namespace

// 49 .. 128 observations in the largest GP of the launch, single-part kernels, d <= 4,
// everything resident in LDS.  By the GPs alone (never by the rows).  SGP_NO_MID=1 /
// sgp_ctx_set_sweep(1 or 2) keep the general kernels (A/B runs, tests).
bool mid_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, int d) {
  static const bool off = getenv("SGP_NO_MID") != nullptr;
  const int choice = ctx->sweep_choice & 3;
  if (off || choice == 1 || choice == 2) return false;
  if (d > 4) return false;
  int nmax = 0;
  for (int g = 0; g < Geff; ++g) {
    if (gh[g].kern.n_parts != 1) return false;
    nmax = std::max(nmax, gh[g].n);
  }
  if (nmax <= kTinyMaxN || nmax > 16 * kMidMaxNB) return false;
  return mid_layout(gh, Geff, d).nb > 0;
}

// ... and 129 .. 256 observations when the launch has factor tables (a tensor grid, RBF
// parts): one launch per GP and pass.  Not with shared factors (a follower would need its
// leader's |L^-1 k|^2 across launches: the 4-wave kernel's riders do that better), and S
// only together with Q (the last launch reads the other GPs' intervals back from there).
bool mid_passes_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, const SepLaunch* sep,
                       const ConfOut& conf) {
  static const bool off = getenv("SGP_NO_MID") != nullptr || getenv("SGP_NO_MID_PASSES") != nullptr;
  const int choice = ctx->sweep_choice & 3;
  if (off || choice == 1 || choice == 2 || !sep || sep->naxes < 1 || sep->naxes > 3) return false;
  int nmax = 0;
  for (int g = 0; g < Geff; ++g) {
    if (gh[g].share >= 0) return false;
    nmax = std::max(nmax, gh[g].n);
  }
  if (nmax <= 16 * kMidMaxNB || nmax > 256) return false;
  return !(conf.S && !conf.Q && Geff > 1);
}

int launch_sweep_mid(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d, int Geff,
                     double flops, const SepLaunch* sep) {
  const int64_t ntiles = (a.pts.N + 15) / 16;
  if (mid_passes_wanted(ctx, gh, Geff, sep, a.conf)) {
    const unsigned nblocks = 0;        // (per pass: launch_mid_v)
    ctx->sweep_partials = 0;
    SweepTimer timer;
    SGP_TRY(timer.begin(ctx, flops));
    for (int g = 0; g < Geff; ++g) {
      MidPass ps[3];
      const int nb = std::max(gh[g].nblk, 4);
      const int np = mid_passes(nb, ps);
      for (int i = 0; i < np; ++i) {
        const bool final = i == np - 1, last = final && g == Geff - 1 && Geff == 1;
        MidParams p{};
        p.gps = a.gps;
        p.G = 1;
        p.g0 = g;
        p.Gtot = Geff;
        p.pts = a.pts;
        p.conf = a.conf;
        if (!last) {
          p.conf.S = nullptr;
          p.conf.partial = nullptr;
        }
        p.sep = *sep;
        for (int h = 0; h < SGP_MAX_GPS; ++h) p.lead[h] = -1;
        size_t off = kExpTabSize;
        p.kb_off = int(off);
        off += size_t(kMidWavesMax) * kMidKbBuf;
        p.a_off[g] = int(off);
        off += size_t(ps[i].b1 * (ps[i].b1 + 1) / 2 - ps[i].b0 * (ps[i].b0 + 1) / 2) * 256;
        p.x_off[g] = int(off);
        p.al_off[g] = int(off);
        off += size_t(ps[i].b1) * 16;
        SGP_CHECK(ctx, off * 8 <= size_t(kMidLds), "launch_sweep_mid: pass of %zu bytes", off * 8);
        int rc = -2;
        switch (sep->naxes) {
          case 1: rc = launch_mid_pass<1>(ctx, p, ps[i], final, off * 8, nblocks); break;
          case 2: rc = launch_mid_pass<2>(ctx, p, ps[i], final, off * 8, nblocks); break;
          case 3: rc = launch_mid_pass<3>(ctx, p, ps[i], final, off * 8, nblocks); break;
        }
        if (rc != 0) {
          sgp_set_error(ctx, "launch_sweep_mid: no instance for row blocks %d .. %d", ps[i].b0,
                        ps[i].b1);
          return rc;
        }
      }
    }
    if (Geff > 1 && a.conf.S) {
      MidFmin fm;
      for (int h = 0; h < SGP_MAX_GPS; ++h) fm.v[h] = a.conf.fmin[h];
      const unsigned nb = unsigned((a.pts.N + 255) / 256);
      hipLaunchKernelGGL(k_mid_safe, dim3(nb), dim3(256), 0, ctx->stream, a.conf.Q, a.conf.S,
                         a.conf.partial, fm, Geff, a.pts.N);
      ctx->sweep_partials = int(nb);
    }
    SGP_HIP(ctx, hipGetLastError());
    return timer.end(ctx);
  }
  MidLayout L = mid_layout(gh, Geff, d);
  SGP_CHECK(ctx, L.nb > 0, "launch_sweep_mid: the GPs do not fit (mid_sweep_wanted)");
  MidParams p = L.p;
  p.gps = a.gps;
  p.G = Geff;
  p.g0 = 0;
  p.Gtot = Geff;
  p.pts = a.pts;
  p.conf = a.conf;
  const int waves = mid_waves(L.nb, d, sep ? sep->naxes : 0);
  if (sep) p.sep = *sep;
  const unsigned nblocks = unsigned(std::max<int64_t>(
      1, std::min<int64_t>(ctx->num_cu, (ntiles + waves - 1) / waves)));
  ctx->sweep_partials = int(nblocks) * waves;
  SweepTimer timer;
  SGP_TRY(timer.begin(ctx, flops));
  int rc = -2;
  if (sep) {
    switch (sep->naxes) {
      case 1: rc = launch_mid_d<1, 1>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
      case 2: rc = launch_mid_d<1, 2>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
      case 3: rc = launch_mid_d<1, 3>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
    }
  } else {
    switch (d) {
      case 1: rc = launch_mid_d<1, 0>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
      case 2: rc = launch_mid_d<2, 0>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
      case 3: rc = launch_mid_d<3, 0>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
      case 4: rc = launch_mid_d<4, 0>(ctx, p, L.nb, L.doubles * 8, nblocks); break;
    }
  }
  if (rc != 0) {
    sgp_set_error(ctx, "launch_sweep_mid: no instance for d = %d, %d row blocks", d, L.nb);
    return rc;
  }
  SGP_HIP(ctx, hipGetLastError());
  return timer.end(ctx);
}
