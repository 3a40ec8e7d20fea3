// The posterior sweep for FEW observations (every GP of the launch has n <= 48 rows --
// the regime of the reference's own examples and tests, which run n <= 20): one thread
// per candidate row, fp64 VALU only.
//
//   v = L^-1 k(X, x),  var = k(x,x) - |v|^2,  mean = alpha . k(X, x)
//   (gp.predict_noiseless, safeopt/gp_opt.py:469; update_confidence_intervals +
//    compute_safe_set, gp_opt.py:453-481)
//
// Why not the matrix-core kernels (sweep.hip): a 16-row tile of theirs walks through a
// chain of LDS round trips, cross-lane folds and a barrier per stage with two waves per
// SIMD to hide it -- at n = 20 a million rows take 0.09 ms, 5 % of either roof
// (profiles/r04/small_n.txt).  On gfx950 the fp64 VALU peak EQUALS the fp64 MFMA peak
// (78.6 TFLOP/s), so for a factor that fits the scalar registers' reach nothing is lost
// by staying on the VALU: the thread keeps its n covariances in registers, the entries
// of L^-1, alpha and the training rows are wave-uniform and come through SCALAR loads
// from constant address space (one s_load_dwordx16 feeds 8 FMAs; a v_fma_f64 takes one
// scalar operand), there is no LDS traffic, no cross-lane operation and no barrier, and
// 3-7 waves per SIMD hide what latency is left.  NP = 8 / 16 / 32 / 48 (compile time):
// the triangular product is fully unrolled, NP (NP + 1) / 2 FMAs.  Measured on 1e6 rows
// (profiles/r04/small_n.txt): n = 8 0.020 ms against 0.060-0.076 for the 4-wave kernel,
// n = 20 0.037 / 0.091-0.117, n = 48 0.094 / 0.141-0.177; at n = 56-64 the 64 + covariance
// registers leave two waves per SIMD and the kernels meet (0.157-0.200 / 0.175-0.231):
// the VALU kernel runs up to 48 observations.
//
// Padding: k_j = 0 for j >= n and alpha is zero padded; rows >= n of the dense L^-1 are
// never SUMMED (they hold whatever the factor left there: the identity of the padding,
// the row of a popped observation).  Results differ from the matrix-core kernels in the
// last bits (another summation order); for a grid the choice of the kernel depends on the
// sizes of the GPs only, so every rank and every shard takes the same one
// (tiny_sweep_wanted below).
#include "tiny_row.h"

namespace {

struct TinyParams {
  const GpDev* gps;
  int G;
  SweepPoints pts;
  ConfOut conf;
};

template <int D, int NP, bool SINGLE>
__global__ __launch_bounds__(256) void k_sweep_tiny(TinyParams p) {
  __shared__ double tab[kExpTabSize];
  __shared__ double sh_max[4];
  exp_tab_init(tab);
  __syncthreads();
  const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const bool valid = row < p.pts.N;
  const int64_t r = valid ? row : p.pts.N - 1;
  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k)
    x[k] = __builtin_nontemporal_load(p.pts.base + r * p.pts.stride_row + k * p.pts.stride_col);
  bool safe = true;
  double l0 = 0.0;
  tiny_row<D, NP, SINGLE>(p.gps, p.G, p.conf, p.pts.N, x, row, valid, tab, safe, l0);
  if (p.conf.S) {
    if (valid) p.conf.S[row] = safe ? 1 : 0;
    // max l0 over the safe rows of the workgroup (folded by the consumer, sets.hip)
    double m = wave_max((valid && safe) ? l0 : -INFINITY);
    if ((threadIdx.x & 63) == 0) sh_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
      p.conf.partial[blockIdx.x] = fmax(fmax(sh_max[0], sh_max[1]), fmax(sh_max[2], sh_max[3]));
  }
}

template <int D, int NP>
int launch_tiny_np(sgp_ctx* ctx, const TinyParams& p, bool single, unsigned nblocks) {
  if (single)
    hipLaunchKernelGGL((k_sweep_tiny<D, NP, true>), dim3(nblocks), dim3(256), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL((k_sweep_tiny<D, NP, false>), dim3(nblocks), dim3(256), 0, ctx->stream, p);
  return 0;
}

template <int D>
int launch_tiny_d(sgp_ctx* ctx, const TinyParams& p, int np, bool single, unsigned nblocks) {
  if (np <= 8) return launch_tiny_np<D, 8>(ctx, p, single, nblocks);
  if (np <= 16) return launch_tiny_np<D, 16>(ctx, p, single, nblocks);
  if (np <= 32) return launch_tiny_np<D, 32>(ctx, p, single, nblocks);
  return launch_tiny_np<D, 48>(ctx, p, single, nblocks);
}

}  // namespace

// Few observations in every GP of the launch: the VALU kernel (SGP_NO_TINY=1 /
// sgp_ctx_set_sweep(1 or 2) keep the matrix-core kernels, A/B runs and tests).
// A grid (the rows are a rank's shard): by the sizes of the GPs alone, so that every rank
// takes the same kernel.  A set of points handed over per call (swarm particles, predict):
// the thread-per-row kernel needs rows to hide its dependent chains behind -- a swarm of
// 20 particles is ONE wave -- and the 4-wave kernel, which spreads a row's training
// points over lanes, is faster below ~1500 rows per observation (20 rows, n = 40: 9.4
// against 15.2 us; crossover at 32 k rows for n = 20, 55 k for n = 40; from n <= 10 the
// VALU kernel wins at any size: scripts/dev/tiny_crossover.py).
bool tiny_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, int64_t rows,
                       bool rows_sharded) {
  static const bool off = getenv("SGP_NO_TINY") != nullptr;
  if (off || (ctx->sweep_choice & 3) != 0) return false;
  int nmax = 0;
  for (int g = 0; g < Geff; ++g) nmax = std::max(nmax, gh[g].n);
  if (nmax > kTinyMaxN) return false;
  return rows_sharded || nmax <= 10 || rows >= int64_t(1536) * nmax;
}

int launch_sweep_tiny(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d, int Geff,
                      double flops) {
  TinyParams p{};
  p.gps = a.gps;
  p.G = Geff;
  p.pts = a.pts;
  p.conf = a.conf;
  int np = 1;
  bool single = true;
  for (int g = 0; g < Geff; ++g) {
    np = std::max(np, gh[g].n);
    single = single && gh[g].kern.n_parts == 1;
  }
  const unsigned nblocks = unsigned((a.pts.N + 255) / 256);
  ctx->sweep_partials = int(nblocks);
  SweepTimer timer;
  SGP_TRY(timer.begin(ctx, flops));
  int rc = -2;
  switch (d) {
    case 1: rc = launch_tiny_d<1>(ctx, p, np, single, nblocks); break;
    case 2: rc = launch_tiny_d<2>(ctx, p, np, single, nblocks); break;
    case 3: rc = launch_tiny_d<3>(ctx, p, np, single, nblocks); break;
    case 4: rc = launch_tiny_d<4>(ctx, p, np, single, nblocks); break;
    case 5: rc = launch_tiny_d<5>(ctx, p, np, single, nblocks); break;
    case 6: rc = launch_tiny_d<6>(ctx, p, np, single, nblocks); break;
    case 7: rc = launch_tiny_d<7>(ctx, p, np, single, nblocks); break;
    case 8: rc = launch_tiny_d<8>(ctx, p, np, single, nblocks); break;
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
  if (rc != 0) return rc;
  SGP_HIP(ctx, hipGetLastError());
  return timer.end(ctx);
}
