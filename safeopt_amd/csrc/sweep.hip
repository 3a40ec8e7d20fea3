// posterior_sweep: the dominant kernel of the path.
//
// Replaces gp.predict_noiseless(self.inputs) + the Q update of
// SafeOpt.update_confidence_intervals (safeopt/gp_opt.py:453-476), the safe-set
// test (:478-481), SafeOptSwarm._compute_particle_fitness (:901-1013) and the
// per-candidate re-prediction of the expander loop (:579-606).
//
// For a tile of candidate rows the kernel forms the covariance tile
// K[j, pt] = k(X_j, x_pt) ON THE FLY in registers, in the (k-step, point) lane
// order of the fp64 matrix instructions, and contracts it with A = L^-1 (lower
// triangular, pre-packed in A-operand order) on the fp64 matrix cores
// (v_mfma_f64_4x4x4_4b_f64, see "matrix part" below):
//     var(pt)  = k(x,x) - || A K[:, pt] ||^2          (n^2 flops / row)
//     mean(pt) = alpha . K[:, pt]                      (2n flops / row)
// K (n x N doubles, 1.6 GB at n=200, N=1e6) is never written to memory.
//
// Work split: workgroup = 4 waves = 64 rows, two workgroups per CU (each SIMD
// hosts one wave of either); wave w owns 16 rows (the MFMA N dimension); the
// waves of a workgroup share the staged A chunk through LDS.  The rows of A are
// processed in chunks of 16 MFMA row blocks (256 rows) held in 16 accumulator
// slots per wave; the triangular structure is exploited at 16x16 block
// granularity (a j-block only feeds row blocks >= its own index).  The kernel
// is persistent: 2 x num_CU workgroups walk over the row tiles, and the
// (tile, GP, chunk, j-block) loop nest is flattened into one stage sequence so
// that the LDS-DMA of stage s+1 always runs under the MFMAs of stage s.
#include <stdio.h>
#include <algorithm>
#include <stdlib.h>

#include "kern_eval.h"

namespace {

constexpr int kMaxWaves = 8;         // waves per CU at 256 VGPRs
constexpr int kSweepWaves = 4;       // waves per workgroup (two workgroups per CU)
constexpr int kIB = 16;                // accumulator slots = 256 rows of L^-1
constexpr int kJC = 16;                // training points per staged chunk
constexpr int kSteps = kJC / 4;        // MFMA k-steps per chunk (one j-block)
constexpr int kATile = kIB * kSteps * 64;           // doubles (32 KB)
constexpr int kXTile = kJC * SGP_MAX_D;             // doubles
constexpr int kBuf = kATile + kXTile + kJC;         // + alpha chunk
constexpr int kTabOff = 2 * kBuf + kMaxWaves;           // exp table (32 doubles)
constexpr size_t kLdsBytes = (size_t(kTabOff) + kExpTabSize) * sizeof(double);

enum { MODE_CONF = 0, MODE_FITNESS = 1 };

// Timing experiments ("what does the kernel cost without X"; results are wrong
// with any bit set): built only with -DSGP_INSTRUMENT, selected at run time by
// SGP_ABLATE=<mask>: 1 no stage barrier, 2 no LDS-DMA, 4 no covariance
// evaluation, 8 no MFMA, 16 no mean/var/Q stores, 32 no per-GP epilogue at all.  profiles/r01/ablation.txt holds the numbers.
#ifdef SGP_INSTRUMENT
#define SGP_ABL(mask) (p.ablate & (mask))
// cycle stamps of one wave's stage phases (s_memtime; waits for lgkmcnt(0))
#define SGP_STAMP(k)                                              \
  if (p.stamps) {                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
    stamp_acc[k] += t_ - stamp_last;                              \
    stamp_last = t_;                                              \
  }
#else
#define SGP_ABL(mask) false
#define SGP_STAMP(k)
#endif

struct SweepParams {
  const GpDev* gps;
  int G;
  int mode;
#ifdef SGP_INSTRUMENT
  int ablate;      // timing experiments (scripts/ablate.py), see SGP_ABL
  int skew;        // SGP_SKEW: the second workgroup of a CU starts 64*skew cycles late
  int* cu_count;   // [8 XCC][256] arrival order per compute unit
  unsigned long long* stamps;   // SGP_STAMPS: [workgroup][wave][8] cycle totals
#endif
  SweepPoints pts;
  ConfOut conf;
  FitnessArgs fit;
  // Covariance cache (n > 256 only): the values a lane evaluates in the
  // triangular part of chunk c are needed again by every later chunk; they are
  // parked in global memory, [workgroup][wave][j-block][lane][4], and read back
  // (one stage ahead) instead of being re-evaluated.  nullptr: re-evaluate.
  double* kvc;
  int kvc_blocks;   // j-blocks per (workgroup, wave) slab
};

// The descriptor fields the stage pipeline touches every iteration, hoisted out
// of the device array once per GP (they live in SGPRs across the stage loop).
// (The pointers come out of a descriptor in memory, so the compiler would emit
// FLAT loads for them; a FLAT load also counts on lgkmcnt, and every LDS wait
// of the stage would then sit out a global-memory latency.  Hence the explicit
// global address space.)
typedef const __attribute__((address_space(1))) double* gptr_t;
struct GpView {
  gptr_t Apack;
  gptr_t Xs;
  gptr_t alpha;
  int nsteps_total;
  // wave-uniform values, pinned to SGPRs (they are loaded through VGPRs and
  // would otherwise stay there -- or get spilled to scratch at 256 VGPRs)
  static __device__ __forceinline__ gptr_t uniform(const double* q) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return (gptr_t)reinterpret_cast<const double*>((uint64_t(hi) << 32) | lo);
  }
  __device__ __forceinline__ void load(const GpDev& gp) {
    Apack = uniform(gp.Apack);
    Xs = uniform(gp.Xs);
    alpha = uniform(gp.alpha);
    nsteps_total = __builtin_amdgcn_readfirstlane(gp.n_pad >> 2);
  }
};

// Asynchronous global -> LDS copy of one A chunk (LDS-DMA, no VGPR round trip).
// LDS image: slot-major A[slot][step][lane]; slot = row block - b0 + shift so
// that the last row block of the chunk always sits in slot 15.  A slot is 2 KB
// = two 1 KB instructions (k-steps {0,1} and {2,3}, the second through the
// instruction offset, which applies to the global and the LDS address alike);
// wave w copies slots w, w + NW, ...  Only the slots the j-block reads are
// fetched (`lo` = its first active slot).  Everything but the per-lane offset
// is scalar, and the address of the next slot is one 64-bit add away: the
// wave's issue slots are the scarce resource of this loop (scripts/
// stagebench.py), so the bookkeeping per copy is kept to a handful of SALU ops.
template <int NW>
__device__ __forceinline__ void stage_dma(const GpView& gp, double* buf, int b0,
                                          int shift, int jb, int lo, int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int64_t row_stride = int64_t(gp.nsteps_total) * 64;   // doubles / row block
  gptr_t src = gp.Apack + (int64_t(b0 - shift + wave) * gp.nsteps_total +
                           jb * kSteps) * 64 + lane * 2;
  double* dst = buf + wave * (2 * 128);                       // wave-uniform
#pragma unroll
  for (int k = 0; k < kIB / NW; ++k) {
    if (wave + NW * k >= lo) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)dst, 16, 1024, 0);
    }
    src += NW * row_stride;
    dst += NW * (2 * 128);
  }
}

// Training rows (pre-scaled) and alpha entries of j-block jb: contiguous runs of
// 16 D and 16 doubles in GpDev::Xs / alpha, copied by LDS-DMA as well -- wave 0
// moves the rows (8 D lanes x 16 B), wave 1 the alpha run (8 lanes x 16 B); the
// other lanes are masked off and write nothing.
template <int D>
__device__ __forceinline__ void stage_x_dma(const GpView& gp, double* buf, int jb,
                                            int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  if (wave == 0) {
    if (lane < 8 * D) {
      gptr_t src = gp.Xs + (jb * kJC * D + lane * 2);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(buf + kATile), 16, 0, 0);
    }
  } else if (wave == 1) {
    if (lane < 8) {
      gptr_t src = gp.alpha + (jb * kJC + lane * 2);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(buf + kATile + kXTile), 16,
          0, 0);
    }
  }
}

// ---- matrix part ------------------------------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 is the fp64 matrix instruction that reaches the chip's
// peak on gfx950 (74-77 TFLOP/s measured vs 49 for v_mfma_f64_16x16x4_f64,
// scripts/microbench.py).  Its operand maps (scripts/probe_mfma_layout.py):
//   A[blk][i][k] <- lane 16k + 4blk + i     B[blk][k][j] <- lane 16k + 4blk + j
//   D[blk][i][j] -> lane 16i + 4blk + j
// Used with blk = 4-row group: the A operand is then the SAME register the
// 16x16x4 form takes (lane = 16k + row, row = 4blk + i), one instruction is a
// 16-row x 4-column x 4-deep product, and a 16-column slab needs four of them
// (m = 0..3) whose B operands are the covariance register with column quad m
// broadcast to all four quads of each 16-lane row -- two ds_swizzle_b32 per
// operand, done once per k-step and reused by every row block.  Accumulator
// component m of a slot holds rows 4((l>>2)&3) + (l>>4), column 4m + (l&3).
__device__ __forceinline__ double quad_bcast(double v, int m) {
  // src lane = (lane & 0b110011) | (m << 2)   (BITMASK_PERM: and 0x13, or m<<2)
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int slo, shi;
  switch (m) {
    case 0: slo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (0 << 5));
            shi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (0 << 5)); break;
    case 1: slo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (4 << 5));
            shi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (4 << 5)); break;
    case 2: slo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (8 << 5));
            shi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (8 << 5)); break;
    default: slo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (12 << 5));
             shi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (12 << 5)); break;
  }
  return __hiloint2double(shi, slo);
}

// A operands of one slot: the 4 k-steps of the staged j-block.
__device__ __forceinline__ void load_slot(double (&ops)[4], const double* aT,
                                          int slot) {
#pragma unroll
  for (int q = 0; q < 4; ++q) ops[q] = aT[(slot * kSteps + q) * 64];
}

// One 16-wide j-block against accumulator slots lo..15.  Slots are guarded by
// wave-uniform branches (the active set is a suffix); a slot is 16 MFMAs on
// four independent accumulators, and the next slot's A operands are read from
// LDS while they execute.
__device__ __forceinline__ void mfma_jblock(int lo, double4_t (&acc)[kIB],
                                            const double* aT,
                                            const double (&kv)[4]) {
  double kb[4][4];  // [k-step][column quad]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int m = 0; m < 4; ++m) kb[q][m] = quad_bcast(kv[q], m);

  double opsA[4], opsB[4];
#pragma unroll
  for (int s = 0; s < kIB; ++s) {
    if (s >= lo) {
      double(&cur)[4] = (s & 1) ? opsB : opsA;
      double(&nxt)[4] = (s & 1) ? opsA : opsB;
      if (s == lo) load_slot(cur, aT, s);
      if (s + 1 < kIB) load_slot(nxt, aT, s + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[s][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(cur[q], kb[q][m],
                                                         acc[s][m], 0, 0, 0);
      }
    }
  }
}

// SafeOptSwarm._compute_penalty (gp_opt.py:874-899) for one value.
__device__ __forceinline__ double swarm_penalty(double slack) {
  double pen = fmin(slack, 0.0);
  if (slack < 0.0 && slack > -0.001) pen *= 2.0;
  if (slack <= -0.001 && slack > -0.1) pen *= 5.0;
  if (slack <= -0.1 && slack > -1.0) pen *= 10.0;
  if (slack < -1.0) pen = -300.0 * pen * pen;
  return pen;
}

// The fitness shaping of SafeOptSwarm._compute_particle_fitness
// (gp_opt.py:925-1013) on resident mean / var ([G][P]) -- the epilogue of
// k_sweep<.., MODE_FITNESS> as a kernel of its own, for the small-swarm path
// (posterior_small in factor.hip) where the posterior does not come out of the
// sweep.  One thread per particle; same formulas, same order as the epilogue.
__global__ void k_fitness_small(int G, int64_t P, const double* mean,
                                const double* var, FitnessArgs f) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int st = f.swarm_type;
  const int Geff = (st == SGP_SWARM_GREEDY) ? 1 : G;
  bool safe = true;
  double values = 0.0, interest = 1.0, total_pen = 0.0, lower = 0.0;
  for (int g = 0; g < Geff; ++g) {
    const double mu = mean[int64_t(g) * P + i];
    const double sd = sqrt(var[int64_t(g) * P + i]);
    lower = mu - f.beta * sd;
    if (g == 0) {
      values = sd / f.scaling[0];
      if (st == SGP_SWARM_EXPANDERS) interest = double(G);
      if (st == SGP_SWARM_MAXIMIZERS) {
        const double upper = mu + f.beta * sd;
        const double z = 10.0 * (upper - f.best_lower_bound) / f.scaling[0];
        interest = 1.0 / (1.0 + exp(-z));  // scipy.special.expit
      }
    } else {
      values = fmax(values, sd / f.scaling[g]);
    }
    if (f.fmin[g] != -INFINITY) {
      double slack = lower - f.fmin[g];
      safe = safe && (slack >= 0.0);
      if (st != SGP_SWARM_SAFE_SET) {
        slack = slack / f.scaling[g];
        total_pen += swarm_penalty(slack);
        if (st == SGP_SWARM_EXPANDERS) {
          const double z = slack / 0.2;   // scipy.stats.norm.pdf(slack, scale=0.2)
          interest *= exp(-0.5 * z * z) / 2.5066282746310002 / 0.2;
        }
      }
    }
  }
  double out;
  bool ok = safe;
  if (st == SGP_SWARM_GREEDY) {
    out = lower;
    ok = true;
  } else if (st == SGP_SWARM_SAFE_SET) {
    out = lower;
  } else {
    out = (values + total_pen) * interest;
  }
  f.values[i] = out;
  f.safe[i] = ok ? 1 : 0;
}

// Position in the flattened stage sequence of one workgroup:
//   for tile: for gp: for chunk (16 row blocks of L^-1): for jb (16 columns)
// A stage's LDS image (A chunk, X rows, alpha) depends on (gp, chunk, jb) only,
// so the image of the NEXT stage is always prefetched while the current one is
// consumed -- also across chunk, GP and tile boundaries.  The kernel is
// persistent (one workgroup per CU walks over tiles) so that nothing but the
// very first stage of a launch pays an exposed global -> LDS latency.
struct StagePos {
  int64_t tile;
  int g, c, jb;
  // derived, per (g, c)
  int b0, nib, shift, njb, nchunks;
};

__device__ __forceinline__ void stage_derive(StagePos& sp, const GpDev* gps) {
  const int nblk = gps[sp.g].nblk;
  sp.nchunks = (nblk + kIB - 1) / kIB;
  sp.b0 = sp.c * kIB;
  sp.nib = min(kIB, nblk - sp.b0);
  sp.shift = kIB - sp.nib;
  sp.njb = sp.b0 + sp.nib;
}

// Advance to the following stage; returns false after the last stage of the
// last tile of this workgroup.
__device__ __forceinline__ bool stage_next(StagePos& sp, const GpDev* gps,
                                           int Geff, int64_t ntiles,
                                           int64_t tile_stride) {
  if (++sp.jb < sp.njb) return true;
  sp.jb = 0;
  if (++sp.c >= sp.nchunks) {
    sp.c = 0;
    if (++sp.g >= Geff) {
      sp.g = 0;
      sp.tile += tile_stride;
      if (sp.tile >= ntiles) return false;
    }
  }
  stage_derive(sp, gps);
  return true;
}

template <int D, int NW>
__device__ __forceinline__ void stage_issue(const StagePos& sp, const GpView& gp,
                                            double* buf, int tid) {
  const int lo = sp.shift + max(0, sp.jb - sp.b0);
  stage_dma<NW>(gp, buf, sp.b0, sp.shift, sp.jb, lo, tid);
  stage_x_dma<D>(gp, buf, sp.jb, tid);
}

template <int D, int NW, int MODE, bool CACHE>
__global__ __launch_bounds__(64 * NW, 2) void k_sweep(SweepParams p) {
  constexpr int kWaves = NW;
  constexpr int kTilePts = 16 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* red = lds + 2 * kBuf;  // all LDS lives in the one dynamic region
  const double* tab = lds + kTabOff;
  exp_tab_init(lds + kTabOff);   // visible after the first staging barrier

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr bool conf = MODE == MODE_CONF;   // compile-time: no dead state
  const int st = p.fit.swarm_type;
  const int Geff = (!conf && st == SGP_SWARM_GREEDY) ? 1 : p.G;
  const int64_t ntiles = (p.pts.N + kTilePts - 1) / kTilePts;

  StagePos cur;
  cur.tile = blockIdx.x;
  cur.g = cur.c = cur.jb = 0;
  if (cur.tile >= ntiles) return;
#ifdef SGP_INSTRUMENT
  if (p.skew > 0) {   // phase experiment: delay the second workgroup of every CU
    int* flag = reinterpret_cast<int*>(lds + 2 * kBuf);
    if (threadIdx.x == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // XCC_ID
      *flag = atomicAdd(p.cu_count + (((xcc & 7) << 8) | ((hw >> 8) & 0xff)), 1);
    }
    __syncthreads();
    const int order = *flag;
    __syncthreads();
    if (order & 1)
      for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(1);
  }
#endif
  stage_derive(cur, p.gps);

  // candidate rows of the current tile (and, prefetched, of the next one)
  auto load_x = [&](int64_t tile, double (&xo)[D]) {
    int64_t r = tile * kTilePts + wave * 16 + (lane & 15);
    r = r < p.pts.N ? r : p.pts.N - 1;
#pragma unroll
    for (int k = 0; k < D; ++k)
      xo[k] = p.pts.base[r * p.pts.stride_row + k * p.pts.stride_col];
  };
  double x[D], xnext[D];
  load_x(cur.tile, x);
#pragma unroll
  for (int k = 0; k < D; ++k) xnext[k] = x[k];

  GpView gv_next;          // GP of the stage being prefetched
  gv_next.load(p.gps[0]);
  KernFast<D> kf(p.gps[0].kern);
  double kdiag = p.gps[0].kern.kdiag;
  stage_issue<D, NW>(cur, gv_next, lds, tid);
  __syncthreads();

  // per-GP state
  double xs[D];
  double sq[4] = {0.0, 0.0, 0.0, 0.0}, mean = 0.0;
  double4_t acc[kIB];
#pragma unroll
  for (int b = 0; b < kIB; ++b) acc[b] = double4_t{0.0, 0.0, 0.0, 0.0};
  // per-tile state of the row epilogue (confidence sweep / swarm fitness)
  bool safe = true;
  double l0 = 0.0, values = 0.0, interest = 1.0, total_pen = 0.0, lower = 0.0;

  // this lane's slab of the covariance cache
  typedef double double2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) double2_t* g2ptr_t;
  // (CACHE is a template parameter: the single-chunk kernel, n <= 256, carries
  // none of this)
  g2ptr_t kslab = nullptr;
  if (CACHE)
    kslab = (g2ptr_t)(p.kvc + ((int64_t(blockIdx.x) * kWaves + wave) *
                               int64_t(p.kvc_blocks)) * 256 + lane * 4);
  double kvn[4] = {0.0, 0.0, 0.0, 0.0};   // values fetched for the next stage
  bool cached = false;                     // ... which is a re-visited j-block

  int bufsel = 0;
  bool more = true;
#ifdef SGP_INSTRUMENT
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
  const unsigned long long stamp_first = stamp_last;
#endif
#pragma unroll 1
  while (more) {
    SGP_STAMP(0)   // loop back edge
    if (cur.c == 0 && cur.jb == 0) kf.prep(x, xs);

    double* cbuf = lds + bufsel * kBuf;
    double* nbuf = lds + (bufsel ^ 1) * kBuf;

    // prefetch the next stage (and the rows of the next tile)
    StagePos nxt = cur;
    more = stage_next(nxt, p.gps, Geff, ntiles, gridDim.x);
    const bool tile_ends = !more || nxt.tile != cur.tile;
    const bool gp_ends = tile_ends || nxt.g != cur.g;
    if (more) {
      if (gp_ends && Geff > 1) gv_next.load(p.gps[nxt.g]);
      if (!SGP_ABL(2)) stage_issue<D, NW>(nxt, gv_next, nbuf, tid);
    }
    const bool chunk_ends = gp_ends || nxt.c != cur.c;
    if (more && tile_ends) load_x(nxt.tile, xnext);
    // a j-block below the diagonal part of its chunk was evaluated (and parked)
    // while an earlier chunk of this (tile, GP) pass was processed
    const bool was_cached = CACHE && cached;
    double kvc_cur[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kvc_cur[q] = kvn[q];
    cached = CACHE && more && nxt.jb < nxt.b0;
    if (cached) {
      const double2_t a = __builtin_nontemporal_load(kslab + nxt.jb * 128);
      const double2_t b = __builtin_nontemporal_load(kslab + nxt.jb * 128 + 1);
      kvn[0] = a.x; kvn[1] = a.y; kvn[2] = b.x; kvn[3] = b.y;
    }

    SGP_STAMP(1)   // bookkeeping + DMA issue + cache traffic
    // this stage: 16 training points against the active row blocks
    const double* xT = cbuf + kATile;
    const double* alT = cbuf + kATile + kXTile;
    double kv[4];
    if (was_cached) {
#pragma unroll
      for (int q = 0; q < 4; ++q) kv[q] = kvc_cur[q];
    } else if (!SGP_ABL(4)) {
      kf.template many<4>(xs, xT + (lane >> 4) * D, 4 * D, tab, kv);
      if (CACHE && cur.c + 1 < cur.nchunks) {       // needed again by later chunks
        double2_t a, b;
        a.x = kv[0]; a.y = kv[1]; b.x = kv[2]; b.y = kv[3];
        __builtin_nontemporal_store(a, kslab + cur.jb * 128);
        __builtin_nontemporal_store(b, kslab + cur.jb * 128 + 1);
      }
    } else {
      kv[0] = xs[0]; kv[1] = xs[0] + 1.0; kv[2] = xs[0] + 2.0; kv[3] = xs[0] + 3.0;
    }
    if (cur.c == cur.nchunks - 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        mean = fma(alT[q * 4 + (lane >> 4)], kv[q], mean);
    }
    SGP_STAMP(2)   // covariance evaluation + mean
    if (!SGP_ABL(8))
      mfma_jblock(cur.shift + max(0, cur.jb - cur.b0), acc, cbuf + lane, kv);
    SGP_STAMP(3)   // swizzles + MFMAs

    if (chunk_ends) {
#pragma unroll
      for (int b = 0; b < kIB; ++b) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          sq[m] = fma(acc[b][m], acc[b][m], sq[m]);
          acc[b][m] = 0.0;
        }
      }
    }

    if (gp_ends && !SGP_ABL(32)) {
      // sq[m]: partial sums for column 4m + (lane & 3) over this lane's rows.
      // Fold the 16 lanes that share (lane & 3), then pick the quad of this
      // lane's own column (lane & 15) = 4 ((lane >> 2) & 3) + (lane & 3).
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double v = sq[m];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        sq[m] = v;
      }
      const int mq = (lane >> 2) & 3;
      const double sumsq =
          (mq == 0) ? sq[0] : (mq == 1) ? sq[1] : (mq == 2) ? sq[2] : sq[3];
      const double mu = sum_lane_groups(mean);
      const double var = fmax(kdiag - sumsq, 1e-15);  // GPy clip
      const double sd = sqrt(var);
      sq[0] = sq[1] = sq[2] = sq[3] = 0.0;
      mean = 0.0;

      const int g = cur.g;
      const int64_t row = cur.tile * kTilePts + wave * 16 + (lane & 15);
      const bool writer = (row < p.pts.N) && (lane < 16);
      if (conf) {
        // update_confidence_intervals + compute_safe_set (gp_opt.py:453-481)
        const double lo = mu - p.conf.beta * sd;
        const double up = mu + p.conf.beta * sd;
        if (g == 0) l0 = lo;
        safe = safe && (lo > p.conf.fmin[g]);
        if (writer && !SGP_ABL(16)) {
          p.conf.mean[int64_t(g) * p.pts.N + row] = mu;
          p.conf.var[int64_t(g) * p.pts.N + row] = var;
          if (p.conf.Q) {
            const double2 q = make_double2(lo, up);
            *reinterpret_cast<double2*>(p.conf.Q + (row * p.G + g) * 2) = q;
          }
        }
      } else {
        // SafeOptSwarm._compute_particle_fitness, gp_opt.py:925-1013
        const FitnessArgs& f = p.fit;
        lower = mu - f.beta * sd;
        if (g == 0) {
          values = sd / f.scaling[0];
          if (st == SGP_SWARM_EXPANDERS) interest = double(p.G);
          if (st == SGP_SWARM_MAXIMIZERS) {
            const double upper = mu + f.beta * sd;
            const double z = 10.0 * (upper - f.best_lower_bound) / f.scaling[0];
            interest = 1.0 / (1.0 + exp(-z));  // scipy.special.expit
          }
        } else {
          values = fmax(values, sd / f.scaling[g]);
        }
        if (f.fmin[g] != -INFINITY) {
          double slack = lower - f.fmin[g];
          safe = safe && (slack >= 0.0);
          if (st != SGP_SWARM_SAFE_SET) {
            slack = slack / f.scaling[g];
            total_pen += swarm_penalty(slack);
            if (st == SGP_SWARM_EXPANDERS) {
              // scipy.stats.norm.pdf(slack, scale=0.2)
              const double z = slack / 0.2;
              interest *= exp(-0.5 * z * z) / 2.5066282746310002 / 0.2;
            }
          }
        }
      }

      if (tile_ends) {
        if (conf) {
          if (p.conf.S) {
            if (writer) p.conf.S[row] = safe ? 1 : 0;
            // maximum of l0 over the safe rows of the tile -> one partial
            double v = (writer && safe) ? l0 : -INFINITY;
            v = wave_max(v);
            if (lane == 0) red[wave] = v;
            __syncthreads();
            if (tid == 0) {
              double m = red[0];
#pragma unroll
              for (int w = 1; w < kWaves; ++w) m = fmax(m, red[w]);
              p.conf.partial[cur.tile] = m;
            }
          }
        } else if (writer) {
          double out;
          bool ok = safe;
          if (st == SGP_SWARM_GREEDY) {
            out = lower;
            ok = true;
          } else if (st == SGP_SWARM_SAFE_SET) {
            out = lower;
          } else {
            out = (values + total_pen) * interest;
          }
          p.fit.values[row] = out;
          p.fit.safe[row] = ok ? 1 : 0;
        }
        safe = true;
        l0 = values = total_pen = lower = 0.0;
        interest = 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = xnext[k];
      }
    }

    if (gp_ends && more && Geff > 1) {   // hyper-parameters of the next GP
      kf = KernFast<D>(p.gps[nxt.g].kern);
      kdiag = p.gps[nxt.g].kern.kdiag;
    }
    SGP_STAMP(4)   // chunk fold + epilogue
    if (!SGP_ABL(1)) __syncthreads();
    SGP_STAMP(5)   // barrier
    bufsel ^= 1;
    cur = nxt;
  }
#ifdef SGP_INSTRUMENT
  if (p.stamps && lane == 0) {
    unsigned long long* o = p.stamps + (int64_t(blockIdx.x) * kWaves + wave) * 8;
    for (int k = 0; k < 6; ++k) o[k] = stamp_acc[k];
    o[6] = stamp_last - stamp_first;
    o[7] = stamp_first;
  }
#endif
}

// ---- expander check ---------------------------------------------------------
// One MFMA row block = up to 16 candidates: acc[cand, pt] = sum_j w_c[j] K[j,pt].
// The 16 rows of a wave are row0 + (lane & 15); lane >> 4 selects the candidate
// quad.  `unsafe` marks the lanes whose row takes part.
template <int D>
__device__ __forceinline__ void expander_rows(const GpDev* gps, int G,
                                              const SweepPoints& pts,
                                              const ExpanderArgs& ea,
                                              int64_t rrow, bool unsafe,
                                              const double* tab, int lane) {
  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k)
    x[k] = pts.base[rrow * pts.stride_row + k * pts.stride_col];

  for (int g = 0; g < G; ++g) {
    if (!ea.active[g]) continue;
    const GpDev& gp = gps[g];
    const KernFast<D> kf(gp.kern);
    const double mu = ea.mean[int64_t(g) * pts.N + rrow];
    const double var = ea.var[int64_t(g) * pts.N + rrow];
    const double kdiag = gp.kern.kdiag;

    // Exact pre-filter.  |c(x)| <= k(x,x_c) + |L^-1 k_x| |L^-1 k_c| with
    // |L^-1 k_x|^2 = k(x,x) - var(x), so an upper bound of the updated lower
    // confidence bound costs ONE covariance evaluation per (row, candidate)
    // instead of n.  Rows far from x_c and from the data (most of the unsafe
    // set) cannot reach fmin and skip the n-term contraction below.
    double kxc[4];
    bool possible = false;
    const double qx = fmax(kdiag - var, 0.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cand = (lane >> 4) + 4 * r;
      kxc[r] = 0.0;
      if (cand < ea.m && unsafe) {
        kxc[r] = kf.raw(x, ea.xc + cand * D, tab);
        const double cmax =
            (fabs(kxc[r]) + sqrt(qx * ea.tn2[g * 16 + cand])) * (1.0 + 1e-9);
        const double mu2 = mu + fabs(ea.delta[g * 16 + cand]) * cmax;
        const double var2 =
            fmax(var - cmax * cmax * ea.inv_s2[g * 16 + cand], 1e-15);
        const double l2max = mu2 - ea.beta * sqrt(var2);
        possible = possible ||
                   ((l2max + 1e-9 * (fabs(mu2) + 1.0) >= ea.fmin[g]) &&
                    (kxc[r] >= ea.near_frac * kdiag));
      }
    }
    if (__ballot(possible) == 0ull) continue;  // wave-uniform

    gptr_t W = (gptr_t)ea.Wpack + int64_t(g) * ea.wstride + lane;
    gptr_t Xj = (gptr_t)gp.Xs + (lane >> 4) * D;
    double xs[D];
    kf.prep(x, xs);
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const int nsteps = gp.n_pad >> 2;  // multiple of 4 (n_pad is 16-aligned)
    // operands of 16 training points per iteration, fetched one iteration ahead
    // (the loop is otherwise a chain of exposed global-memory latencies)
    double a[4], xr[4][D], an[4], xn[4][D];
    auto fetch = [&](int s0, double (&ao)[4], double (&xo)[4][D]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ao[q] = W[(s0 + q) * 64];
#pragma unroll
        for (int k = 0; k < D; ++k) xo[q][k] = Xj[(s0 + q) * 4 * D + k];
      }
    };
    fetch(0, a, xr);
#pragma unroll 1
    for (int s0 = 0; s0 < nsteps; s0 += 4) {
      if (s0 + 4 < nsteps) fetch(s0 + 4, an, xn);
      double kv[4];
      kf.template many<4>(xs, &xr[0][0], D, tab, kv);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], kv[q], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = an[q];
#pragma unroll
        for (int k = 0; k < D; ++k) xr[q][k] = xn[q][k];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 r
      const int cand = (lane >> 4) + 4 * r;
      bool hit = false;
      if (cand < ea.m && unsafe) {
        const double cx = kxc[r] - acc[r];
        const double mu2 = mu + cx * ea.delta[g * 16 + cand];
        const double var2 =
            fmax(var - cx * cx * ea.inv_s2[g * 16 + cand], 1e-15);
        const double l2 = mu2 - ea.beta * sqrt(var2);
        hit = l2 >= ea.fmin[g];
      }
      const unsigned long long b = __ballot(hit);
      if (lane == 0 && b != 0ull) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
          if ((b >> (16 * grp)) & 0xffffull)
            atomicOr(&ea.flags[(grp + 4 * r) * G + g], 1);
        }
      }
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_expander(const GpDev* gps, int G,
                                                  SweepPoints pts,
                                                  ExpanderArgs ea) {
  __shared__ double tab[kExpTabSize];
  exp_tab_init(tab);
  __syncthreads();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t row = int64_t(blockIdx.x) * 64 + wave * 16 + (lane & 15);
  const bool valid = row < pts.N;
  const int64_t rrow = valid ? row : pts.N - 1;
  const bool unsafe = valid && (ea.S[rrow] == 0);
  // skip waves with no unsafe row (wave-uniform)
  if (__ballot(unsafe) == 0ull) return;
  expander_rows<D>(gps, G, pts, ea, rrow, unsafe, tab, lane);
}

// Single candidate (the probe of the first candidate and its exact re-scan):
// the pre-filter runs with one row per lane over the whole shard and appends
// the rows that could be lifted above fmin to a list (a few per cent of the
// unsafe set); k_expander_list then computes c(x) = k(x,x_c) - w . k(X,x) for
// those rows only, one row per lane and the training rows / w through LDS --
// with a single candidate the contraction is a dot product, not a matrix
// product.  Same pre-filter and the same update formulas as k_expander.
template <int D>
__global__ __launch_bounds__(256) void k_expander_filter(const GpDev* gps, int G,
                                                         SweepPoints pts,
                                                         ExpanderArgs ea,
                                                         int* count, int* list) {
  __shared__ double tab[kExpTabSize];
  exp_tab_init(tab);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const bool valid = row < pts.N;
  const int64_t rrow = valid ? row : pts.N - 1;
  const bool unsafe = valid && (ea.S[rrow] == 0);
  if (__ballot(unsafe) == 0ull) return;
  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k)
    x[k] = pts.base[rrow * pts.stride_row + k * pts.stride_col];
  bool possible = false;
  if (unsafe) {
    for (int g = 0; g < G; ++g) {
      if (!ea.active[g]) continue;
      const GpDev& gp = gps[g];
      const KernFast<D> kf(gp.kern);
      const double mu = ea.mean[int64_t(g) * pts.N + rrow];
      const double var = ea.var[int64_t(g) * pts.N + rrow];
      const double kdiag = gp.kern.kdiag;
      const double qx = fmax(kdiag - var, 0.0);
      const double kxc = kf.raw(x, ea.xc, tab);
      const double cmax =
          (fabs(kxc) + sqrt(qx * ea.tn2[g * 16])) * (1.0 + 1e-9);
      const double mu2 = mu + fabs(ea.delta[g * 16]) * cmax;
      const double var2 = fmax(var - cmax * cmax * ea.inv_s2[g * 16], 1e-15);
      const double l2max = mu2 - ea.beta * sqrt(var2);
      possible = possible ||
                 ((l2max + 1e-9 * (fabs(mu2) + 1.0) >= ea.fmin[g]) &&
                  (kxc >= ea.near_frac * kdiag));
    }
  }
  const unsigned long long b = __ballot(possible);
  if (b == 0ull) return;
  int at = 0;
  if (lane == 0) at = atomicAdd(count, __popcll(b));
  at = __builtin_amdgcn_readfirstlane(at);
  if (possible) list[at + __popcll(b & ((1ull << lane) - 1ull))] = int(row);
}

constexpr int kExpLds = 6144;     // doubles of staged training data (48 KB)

template <int D>
__global__ __launch_bounds__(256) void k_expander_list(const GpDev* gps, int G,
                                                       SweepPoints pts,
                                                       ExpanderArgs ea,
                                                       const int* count,
                                                       const int* list) {
  __shared__ double tab[kExpTabSize];
  __shared__ double stage[kExpLds];      // [n_pad][D] scaled rows | [n_pad] w
  exp_tab_init(tab);
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63;
  const int nrows = *count;
  // 16 rows per wave (lane & 15); the four 16-lane groups split the training
  // points (j = 4 s + (lane >> 4)) and fold their partial dot products at the end
  const int first = (blockIdx.x * 4 + (tid >> 6)) * 16, stride = gridDim.x * 64;
  for (int g = 0; g < G; ++g) {
    if (!ea.active[g]) continue;
    const GpDev& gp = gps[g];
    const KernFast<D> kf(gp.kern);
    const int np = gp.n_pad;
    // w_j of the single candidate sits in lane 16 (j & 3) of k-step j >> 2 of
    // the packed operand
    const double* Wp = ea.Wpack + int64_t(g) * ea.wstride;
    const bool staged = np * (D + 1) <= kExpLds;        // block-uniform
    const double* Xj = gp.Xs;
    if (staged) {
      __syncthreads();                                   // previous GP's readers
      for (int e = tid; e < np * D; e += 256) stage[e] = gp.Xs[e];
      for (int j = tid; j < np; j += 256)
        stage[np * D + j] = Wp[(j >> 2) * 64 + (j & 3) * 16];
      __syncthreads();
      Xj = stage;
    }
    const double kdiag = gp.kern.kdiag;
    for (int i0 = first; i0 < nrows; i0 += stride) {
      const bool valid = i0 + (lane & 15) < nrows;
      const int64_t rrow = list[valid ? i0 + (lane & 15) : i0];
      double x[D], xs[D];
#pragma unroll
      for (int k = 0; k < D; ++k)
        x[k] = pts.base[rrow * pts.stride_row + k * pts.stride_col];
      kf.prep(x, xs);
      const double mu = ea.mean[int64_t(g) * pts.N + rrow];
      const double var = ea.var[int64_t(g) * pts.N + rrow];
      double dot = 0.0;
      const int ph = lane >> 4;
#pragma unroll 1
      for (int s0 = 0; s0 < (np >> 2); s0 += 4) {   // 16 training points / step
        double kq[4], wq[4];
        kf.template many<4>(xs, Xj + (s0 * 4 + ph) * D, 4 * D, tab, kq);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          wq[q] = staged ? stage[np * D + (s0 + q) * 4 + ph]
                         : Wp[(s0 + q) * 64 + ph * 16];
#pragma unroll
        for (int q = 0; q < 4; ++q) dot = fma(wq[q], kq[q], dot);
      }
      dot = sum_lane_groups(dot);
      const double kxc = kf.raw(x, ea.xc, tab);
      bool hit = false;
      if (valid && kxc >= ea.near_frac * kdiag) {
        const double cx = kxc - dot;
        const double mu2 = mu + cx * ea.delta[g * 16];
        const double var2 = fmax(var - cx * cx * ea.inv_s2[g * 16], 1e-15);
        hit = mu2 - ea.beta * sqrt(var2) >= ea.fmin[g];
      }
      if (__ballot(hit) != 0ull && lane == 0) atomicOr(&ea.flags[g], 1);
    }
  }
}

// ---- rank-1 update of the resident posterior --------------------------------------
// After one appended observation (x*, y*) the posterior at every row changes by
// a closed form (the same algebra as the expander test); per row and updated
// GP this is n covariance evaluations and n FMAs on the VALU -- no n^2 term.
// Lane (r, q) = (lane & 15, lane >> 4) handles row r and training points
// j = q (mod 4); the four partial dot products are folded with two shuffles.
constexpr int kRank1Lds = 6144;   // doubles of staged training data (48 KB)

template <int D>
__global__ __launch_bounds__(256) void k_rank1(const GpDev* gps, int G,
                                               SweepPoints pts, Rank1Args ra) {
  __shared__ double tab[kExpTabSize];
  __shared__ double red[4];
  __shared__ double stage[kRank1Lds];   // [n_pad][D] scaled rows | [n_pad] w
  exp_tab_init(tab);
  __syncthreads();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t row = int64_t(blockIdx.x) * 64 + wave * 16 + (lane & 15);
  const bool valid = row < pts.N;
  const int64_t rrow = valid ? row : pts.N - 1;
  const bool writer = valid && (lane < 16);

  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k)
    x[k] = pts.base[rrow * pts.stride_row + k * pts.stride_col];

  bool safe = true;
  double l0 = 0.0;
  for (int g = 0; g < G; ++g) {
    double mean = ra.mean[int64_t(g) * pts.N + rrow];
    double var = ra.var[int64_t(g) * pts.N + rrow];
    if (ra.which[g]) {
      const GpDev& gp = gps[g];
      const KernFast<D> kf(gp.kern);
      double xs[D];
      kf.prep(x, xs);
      // training rows and the update vector: through LDS when they fit (one
      // coalesced pass per workgroup instead of a global round trip per step)
      const int np = gp.n_pad;
      const bool staged = np * (D + 1) <= kRank1Lds;      // block-uniform
      const double* Xj = gp.Xs + (lane >> 4) * D;
      const double* w = gp.upd_w + (lane >> 4);
      if (staged) {
        __syncthreads();                                   // previous GP's readers
        for (int e = tid; e < np * D; e += 256) stage[e] = gp.Xs[e];
        for (int e = tid; e < np; e += 256) stage[np * D + e] = gp.upd_w[e];
        __syncthreads();
        Xj = stage + (lane >> 4) * D;
        w = stage + np * D + (lane >> 4);
      }
      double dot = 0.0;
      const int nsteps = np >> 2;
#pragma unroll 1
      for (int s = 0; s < nsteps; s += 4) {   // n_pad is a multiple of 16
        double kq[4];
        kf.template many<4>(xs, Xj + s * 4 * D, 4 * D, tab, kq);
#pragma unroll
        for (int q = 0; q < 4; ++q) dot = fma(w[(s + q) * 4], kq[q], dot);
      }
      dot = sum_lane_groups(dot);
      const double cx = kf.raw(x, gp.upd + 2, tab) - dot;
      mean = fma(cx, gp.upd[0], mean);
      var = fmax(var - cx * cx * gp.upd[1], 1e-15);
      if (writer) {
        ra.mean[int64_t(g) * pts.N + row] = mean;
        ra.var[int64_t(g) * pts.N + row] = var;
      }
    }
    const double sd = sqrt(var);
    const double lo = mean - ra.beta * sd;
    const double up = mean + ra.beta * sd;
    if (g == 0) l0 = lo;
    safe = safe && (lo > ra.fmin[g]);
    if (writer) {
      const double2 q = make_double2(lo, up);
      *reinterpret_cast<double2*>(ra.Q + (row * G + g) * 2) = q;
    }
  }
  if (writer) ra.S[row] = safe ? 1 : 0;
  double v = (writer && safe) ? l0 : -INFINITY;
  v = wave_max(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (tid == 0)
    ra.partial[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ---- operand-layout probe (test hook): one MFMA with caller-given lane values ------
__global__ void k_probe_mfma(int which, const double* a, const double* b,
                             const double* c, double* d) {
  const int l = threadIdx.x;
  if (which == 0) {
    double4_t acc = {c[l * 4], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
  } else {
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
  }
}

// ---- fp64 issue-rate microbenchmarks ---------------------------------------------
// MODE 6: 16 chains of v_mfma_f64_4x4x4_4b_f64 (512 flop each)
// MODE 0: 8 independent MFMA chains   1: 4 chains   2: 8 chains + 8 v_fma_f64
// per MFMA   3: v_fma_f64 only (16 chains)   4: 8 MFMA chains + 2 v_fma_f64 per
// MFMA   5: 16 MFMA chains
template <int MODE>
__global__ __launch_bounds__(256) void k_mfma_bench(double* out, int iters) {
  extern __shared__ double dyn_lds[];  // only limits residency
  constexpr int NA = (MODE == 1) ? 4 : (MODE == 5 ? 16 : 8);
  double4_t acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-6;
  const double av = 1.0 + threadIdx.x * 1e-9, bv = 1.0 - threadIdx.x * 1e-9;
  const double m = 0.999999, c = 1e-7;
  double a4[4], b16[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) a4[q] = av + q * 1e-3;
#pragma unroll
  for (int j = 0; j < 16; ++j) b16[j] = bv + j * 1e-3;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 9) {
      // the sweep's operand pattern: 4 A registers, 16 B registers, 16
      // accumulators -- every MFMA reads a different (A, B, C) triple
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          f[4 * q + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(
              a4[q], b16[4 * q + m], f[4 * q + m], 0, 0, 0);
    } else if (MODE == 6 || MODE == 7 || MODE == 8) {
      // 16 / 4 / 2 independent chains, 16 instructions per iteration
      constexpr int NC = (MODE == 6) ? 16 : (MODE == 7 ? 4 : 2);
#pragma unroll
      for (int rep = 0; rep < 16 / NC; ++rep)
#pragma unroll
        for (int j = 0; j < NC; ++j)
          f[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, f[j], 0, 0, 0);
    } else if (MODE != 3) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fma(f[j], m, c);
        }
        if (MODE == 4) {
          f[(2 * i) & 15] = fma(f[(2 * i) & 15], m, c);
          f[(2 * i + 1) & 15] = fma(f[(2 * i + 1) & 15], m, c);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) f[j] = fma(f[j], m, c);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += f[i];
  if (iters < 0) s += dyn_lds[threadIdx.x];   // keeps the LDS allocation referenced
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// One stage of the sweep's inner loop in isolation (scripts/microbench.py):
//   STAGE 0  mfma_jblock only (B-operand swizzles, slot guards, A operand reads)
//   STAGE 1  + the stage barrier
//   STAGE 2  + the LDS-DMA of the next A chunk (8 x 1 KB per wave) + barrier
//   STAGE 3  + the 4 covariance evaluations per lane (RBF, d = 2)
//   STAGE 4  covariance evaluations + mfma_jblock, no DMA, no barrier
// `lo` = first active accumulator slot (0: all 16 slots, 9: config 2's average).
template <int STAGE>
__global__ __launch_bounds__(256, 2) void k_stage_bench(const double* src,
                                                        double* out, int iters,
                                                        int lo) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* tab = lds + kTabOff;
  exp_tab_init(lds + kTabOff);
  for (int i = tid; i < 2 * kBuf; i += 256) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  double4_t acc[kIB];
#pragma unroll
  for (int b = 0; b < kIB; ++b) acc[b] = double4_t{0.0, 0.0, 0.0, 0.0};
  double kv[4] = {1.0 + lane * 1e-3, 1.1, 1.2 - lane * 1e-3, 1.3};
  const double xs[2] = {lane * 0.01, 0.3 + blockIdx.x * 1e-4};
  GpView gv;
  gv.Apack = (gptr_t)src;
  gv.Xs = (gptr_t)src;
  gv.alpha = (gptr_t)src;
  gv.nsteps_total = kSteps;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    double* cbuf = lds + (it & 1) * kBuf;
    double* nbuf = lds + ((it & 1) ^ 1) * kBuf;
    if (STAGE == 2 || STAGE == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int piece = wave + 4 * k;
        if ((piece >> 1) >= lo) {
          const double* g = src + piece * 128 + lane * 2;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)g,
              (__attribute__((address_space(3))) void*)(nbuf + piece * 128), 16,
              0, 0);
        }
      }
    }
    if (STAGE >= 3) {
      const double* xT = cbuf + kATile + (lane >> 4) * 2;
      double r2[4], u[4], e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double t0 = xs[0] - xT[q * 8], t1 = xs[1] - xT[q * 8 + 1];
        r2[q] = fma(t1, t1, t0 * t0);
        u[q] = -r2[q];
      }
      exp2_32x4(u, tab, e);
#pragma unroll
      for (int q = 0; q < 4; ++q) kv[q] = 2.0 * e[q];
    }
    mfma_jblock(lo, acc, cbuf + lane, kv);
    if (STAGE >= 1 && STAGE <= 3) __syncthreads();
  }
  double s = 0;
#pragma unroll
  for (int b = 0; b < kIB; ++b)
    for (int r = 0; r < 4; ++r) s += acc[b][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int D, int NW, int MODE, bool CACHE>
int launch_sweep_v(sgp_ctx* ctx, const SweepParams& p, double flops) {
  static bool attr_set = false;
  if (!attr_set) {
    SGP_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(&k_sweep<D, NW, MODE, CACHE>),
                     hipFuncAttributeMaxDynamicSharedMemorySize,
                     int(kLdsBytes)));
    attr_set = true;
  }
  const int tile = 16 * NW;
  const int64_t ntiles = (p.pts.N + tile - 1) / tile;
  // persistent: as many workgroups as are resident at once (256 VGPRs per
  // thread -> 8 waves per CU) walk over the tiles
  const int64_t resident = int64_t(ctx->num_cu) * (kMaxWaves / NW);
  const int nblocks = int(ntiles < resident ? ntiles : resident);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profiling) {
    if (ctx->prof_used + 2 > ctx->prof_events.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        SGP_HIP(ctx, hipEventCreate(&e));
        ctx->prof_events.push_back(e);
      }
    }
    e0 = ctx->prof_events[ctx->prof_used];
    e1 = ctx->prof_events[ctx->prof_used + 1];
    ctx->prof_used += 2;
    ctx->prof_flops += flops;
    SGP_HIP(ctx, hipEventRecord(e0, ctx->stream));
  }
  SweepParams pp = p;
#ifdef SGP_INSTRUMENT
  static const int ablate = getenv("SGP_ABLATE") ? atoi(getenv("SGP_ABLATE")) : 0;
  pp.ablate = ablate;
  static const int skew = getenv("SGP_SKEW") ? atoi(getenv("SGP_SKEW")) : 0;
  static int* cu_count = nullptr;
  pp.skew = skew;
  if (skew > 0) {
    if (!cu_count) SGP_HIP(ctx, hipMalloc(&cu_count, 2048 * sizeof(int)));
    SGP_HIP(ctx, hipMemsetAsync(cu_count, 0, 2048 * sizeof(int), ctx->stream));
  }
  pp.cu_count = cu_count;
  static const bool want_stamps = getenv("SGP_STAMPS") != nullptr;
  static unsigned long long* stamps = nullptr;
  if (want_stamps && !stamps)
    SGP_HIP(ctx, hipMalloc(&stamps, size_t(nblocks) * NW * 8 * 8));
  pp.stamps = stamps;
#endif
  hipLaunchKernelGGL((k_sweep<D, NW, MODE, CACHE>), dim3(nblocks), dim3(64 * NW),
                     kLdsBytes, ctx->stream, pp);
  SGP_HIP(ctx, hipGetLastError());
  if (e1) SGP_HIP(ctx, hipEventRecord(e1, ctx->stream));
#ifdef SGP_INSTRUMENT
  if (pp.stamps) {   // per-phase cycle totals, averaged over all waves
    std::vector<unsigned long long> h(size_t(nblocks) * NW * 8);
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGP_HIP(ctx, hipMemcpy(h.data(), pp.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double tot[8] = {0};
    unsigned long long first = ~0ull, last = 0;
    for (size_t w = 0; w < h.size() / 8; ++w) {
      for (int k = 0; k < 7; ++k) tot[k] += double(h[w * 8 + k]);
      first = std::min(first, h[w * 8 + 7]);
      last = std::max(last, h[w * 8 + 7] + h[w * 8 + 6]);
    }
    const double nw = double(h.size() / 8);
    fprintf(stderr, "stamps (cycles/wave): back %.0f  book+dma %.0f  eval %.0f  "
            "mfma %.0f  epilogue %.0f  barrier %.0f  | loop %.0f  span %.0f\n",
            tot[0] / nw, tot[1] / nw, tot[2] / nw, tot[3] / nw, tot[4] / nw,
            tot[5] / nw, tot[6] / nw, double(last - first));
  }
#endif
  return 0;
}

template <int D>
int launch_sweep_d(sgp_ctx* ctx, const SweepParams& p, double flops) {
  if (p.mode == MODE_CONF)
    return p.kvc ? launch_sweep_v<D, kSweepWaves, MODE_CONF, true>(ctx, p, flops)
                 : launch_sweep_v<D, kSweepWaves, MODE_CONF, false>(ctx, p, flops);
  return p.kvc ? launch_sweep_v<D, kSweepWaves, MODE_FITNESS, true>(ctx, p, flops)
               : launch_sweep_v<D, kSweepWaves, MODE_FITNESS, false>(ctx, p, flops);
}

int launch_sweep(sgp_ctx* ctx, const SweepParams& p, const GpDev* gh, int d) {
  // algorithmic flops: G * (n^2 + 2n) per row
  double flops = 0.0;
  const int Geff =
      (p.mode == MODE_FITNESS && p.fit.swarm_type == SGP_SWARM_GREEDY) ? 1
                                                                       : p.G;
  for (int g = 0; g < Geff; ++g)
    flops += (double(gh[g].n) * gh[g].n + 2.0 * gh[g].n) * double(p.pts.N);
  if (p.pts.N <= 0) return 0;
  // covariance cache: j-blocks of all but the last chunk, per resident wave
  SweepParams q = p;
  q.kvc = nullptr;
  q.kvc_blocks = 0;
  for (int g = 0; g < Geff; ++g) {
    const int nchunks = (gh[g].nblk + kIB - 1) / kIB;
    q.kvc_blocks = std::max(q.kvc_blocks, (nchunks - 1) * kIB);
  }
  // (SGP_NO_KVCACHE=1 re-evaluates instead -- the A/B switch behind the numbers
  // in profiles/README.md)
  if (q.kvc_blocks > 0 && !getenv("SGP_NO_KVCACHE")) {
    const int64_t tiles = (p.pts.N + 16 * kSweepWaves - 1) / (16 * kSweepWaves);
    const int64_t wgs = std::min<int64_t>(tiles, int64_t(ctx->num_cu) *
                                                     (kMaxWaves / kSweepWaves));
    q.kvc = static_cast<double*>(sgp_scratch(
        ctx, 0, size_t(wgs) * kSweepWaves * q.kvc_blocks * 256 * sizeof(double)));
    if (!q.kvc) return -1;
  }
  switch (d) {
    case 1: return launch_sweep_d<1>(ctx, q, flops);
    case 2: return launch_sweep_d<2>(ctx, q, flops);
    case 3: return launch_sweep_d<3>(ctx, q, flops);
    case 4: return launch_sweep_d<4>(ctx, q, flops);
    case 5: return launch_sweep_d<5>(ctx, q, flops);
    case 6: return launch_sweep_d<6>(ctx, q, flops);
    case 7: return launch_sweep_d<7>(ctx, q, flops);
    case 8: return launch_sweep_d<8>(ctx, q, flops);
  }
  sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
  return -2;
}

}  // namespace

int sweep_num_blocks(int64_t N) {   // = number of tiles = number of partials
  constexpr int t = 16 * kSweepWaves;
  return int((N + t - 1) / t);
}

int launch_sweep_conf(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                      int G, int d, SweepPoints pts, ConfOut out) {
  SweepParams p;
  p.gps = gps_dev;
  p.G = G;
  p.mode = MODE_CONF;
  p.pts = pts;
  p.conf = out;
  p.fit = FitnessArgs{};
  return launch_sweep(ctx, p, gps_host, d);
}

int launch_sweep_fitness(sgp_ctx* ctx, const GpDev* gps_dev,
                         const GpDev* gps_host, int G, int d, SweepPoints pts,
                         FitnessArgs fa) {
  SweepParams p;
  p.gps = gps_dev;
  p.G = G;
  p.mode = MODE_FITNESS;
  p.pts = pts;
  p.conf = ConfOut{};
  p.fit = fa;
  return launch_sweep(ctx, p, gps_host, d);
}

int launch_fitness_small(sgp_ctx* ctx, int G, int64_t P, const double* mean,
                         const double* var, FitnessArgs fa) {
  hipLaunchKernelGGL(k_fitness_small, dim3(unsigned((P + 255) / 256)), dim3(256),
                     0, ctx->stream, G, P, mean, var, fa);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_expander_check(sgp_ctx* ctx, const GpDev* gps_dev,
                          const GpDev* gps_host, int G, int d, SweepPoints pts,
                          ExpanderArgs ea) {
  (void)gps_host;
  if (pts.N <= 0) return 0;
  const int nblocks = int((pts.N + 63) / 64);
  const bool listed = ea.m == 1 && ea.count && ea.list;
  const int nfilter = int((pts.N + 255) / 256), nlist = ctx->num_cu * 2;
#define EXP_CASE(DD)                                                          \
  case DD:                                                                    \
    if (listed) {                                                             \
      hipLaunchKernelGGL(k_expander_filter<DD>, dim3(nfilter), dim3(256), 0,  \
                         ctx->stream, gps_dev, G, pts, ea, ea.count, ea.list);\
      hipLaunchKernelGGL(k_expander_list<DD>, dim3(nlist), dim3(256), 0,      \
                         ctx->stream, gps_dev, G, pts, ea, ea.count, ea.list);\
    } else {                                                                  \
      hipLaunchKernelGGL(k_expander<DD>, dim3(nblocks), dim3(256), 0,         \
                         ctx->stream, gps_dev, G, pts, ea);                   \
    }                                                                         \
    break;
  switch (d) {
    EXP_CASE(1) EXP_CASE(2) EXP_CASE(3) EXP_CASE(4)
    EXP_CASE(5) EXP_CASE(6) EXP_CASE(7) EXP_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
#undef EXP_CASE
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

template <int MODE>
int run_microbench(sgp_ctx* ctx, int iters, int lds_bytes, int nblocks,
                   double* out, float* ms) {
  if (lds_bytes > 64 * 1024)
    SGP_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(&k_mfma_bench<MODE>),
                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipLaunchKernelGGL(k_mfma_bench<MODE>, dim3(nblocks), dim3(256), lds_bytes,
                     ctx->stream, out, 16);  // warm-up
  SGP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_mfma_bench<MODE>, dim3(nblocks), dim3(256), lds_bytes,
                     ctx->stream, out, iters);
  SGP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  SGP_HIP(ctx, hipEventSynchronize(ctx->ev1));
  SGP_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return 0;
}

// tflops[0] = MFMA flops rate, tflops[1] = VALU FMA flops rate
int launch_probe_mfma(sgp_ctx* ctx, int which, const double* a, const double* b,
                      const double* c, double* d) {
  hipLaunchKernelGGL(k_probe_mfma, dim3(1), dim3(64), 0, ctx->stream, which, a,
                     b, c, d);
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int rank1_num_blocks(int64_t N) { return int((N + 63) / 64); }

int launch_rank1(sgp_ctx* ctx, const GpDev* gps_dev, int G, int d,
                 SweepPoints pts, Rank1Args ra) {
  if (pts.N <= 0) return 0;
  const int nblocks = rank1_num_blocks(pts.N);
#define R1_CASE(DD)                                                           \
  case DD:                                                                    \
    hipLaunchKernelGGL(k_rank1<DD>, dim3(nblocks), dim3(256), 0, ctx->stream, \
                       gps_dev, G, pts, ra);                                  \
    break;
  switch (d) {
    R1_CASE(1) R1_CASE(2) R1_CASE(3) R1_CASE(4)
    R1_CASE(5) R1_CASE(6) R1_CASE(7) R1_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
#undef R1_CASE
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_microbench(sgp_ctx* ctx, int mode, int iters, int lds_bytes,
                      double* tflops) {
  const int nblocks = ctx->num_cu * 8;
  double* out = static_cast<double*>(
      sgp_scratch(ctx, 0, size_t(nblocks) * 256 * sizeof(double)));
  if (!out) return -1;
  float ms = 0.f;
  int na = 8, valu_per_it = 0;
  switch (mode) {
    case 0: SGP_TRY(run_microbench<0>(ctx, iters, lds_bytes, nblocks, out, &ms)); break;
    case 1: SGP_TRY(run_microbench<1>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 4; break;
    case 2: SGP_TRY(run_microbench<2>(ctx, iters, lds_bytes, nblocks, out, &ms)); valu_per_it = 64; break;
    case 3: SGP_TRY(run_microbench<3>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 0; valu_per_it = 64; break;
    case 4: SGP_TRY(run_microbench<4>(ctx, iters, lds_bytes, nblocks, out, &ms)); valu_per_it = 16; break;
    case 5: SGP_TRY(run_microbench<5>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 16; break;
    case 6: SGP_TRY(run_microbench<6>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 4; break;  // 16 x 512 flop
    case 7: SGP_TRY(run_microbench<7>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 4; break;
    case 8: SGP_TRY(run_microbench<8>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 4; break;
    case 9: SGP_TRY(run_microbench<9>(ctx, iters, lds_bytes, nblocks, out, &ms)); na = 4; break;
    case 20: case 21: case 22: case 23: case 24: {
      // stage probes: lds_bytes carries `lo`; two 4-wave workgroups per CU
      const int lo = lds_bytes, nb = ctx->num_cu * 2;
      double* src = static_cast<double*>(sgp_scratch(ctx, 1, size_t(kATile) * 8));
      if (!src) return -1;
      SGP_HIP(ctx, hipMemsetAsync(src, 0, size_t(kATile) * 8, ctx->stream));
#define STAGE_RUN(S)                                                            \
  SGP_HIP(ctx, hipFuncSetAttribute(                                             \
                   reinterpret_cast<const void*>(&k_stage_bench<S>),            \
                   hipFuncAttributeMaxDynamicSharedMemorySize, int(kLdsBytes)));\
  hipLaunchKernelGGL(k_stage_bench<S>, dim3(nb), dim3(256), kLdsBytes,          \
                     ctx->stream, src, out, 16, lo);                            \
  SGP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));                          \
  hipLaunchKernelGGL(k_stage_bench<S>, dim3(nb), dim3(256), kLdsBytes,          \
                     ctx->stream, src, out, iters, lo);                         \
  SGP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
      switch (mode) {
        case 20: STAGE_RUN(0) break;
        case 21: STAGE_RUN(1) break;
        case 22: STAGE_RUN(2) break;
        case 23: STAGE_RUN(3) break;
        default: STAGE_RUN(4) break;
      }
#undef STAGE_RUN
      SGP_HIP(ctx, hipEventSynchronize(ctx->ev1));
      SGP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      tflops[0] = double(nb) * 4.0 * iters * (16 - lo) * 16 * 512.0 /
                  (double(ms) * 1e-3) / 1e12;
      tflops[1] = double(ms) * 1e6 / iters;     // ns per stage
      return 0;
    }
    default: sgp_set_error(ctx, "unknown microbench mode %d", mode); return -2;
  }
  const double waves = double(nblocks) * 4.0;
  tflops[0] = waves * iters * na * 2048.0 / (double(ms) * 1e-3) / 1e12;
  tflops[1] = waves * iters * valu_per_it * 128.0 / (double(ms) * 1e-3) / 1e12;
  return 0;
}
