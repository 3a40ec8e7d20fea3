// posterior_sweep: the dominant kernel of the path.
//
// Replaces gp.predict_noiseless(self.inputs) + the Q update of
// SafeOpt.update_confidence_intervals (safeopt/gp_opt.py:453-476), the safe-set
// test (:478-481), the posterior half of SafeOptSwarm._compute_particle_fitness
// (:901-1013; the shaping is k_fitness_small on the sweep's mean / var, launch_sweep
// below) and the per-candidate re-prediction of the expander loop (:579-606).
// This file: the 4-wave kernel for factors up to 256 rows (config 2), the dispatch
// to the paired-wave kernel above that (sweep_pair.hip), the expander / rank-1 kernels.
//
// For a tile of candidate rows the kernel forms the covariance tile
// K[j, pt] = k(X_j, x_pt) ON THE FLY in registers and contracts it with
// A = L^-1 (lower triangular, pre-packed in MFMA operand order) on the fp64
// matrix cores (v_mfma_f64_4x4x4_4b_f64, see "matrix part" below):
//     var(pt)  = k(x,x) - || A K[:, pt] ||^2          (n^2 flops / row)
//     mean(pt) = alpha . K[:, pt]                      (2n flops / row)
// K (n x N doubles, 4 GB per GP at n = 500, N = 1e6) is never written to memory.
//
// Work split: workgroup = NW waves, wave w owns 16 rows (the MFMA N dimension);
// the waves of a workgroup share the staged A chunk through LDS (LDS-DMA,
// double buffered, one barrier per stage).  The rows of A are processed in
// chunks of 16 MFMA row blocks (256 rows) held in 16 accumulator slots per wave;
// the triangular structure is exploited at 16x16 block granularity.  The kernel
// is persistent; the (tile, GP, chunk, j-block) loop nest is flattened into one
// stage sequence described by a small table the HOST builds (StageEnt), so the
// per-stage bookkeeping is one scalar load and a few scalar ALU instructions.
//
// What sets the speed (profiles/r02/probes.txt): the fp64 MFMA occupies its SIMD
// for 16 cycles and a co-resident wave gets ONE issue slot per MFMA of its
// partner, of any instruction type.  Everything that is not an MFMA therefore
// costs wall time roughly in proportion to its instruction COUNT; the stage loop
// below is written for few instructions per MFMA:
//   * accumulator slots are ordered so that the active ones are a prefix
//     (slot s <-> row block bend-1-s): one compare + branch per executed slot;
//   * the covariance operand is broadcast to the four column quads through a
//     wave-private LDS transpose (2 stores + 8 16-byte loads instead of 32
//     swizzles);
//   * covariances needed again by a later accumulator chunk are re-evaluated,
//     not parked in global memory (a slab of 64-459 MB and ~12 GB of traffic per
//     launch in round 1 for 3-5 %).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "small_path.h"
#include "sweep_shared.h"
#include "sweep_slots.h"

namespace {

// 1: a wave's share of the next A chunk (slots w, w + 4, ..) is copied piecewise
// between the slots of the running stage instead of in one burst behind the barrier
// (the paired kernel's finding, profiles/r03/experiments.txt section 4)

constexpr int kJC = 16;                // training points per stage (one j-block)
constexpr int kSteps = kJC / 4;        // MFMA k-steps per stage

// LDS layout for SL accumulator slots per wave (SL x 16 rows of L^-1 per chunk):
//   [2][A chunk | training rows | alpha]  exp table  [NW] broadcast buffers
// SL = 16: 256 VGPRs, two waves per SIMD (two 4-wave workgroups per CU).  (32
// slots with one wave per SIMD and AGPRs, 8-wave workgroups and an explicit
// ping-pong of the two waves of a SIMD were measured and dropped:
// profiles/r02/experiments.txt, commits 1869720 and ad93c1b.)
// A wave's broadcast buffer holds 4 k-rows of 64 doubles, kKbRow doubles apart
// (an odd multiple of 16: the k-rows then sit on disjoint banks).  The padding
// between them doubles as the staging area of the wave's Q rows (kQCap doubles):
// 48 doubles per k-row up to d = 4 (Q rows of up to 6 GPs), 16 beyond (2 GPs) --
// what keeps two workgroups per CU inside 160 KB of LDS.
// R: alpha chunks of up to R riders behind the leader's (kSweepRide, see launch_posterior)
template <int SL, int D, int R = 0>
struct Lay {
  static constexpr int kATile = SL * kSteps * 64;           // doubles
  static constexpr int kXTile = kJC * D;
  static constexpr int kBuf = kATile + kXTile + kJC * (1 + R);   // + alpha chunk(s)
  static constexpr int kTabOff = 2 * kBuf;                  // exp table
  static constexpr int kKbOff = kTabOff + kExpTabSize;
  static constexpr int kKbRow = D <= 4 ? 112 : 80;
  static constexpr int kKbBuf = 4 * kKbRow;
  static constexpr int kQPad = kKbRow - 64;                 // doubles per k-row
  static constexpr int kQMaxG = kQPad / 8;                  // 16 rows x 2 G doubles
  static constexpr size_t bytes(int nw) {
    return (size_t(kKbOff) + size_t(nw) * kKbBuf) * sizeof(double);
  }
  // double2 number i of a wave's staged Q block -> offset (doubles) in its buffer
  static __device__ __forceinline__ int qoff(int i) {
    return (i / (kQPad / 2)) * kKbRow + 64 + 2 * (i % (kQPad / 2));
  }
};

// One stage of the flattened (GP, chunk, j-block) sequence of a tile, built on
// the host (stage_table) with ABSOLUTE device addresses: no pointer arithmetic per
// stage on the device, one scalar load per entry.  Position t of the staged A chunk
// holds row block bend-1-t of the chunk, k-steps 4 jb .. 4 jb + 3; its source is
// a_src - t * rs_bytes.
struct StageEnt {
  uint64_t a_src;      // device address of position 0's 2 KB
  uint64_t xa;         // device address of the j-block's [16 d rows | 16 alpha] (GpDev::XA);
                       // launches on factor tables (SEP): of its 16 alpha
  uint32_t rs_bytes;   // bytes between consecutive row blocks in Apack
  uint32_t jb;         // j-block: training points 16 jb .. 16 jb + 15
  uint32_t word;       // see SW_*
  uint32_t slot0;      // 2-KB positions of the stages in front of this one (resident mode)
};
enum : uint32_t {
  SW_NACT_MASK = 63u,       // active slots 0 .. nact-1 (1..32)
  SW_CHUNK_END = 1u << 6,   // last j-block of an accumulator chunk: fold
  SW_GP_END = 1u << 7,      // last stage of a GP: row epilogue
  SW_TILE_END = 1u << 8,    // last stage of the tile
  SW_MEAN = 1u << 9,        // stage of the LAST chunk: accumulate alpha . k
  SW_NARROW = 1u << 10,     // slot 0 is the last row block with <= 12 real rows: taken
                            // as SW_NGRP narrow 4-row groups (narrow_groups)
  SW_G_SHIFT = 12,          // GP index (3 bits)
  SW_NGRP_SHIFT = 16,       // narrow groups of slot 0 (1..3)
  SW_FIRST = 1u << 18,      // first stage (j-block 0) of an accumulator chunk
  SW_NSL_SHIFT = 19         // slots of the stage's chunk (1..16; 5 bits)
};

// Timing experiments ("what does the kernel cost without X"; results are wrong
// with any bit set): built only with -DSGP_INSTRUMENT, selected at run time by
// SGP_ABLATE=<mask>: 1 no stage barrier, 2 no LDS-DMA, 4 no covariance
// evaluation, 8 no MFMA, 16 no mean/var/Q stores, 32 no per-GP epilogue at all.
#ifdef SGP_INSTRUMENT
#define SGP_ABL(mask) (p.ablate & (mask))
#else
#define SGP_ABL(mask) false
#endif

struct SweepParams {
  const GpDev* gps;
  int G;
  int mode;
#ifdef SGP_INSTRUMENT
  int ablate;      // timing experiments (scripts/ablate.py), see SGP_ABL
#endif
  SweepPoints pts;
  ConfOut conf;
  FitnessArgs fit;
  const StageEnt* stages;   // [nstages] one tile's stage sequence (all GPs)
  int nstages;
  // Small factors stay in LDS for the whole launch (n <= ~112 rows in all GPs together):
  // every stage's positions are copied ONCE, packed back to back, the [rows | alpha]
  // blocks behind them from byte res_xbase on -- no LDS-DMA, no wait and no barrier per
  // stage, the waves of a workgroup run free.  0: the chunks are streamed (double buffer).
  int resident;
  unsigned res_xbase;
  int single;               // every GP has a one-part kernel (pre-scaled inputs)
  long long ride_delta[SGP_MAX_GPS];   // rider g: bytes from its leader's XA to its own
  int nride[SGP_MAX_GPS];   // riders of GP g: the GPs g + 1 .. g + nride[g] share its
                            // factor AND its covariances (GpDev::share) and have no stages
                            // of their own -- their alpha . k is formed in g's stages
  int slots;                // accumulator slots per wave: 16 or 32 (host only)
  SepLaunch sep;            // tensor grid + factor tables (instances with SEP > 0)
#ifdef SGP_STAMPS
  unsigned long long* stamps;   // [blocks][4 waves][8] cycles per phase (debug build)
#endif
};

// Per-phase cycle stamps of the stage loop (-DSGP_STAMPS, scripts/dev): s_memtime at
// the phase boundaries, summed per wave.  Perturbs the run (every stamp drains the
// LDS / scalar-load counter); for attribution only.
#ifdef SGP_STAMPS
#define SGP_STAMP(i)                                                   \
  do {                                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
    stamp_acc[i] += (unsigned int)(t_ - stamp_prev);                   \
    stamp_prev = t_;                                                   \
  } while (0)
#else
#define SGP_STAMP(i) do {} while (0)
#endif

typedef const __attribute__((address_space(1))) double* gptr_t;
// the stage table is read-only for the whole launch: constant address space, so
// that an entry is ONE scalar load (a plain global pointer becomes a vector load
// with a full memory wait, since the kernel also stores to global memory)
typedef const __attribute__((address_space(4))) StageEnt* stage_ptr_t;
__device__ __forceinline__ StageEnt load_stage(stage_ptr_t t, int i) {
  StageEnt e;      // member-wise: scalar loads
  e.a_src = t[i].a_src;
  e.xa = t[i].xa;
  e.rs_bytes = t[i].rs_bytes;
  e.jb = t[i].jb;
  e.word = t[i].word;
  e.slot0 = t[i].slot0;
  return e;
}
typedef const __attribute__((address_space(4))) GpDev* gpdev_cptr_t;

constexpr int kSweepRide = 2;      // riders per leader (what two workgroups' LDS holds)

// ---- matrix part ------------------------------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 is the fp64 matrix instruction that reaches the chip's
// peak on gfx950 (74-77 TFLOP/s measured vs 49 for v_mfma_f64_16x16x4_f64).  Its
// operand maps (probed):
//   A[blk][i][k] <- lane 16k + 4blk + i     B[blk][k][j] <- lane 16k + 4blk + j
//   D[blk][i][j] -> lane 16i + 4blk + j
// Used with blk = 4-row group: the A operand is then the SAME register the
// 16x16x4 form takes (lane = 16k + row, row = 4blk + i), one instruction is a
// 16-row x 4-column x 4-deep product, and a 16-column slab needs four of them
// (m = 0..3) whose B operands are the covariance register with column quad m
// broadcast to all four quads of each 16-lane row.  (cbsz / abid do NOT
// broadcast for this opcode on gfx950 -- probed, scripts/dev/probe_cbsz.hip.)
// Accumulator component m of a slot holds rows 4((l>>2)&3) + (l>>4), column
// 4m + (l&3).

// (broadcast_quads: sweep_shared.h)

// A operands of one slot: the 4 k-steps of the staged j-block.
__device__ __forceinline__ void load_slot(double (&ops)[4], const double* aT,
                                          int slot) {
#pragma unroll
  for (int q = 0; q < 4; ++q) ops[q] = aT[(slot * kSteps + q) * 64];
}

// One 16-wide j-block against accumulator slots 0..nact-1: a slot is 16 MFMAs on
// four independent accumulators (groups of four separated by one wait state --
// measured: a dense stream without them runs 17.4 cycles per MFMA instead of
// 16.25, profiles/r02/probes.txt); the next slot's A operands are read from LDS
// while they execute (the read of slot nact is harmless: it stays inside the
// stage buffer; the empty asm keeps the compiler from sinking the reads into the
// next slot's block, where their latency would be exposed).  The active slots are
// a prefix, so the guards nest: the first inactive slot leaves the whole sequence
// with one branch.
//
// The MFMA goes through inline asm with the accumulator tied to destination AND
// addend: the builtin lets the register allocator rename the destination, which
// costs v_mov_b64 copies at every join of the guarded sequence.
// (mfma_acc: sweep_shared.h)

//
// Slot 0 of the last chunk is the LAST row block.  With <= 12 real rows it is taken as
// 1..3 NARROW 4-row groups: the A operand of group g carries rows 4 g .. 4 g + 3 in
// all four MFMA blocks -- read from the staged slot (standard layout, lane = 16 k +
// row) with the lane address 16 k + 4 g + (lane & 3): the replication is a broadcast
// read of LDS, no special packing -- so the plain covariance register kv[q], a
// different point quad per block, is the B operand and ONE instruction per k-step
// and group covers all 16 points (accumulator accx[g]: rows l >> 4 of the group,
// point l & 15).  n = 200: 8 real rows in the 13th block, 8 MFMAs per stage instead
// of 16 in the one slot that meets EVERY j-block.  The groups need kv only, not the
// broadcast operands: they are issued in front of the LDS transpose, whose round
// trip passes under them.
constexpr int kMaxNg = 2;     // (three groups, 9..12 rows: measured, no gain over a full slot)
// ONE asm statement for all groups, the groups a GP does not have skipped by a scalar
// branch INSIDE it: variants of the sequence at the source level, or a slot that is
// narrow on one path and full on the other, make the register allocator duplicate
// the 64 accumulators around the joins (400+ bytes of scratch).  Hence also: the
// narrow groups are a region of their own in FRONT of the full slots (position 0 of
// the staged chunk; the full slots then start at position 1) and touch no register
// of theirs.
// Per group four DEPENDENT MFMAs on one accumulator: the addend must not be read
// before the previous result is written (4 wait states for this opcode; nothing pads
// inside asm).  s_nop 1 in front: kv / the register copies the compiler may place
// here are VALU writes, two wait states before an MFMA may read them -- the hazard
// recogniser does not see into the asm.
#define SGP_NARROW_GROUP(ACC, A0, A1, A2, A3)                       \
  "v_mfma_f64_4x4x4_4b_f64 " ACC ", " A0 ", %[k0], " ACC "\n\ts_nop 4\n\t" \
  "v_mfma_f64_4x4x4_4b_f64 " ACC ", " A1 ", %[k1], " ACC "\n\ts_nop 4\n\t" \
  "v_mfma_f64_4x4x4_4b_f64 " ACC ", " A2 ", %[k2], " ACC "\n\ts_nop 4\n\t" \
  "v_mfma_f64_4x4x4_4b_f64 " ACC ", " A3 ", %[k3], " ACC "\n\ts_nop 4\n\t"
__device__ __forceinline__ void narrow_groups(int ngrp, double (&accx)[kMaxNg],
                                              const double (&an)[kMaxNg][4],
                                              const double (&kv)[4]) {
  constexpr int g1 = kMaxNg > 1 ? 1 : 0, g2 = kMaxNg > 2 ? 2 : 0;
  if constexpr (kMaxNg == 1) {
    asm volatile("s_nop 1\n\t" SGP_NARROW_GROUP("%[c0]", "%[a00]", "%[a01]", "%[a02]", "%[a03]")
                 : [c0] "+v"(accx[0])
                 : [a00] "v"(an[0][0]), [a01] "v"(an[0][1]), [a02] "v"(an[0][2]),
                   [a03] "v"(an[0][3]),
                   [k0] "v"(kv[0]), [k1] "v"(kv[1]), [k2] "v"(kv[2]), [k3] "v"(kv[3]));
  } else if constexpr (kMaxNg == 2) {
    // two groups: their chains interleaved (the partner's instruction and s_nop 2 make
    // the 4 wait states of a dependent pair), the second group's half skipped by a branch
    // per k-step when the GP has one group only
#define SGP_NARROW_STEP(Q, A0, A1, K)                                                  \
  "v_mfma_f64_4x4x4_4b_f64 %[c0], " A0 ", " K ", %[c0]\n\t"                             \
  "s_cbranch_scc1 .Lsgp_n1_" #Q "_%=\n\t"                                               \
  "v_mfma_f64_4x4x4_4b_f64 %[c1], " A1 ", " K ", %[c1]\n\t"                             \
  ".Lsgp_n1_" #Q "_%=:\n\ts_nop 3\n\t"
    asm volatile("s_cmp_lt_u32 %[ng], 2\n\ts_nop 0\n\t"
                 SGP_NARROW_STEP(0, "%[a00]", "%[a10]", "%[k0]")
                 SGP_NARROW_STEP(1, "%[a01]", "%[a11]", "%[k1]")
                 SGP_NARROW_STEP(2, "%[a02]", "%[a12]", "%[k2]")
                 SGP_NARROW_STEP(3, "%[a03]", "%[a13]", "%[k3]")
                 "s_nop 2"      // (a VALU instruction may read the results next: 6 wait states)
                 : [c0] "+v"(accx[0]), [c1] "+v"(accx[g1])
                 : [a00] "v"(an[0][0]), [a01] "v"(an[0][1]), [a02] "v"(an[0][2]),
                   [a03] "v"(an[0][3]),
                   [a10] "v"(an[g1][0]), [a11] "v"(an[g1][1]), [a12] "v"(an[g1][2]),
                   [a13] "v"(an[g1][3]),
                   [k0] "v"(kv[0]), [k1] "v"(kv[1]), [k2] "v"(kv[2]), [k3] "v"(kv[3]),
                   [ng] "s"(ngrp)
                 : "scc");
#undef SGP_NARROW_STEP
  } else {
    asm volatile("s_nop 1\n\t" SGP_NARROW_GROUP("%[c0]", "%[a00]", "%[a01]", "%[a02]", "%[a03]")
                 "s_cmp_lt_u32 %[ng], 2\n\ts_cbranch_scc1 .Lsgp_narrow_end_%=\n\t"
                 SGP_NARROW_GROUP("%[c1]", "%[a10]", "%[a11]", "%[a12]", "%[a13]")
                 "s_cmp_lt_u32 %[ng], 3\n\ts_cbranch_scc1 .Lsgp_narrow_end_%=\n\t"
                 SGP_NARROW_GROUP("%[c2]", "%[a20]", "%[a21]", "%[a22]", "%[a23]")
                 ".Lsgp_narrow_end_%=:"
                 : [c0] "+v"(accx[0]), [c1] "+v"(accx[g1]), [c2] "+v"(accx[g2])
                 : [a00] "v"(an[0][0]), [a01] "v"(an[0][1]), [a02] "v"(an[0][2]),
                   [a03] "v"(an[0][3]),
                   [a10] "v"(an[g1][0]), [a11] "v"(an[g1][1]), [a12] "v"(an[g1][2]),
                   [a13] "v"(an[g1][3]),
                   [a20] "v"(an[g2][0]), [a21] "v"(an[g2][1]), [a22] "v"(an[g2][2]),
                   [a23] "v"(an[g2][3]),
                   [k0] "v"(kv[0]), [k1] "v"(kv[1]), [k2] "v"(kv[2]), [k3] "v"(kv[3]),
                   [ng] "s"(ngrp)
                 : "scc");
  }
}
#undef SGP_NARROW_GROUP
// SEP > 0: the rows are a tensor grid with SEP axes and every kernel is a product of
// RBF parts (SepLaunch): a covariance is the product of SEP table entries -- two
// 16-byte loads per axis, lane and stage, issued one stage ahead, instead of ~22 fp64
// instructions per value.  (Instantiated with D = 1: the rows themselves are not read.)
// RES: the factors are small enough to stay in LDS for the whole launch (SweepParams::
// res_xbase) -- instances of their own, so that the streaming instances carry none of it.
template <int D, int NW, int SL, int MODE, bool SINGLE, int R = 0, int SEP = 0, bool RES = false>
__global__ __launch_bounds__(64 * NW, 2) void k_sweep(SweepParams p) {
  constexpr int kTilePts = 16 * NW;
  // (2 kMaxNg doubles live across the evaluation: where the registers are to be had)
  constexpr bool kAnEarly = SEP > 0 && SEP <= 2 && !(SEP == 2 && R > 0);
  // d >= 5 (and products at d = 4): no registers for the raw row of this tile and of
  // the next one -- the row is read (an L2 hit) where a GP's scaled row is formed
  constexpr bool kLeanX = SEP == 0 && (D >= 5 || (D == 4 && !SINGLE));
  // the entry operands of the slot sequence: requested at the top of the stage (8
  // registers across the evaluation) or, where those are not to be had, in front of it
  constexpr bool kEntryEarly = D <= 5;
  // where the LDS-DMA of the next stage is issued: at the top of the stage, or in front
  // of the full slots (measured per variant: with factor tables the table loads want
  // the head of the queue)
  constexpr bool kDmaLate = SEP > 0;
  typedef Lay<SL, D, R> L;
  constexpr int kATile = L::kATile, kBuf = L::kBuf, kXTile = L::kXTile;
  constexpr int kTabOff = L::kTabOff, kKbOff = L::kKbOff, kKbBuf = L::kKbBuf;
  constexpr bool conf = MODE == MODE_CONF;   // compile-time: no dead state
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const double* tab = lds + kTabOff;
  exp_tab_init(lds + kTabOff);   // visible after the first staging barrier

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* kbw = lds + kKbOff + wave * kKbBuf;
  const int ntiles = int((p.pts.N + kTilePts - 1) / kTilePts);
  const stage_ptr_t stages = (stage_ptr_t)(p.stages);
  const gpdev_cptr_t gpc = (gpdev_cptr_t)(p.gps);
  const int nstages = p.nstages;
  const int tstep = int(gridDim.x);
  const uint32_t lds0 = lds_addr_of(lds);
  const uint32_t voff = uint32_t(lane) * 16u;

  int tile = int(blockIdx.x);          // tile of the stage being multiplied
  if (tile >= ntiles) return;
  int left = ((ntiles - tile + tstep - 1) / tstep) * nstages;   // stages still to go
  int tile_p = tile;                   // tile of the stage being prefetched

  // candidate rows of the current tile (and, prefetched, of the next one)
  auto load_x = [&](int t, double (&xo)[D]) {
    int64_t r = int64_t(t) * kTilePts + wave * 16 + (lane & 15);
    r = r < p.pts.N ? r : p.pts.N - 1;
#pragma unroll
    for (int k = 0; k < D; ++k)
      xo[k] = __builtin_nontemporal_load(p.pts.base + r * p.pts.stride_row + k * p.pts.stride_col);
  };
  double x[kLeanX ? 1 : D], xnext[kLeanX ? 1 : D];
  if constexpr (!kLeanX) {
    if (SEP == 0) load_x(tile, x);
#pragma unroll
    for (int k = 0; k < D; ++k) xnext[k] = SEP == 0 ? x[k] : 0.0;
  }
  // SEP: byte offsets of this lane's row (lane & 15) and training points (4 (lane >> 4)
  // .. + 3 of a block of 16) in the tables of the tile being PREFETCHED; axis a's index
  // = (global row / stride_a) % count_a with stride_a = count_0 .. count_{a-1}
  constexpr int kAx = SEP > 0 ? SEP : 1;
  uint32_t soff[kAx];
  auto sep_offsets = [&](int t) {
    int64_t r = int64_t(t) * kTilePts + wave * 16 + (lane & 15);
    r = r < p.pts.N ? r : p.pts.N - 1;
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
      soff[a] = idx * 128u + uint32_t(lane >> 4) * 32u;
    }
  };
  // the factors of the stage about to be multiplied (loaded one stage ahead); the
  // tables of the GP being prefetched stay in scalar registers
  double4_t efn[kAx];
  const char* sep_tab[kAx];
  uint32_t sep_pitch[kAx];
  int sep_g = -1;
  auto sep_fetch = [&](const StageEnt& e) {
    const int g = int(e.word >> SW_G_SHIFT) & 7;
    if (g != sep_g) {
      sep_g = g;
#pragma unroll
      for (int a = 0; a < kAx; ++a) {
        sep_tab[a] = reinterpret_cast<const char*>(uniform_ptr(p.sep.tab[g][a]));
        sep_pitch[a] = __builtin_amdgcn_readfirstlane(p.sep.count[a] * 128u);
      }
    }
    // (explicitly GLOBAL loads: a flat load also counts on lgkmcnt, and every LDS wait
    // of the stage would sit out its latency)
    typedef const __attribute__((address_space(1))) double4_t* gvec_t;
#pragma unroll
    for (int a = 0; a < kAx; ++a) {
      const char* src = sep_tab[a] + e.jb * sep_pitch[a];      // (<= 256 MB per GP: sep_launch)
      efn[a] = *(gvec_t)(reinterpret_cast<const double4_t*>(src + soff[a]));
    }
  };

  // What a stage needs besides its table entry comes by LDS-DMA from absolute
  // addresses of the entry: the A chunk (position t = row block bend-1-t, 2 KB each;
  // wave w copies positions w, w + NW, ..) and the block [16 d rows | 16 alpha] of the
  // j-block (one instruction of one wave; SEP: the 16 alpha only), + the alpha
  // blocks of the riders behind it.
  auto prefetch_to = [&](const StageEnt& e, uint32_t a_dst, uint32_t x_dst) {
    const int nact = int(e.word & SW_NACT_MASK);
    const uint64_t src0 = e.a_src - uint64_t(uint32_t(wave)) * e.rs_bytes;
    const uint64_t step = uint64_t(e.rs_bytes) * NW;
#pragma unroll
    for (int i = 0; i < SL / NW; ++i)
      if (nact > wave + NW * i)
        dma_2k(src0 - uint64_t(i) * step, a_dst + uint32_t(wave + NW * i) * 2048u, voff);
    if (wave == NW - 1) {
      if (SEP > 0) {
        if (lane < 8) dma_1k(e.xa, x_dst + kXTile * 8u, voff);
      } else {
        constexpr int kLanes = 8 * D + 8;              // 16 bytes each
        if (lane < (kLanes < 64 ? kLanes : 64)) dma_1k(e.xa, x_dst, voff);
        if (kLanes > 64) {
          if (lane < kLanes - 64) dma_1k(e.xa + 1024, x_dst + 1024, voff);
        }
      }
      if (R > 0) {
        const int g = int(e.word >> SW_G_SHIFT) & 7;
        const int nr = p.nride[g];
        const uint64_t al = SEP > 0 ? e.xa : e.xa + 128u * D;
        for (int f = 0; f < nr; ++f)       // (wave-uniform)
          if (lane < 8)
            dma_1k(al + uint64_t(p.ride_delta[g + 1 + f]),
                   x_dst + uint32_t(kXTile + kJC * (1 + f)) * 8u, voff);
      }
    }
  };
  auto prefetch = [&](const StageEnt& e, int buf) {
    const uint32_t a_dst = lds0 + uint32_t(buf) * (kBuf * 8u);
    prefetch_to(e, a_dst, a_dst + kATile * 8u);
  };
  // resident mode: where stage si of the sequence lives (bytes from lds0)
  constexpr uint32_t kXBlk = uint32_t(kXTile + kJC * (1 + R)) * 8u;
  constexpr bool resident = RES;

  // Stage cursors: the stage being multiplied (its entry word in wcur), the stage
  // being prefetched (entry e1, one ahead) and the stage whose entry is being loaded
  // (two ahead: a scalar load issued a whole stage before its use).
  KernFast<D> kf;
  double kdiag;
  int nr_cur = 0;
  auto load_gp = [&](int g) {
    if (SEP == 0) kf.load_const(&p.gps[g].kern);
    kdiag = gpc[g].kern.kdiag;
    if (R > 0) nr_cur = p.nride[g];
  };
  StageEnt e1 = load_stage(stages, 0);
  load_gp(int(e1.word >> SW_G_SHIFT) & 7);
  if (resident) {
#pragma unroll 1
    for (int si = 0; si < nstages; ++si) {
      const StageEnt es = load_stage(stages, si);
      prefetch_to(es, lds0 + es.slot0 * 2048u, lds0 + p.res_xbase + uint32_t(si) * kXBlk);
    }
  } else {
    prefetch(e1, 0);
  }
  if (SEP > 0) {
    sep_offsets(tile);
    sep_fetch(e1);
  }
  uint32_t wcur = e1.word;
  uint32_t s0cur = e1.slot0;           // (resident mode: position 0 / index of the stage
  int sicur = 0;                       //  being multiplied)
  int si1 = nstages > 1 ? 1 : 0;       // index of the stage being prefetched
  if (si1 == 0) tile_p += tstep;
  if (left > 1) e1 = load_stage(stages, si1);
  wait_dma();
  __syncthreads();

  // per-GP state
  double xs[D];
  double ssq_run = 0.0, mean = 0.0;   // folded squares of the GP's finished chunks (per lane)
  double mean_r[R > 0 ? R : 1];      // alpha . k of the riders of the GP being swept
#pragma unroll
  for (int f = 0; f < (R > 0 ? R : 1); ++f) mean_r[f] = 0.0;
  double accx[kMaxNg];              // narrow groups of slot 0 (narrow_groups)
#pragma unroll
  for (int g = 0; g < kMaxNg; ++g) accx[g] = 0.0;
  // per-tile state of the row epilogue
  bool safe = true;
  double l0 = 0.0;
  double lmax = -INFINITY;   // max l0 over the safe rows this wave has seen
  bool gp_start = true;

#ifdef SGP_STAMPS
  unsigned int stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  int par = 0;
#pragma unroll 1
  while (true) {
    // the stage's positions and its [rows | alpha] block: the buffer of this parity, or
    // (resident mode) where the stage was copied to at the start
    const uint32_t a_off = resident ? s0cur * 2048u : uint32_t(par) * (kBuf * 8u);
    const uint32_t x_off = resident ? p.res_xbase + uint32_t(sicur) * kXBlk
                                    : a_off + kATile * 8u;
    double* cbuf = lds + (a_off >> 3);

    // SEP: the covariances of THIS stage from the factors fetched a stage ago (their
    // registers take the next stage's right below)
    double kv[4];
    if (SEP > 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        kv[q] = efn[0][q];
#pragma unroll
        for (int a = 1; a < kAx; ++a) kv[q] *= efn[a][q];
      }
      // (... BEFORE the next stage's factors are requested: loads complete in order, so a
      // wait for these values placed behind the new requests would sit out THEIR latency)
      asm volatile("" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]));
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- prefetch: the stage after this one (possibly of the next tile)
    const bool more = left > 1;
    const uint32_t wnext = e1.word;
    const uint32_t s0next = e1.slot0;
    if (more) {
      if (!kDmaLate && !resident && !SGP_ABL(2)) prefetch(e1, par ^ 1);
      const bool next_tile = si1 == 0;
      if (SEP == 0) {
        if constexpr (!kLeanX) {
          if (next_tile) load_x(tile_p, xnext);
        }
      } else {
        if (next_tile) sep_offsets(tile_p);
        sep_fetch(e1);
      }
    }
    // the entry after that: loaded now, first used at the top of the next stage
    int si2 = si1 + 1;
    if (si2 == nstages) {
      si2 = 0;
      tile_p += tstep;
    }
    StageEnt e2 = e1;
    if (left > 2) e2 = load_stage(stages, si2);

    SGP_STAMP(0);   // prefetch issue, table entry
    // ---- this stage: 16 training points against the active row blocks
    if (SEP == 0 && gp_start) {
      if constexpr (kLeanX) {
        double xr[D];
        load_x(tile, xr);
        kf.template prep_t<SINGLE>(xr, xs);
      } else {
        kf.template prep_t<SINGLE>(x, xs);
      }
      gp_start = false;
    }
    const double* xT = lds + (x_off >> 3);
    const double* alT = xT + kXTile;
    // A operands of the narrow groups (position 0 of the staged chunk), read FIRST in
    // the stage: they arrive under the evaluation.  (All kMaxNg groups whether the GP
    // has them or not: the reads stay inside the slot.)
    const int ngrp = (wcur & SW_NARROW) ? int(wcur >> SW_NGRP_SHIFT) & 3 : 0;
    // the full slots of this stage (sweep_slots.h): the operands of the first one are
    // requested now
    const int nfull = int(wcur & SW_NACT_MASK) - (ngrp > 0 ? 1 : 0);
    const uint32_t abase = lds0 + a_off + uint32_t(lane) * 8u +
                           (ngrp > 0 ? kSteps * 512u : 0u);
    SgpEntryOps entry;
    if (kEntryEarly && nfull > 0 && !SGP_ABL(8))
      sgp_slots_prefetch(nfull, cbuf + lane + (ngrp > 0 ? kSteps * 64 : 0), entry);
    double an[kMaxNg][4];
    auto load_an = [&]() {
      const double* aN = cbuf + (lane & 0x33);
#pragma unroll
      for (int g = 0; g < kMaxNg; ++g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) an[g][q] = aN[q * 64 + 4 * g];
      }
    };
    if (kAnEarly && ngrp > 0 && !SGP_ABL(8)) load_an();
    if (SEP > 0) {
      // (done at the top of the stage)
    } else if (!SGP_ABL(4)) {
      kf.template many4_t<SINGLE>(xs, xT + (lane >> 4) * D, 4 * D, tab, kv);
    } else {
      kv[0] = xs[0]; kv[1] = xs[0] + 1.0; kv[2] = xs[0] + 2.0; kv[3] = xs[0] + 3.0;
    }
    if (wcur & SW_MEAN) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        mean = fma(alT[q * 4 + (lane >> 4)], kv[q], mean);
      if (R > 0) {
#pragma unroll
        for (int f = 0; f < R; ++f) {
          if (f < nr_cur) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              mean_r[f] = fma(alT[kJC * (1 + f) + q * 4 + (lane >> 4)], kv[q], mean_r[f]);
          }
        }
      }
    }
    // the narrow groups of the last row block (position 0 of the staged chunk): they
    // need kv only -- issued in front of the LDS transpose, whose round trip passes
    // under them
    SGP_STAMP(1);   // covariance evaluation (+ alpha . k)
    if (ngrp > 0 && !SGP_ABL(8)) {
      if (!kAnEarly) load_an();
      narrow_groups(ngrp, accx, an, kv);
    }
    SGP_STAMP(2);   // narrow groups
    double kb[4][4];
    if (!SGP_ABL(8)) broadcast_quads<L::kKbRow>(kv, kbw, lane, kb);
#ifdef SGP_STAMPS
    asm volatile("" : "+v"(kb[0][0]), "+v"(kb[1][0]), "+v"(kb[2][0]), "+v"(kb[3][0]));
    asm volatile("" : "+v"(kb[0][3]), "+v"(kb[1][3]), "+v"(kb[2][3]), "+v"(kb[3][3]));
#endif
    SGP_STAMP(3);   // LDS transpose round trip
    if (kDmaLate && more && !resident && !SGP_ABL(2)) prefetch(e1, par ^ 1);
    if (!SGP_ABL(8)) {
      // the full slots: sweep_slots.h (hand-written, accumulators in a0..a127)
      if (!kEntryEarly && nfull > 0)
        sgp_slots_prefetch(nfull, cbuf + lane + (ngrp > 0 ? kSteps * 64 : 0), entry);
      if (nfull > 0) sgp_slots(nfull, int(wcur & SW_FIRST), abase, kb, entry);
    }

    SGP_STAMP(4);   // full slots
    if (wcur & SW_CHUNK_END) {
      // squares of the chunk's accumulators, folded at once to this lane's share of
      // |L^-1 k|^2 for its point (lane l: point l & 15): ONE register survives the chunk
      // (the chunk's slots: what its LAST stage -- this one -- has active is the
      // diagonal block only; the table carries the chunk's slot count)
      double sq[4] = {0.0, 0.0, 0.0, 0.0};
      const int nsl = (int(wcur >> SW_NSL_SHIFT) & 31) - (ngrp > 0 ? 1 : 0);
      if (nsl > 0 && !SGP_ABL(8)) sgp_fold_slots(nsl, sq);
      // lane 16 i' + 4 blk + j' of sq[m]: the squares for point 4 m + j' of row group
      // blk on the DIAGONAL lanes i' = j' only (cross terms elsewhere)
      if ((lane >> 4) != (lane & 3)) sq[0] = sq[1] = sq[2] = sq[3] = 0.0;
      // sq[m]: partial sums for column 4m + (lane & 3) over this lane's rows.
      // Transposing fold over the lanes that share (lane & 3): after the xor-4
      // and xor-8 exchanges each lane holds the quad of its OWN column
      // (lane & 15) = 4 ((lane >> 2) & 3) + (lane & 3).
      const bool a0 = (lane & 4) != 0, a1 = (lane & 8) != 0;
      // (exchanges by DPP row rotations, the lane-group sums below by permlane swaps --
      // sweep_shared.h: same values in the same order as the ds_bpermute forms, without
      // their LDS round trips on the tile's critical path)
      const double v0 = (a0 ? sq[1] : sq[0]) + take_xor4(a0 ? sq[0] : sq[1], a0);
      const double v1 = (a0 ? sq[3] : sq[2]) + take_xor4(a0 ? sq[2] : sq[3], a0);
      double t = (a1 ? v1 : v0) + take_xor8(a1 ? v0 : v1);
      // (the narrow groups: rows l >> 4 of the group, point l & 15 already)
#pragma unroll
      for (int g = 0; g < kMaxNg; ++g) {
        t = fma(accx[g], accx[g], t);
        accx[g] = 0.0;
      }
      ssq_run += t;
    }

    if ((wcur & SW_GP_END) && !SGP_ABL(32)) {
      // ... then the 4 k-rows
      const double sumsq = sum_lane_groups_valu(ssq_run);
      const double mu = sum_lane_groups_valu(mean);
      ssq_run = 0.0;
      mean = 0.0;

      const int g = int(wcur >> SW_G_SHIFT) & 7;
      const int64_t row = int64_t(tile) * kTilePts + wave * 16 + (lane & 15);
      const bool writer = (row < p.pts.N) && (lane < 16);
      // one GP's posterior at the wave's rows -> mean / var, its interval, the safe
      // set (update_confidence_intervals + compute_safe_set, gp_opt.py:453-481)
      auto emit = [&](int gi, double mu_i, double kdiag_i) {
        const double var = fmax(kdiag_i - sumsq, 1e-15);  // GPy clip
        const double sd = sqrt(var);
        const double lo = mu_i - p.conf.beta * sd;
        const double up = mu_i + p.conf.beta * sd;
        if (gi == 0) l0 = lo;
        safe = safe && (lo > p.conf.fmin[gi]);
        if (writer && !SGP_ABL(16)) {
          // streaming rows in / results out: non-temporal, so that they do not push
          // the L^-1 chunks every workgroup re-reads out of the 4 MB L2 of the XCD
          __builtin_nontemporal_store(mu_i, p.conf.mean + int64_t(gi) * p.pts.N + row);
          __builtin_nontemporal_store(var, p.conf.var + int64_t(gi) * p.pts.N + row);
        }
        // Q row = [l0, u0, l1, u1, ...] (gp_opt.py:375): the intervals of the G
        // passes are collected in the padding of the wave's broadcast buffer
        // and leave as ONE contiguous block per wave at the end of the tile --
        // full 128-byte lines from all 64 lanes instead of 16-byte pieces of a
        // row from 16 lanes per GP
        if (p.conf.Q && lane < 16 && !SGP_ABL(16)) {
          if (p.G <= L::kQMaxG)
            *reinterpret_cast<double2_t*>(kbw + L::qoff(lane * p.G + gi)) =
                double2_t{lo, up};
          else if (writer)
            *reinterpret_cast<double2_t*>(p.conf.Q + (row * p.G + gi) * 2) =
                double2_t{lo, up};
        }
      };
      if (conf) {
        emit(g, mu, kdiag);
        if (R > 0) {
          // riders: the leader's |L^-1 k|^2, their own alpha . k and prior variance
#pragma unroll
          for (int f = 0; f < R; ++f) {
            if (f < nr_cur) {
              const double mu_f = sum_lane_groups_valu(mean_r[f]);
              mean_r[f] = 0.0;
              emit(g + 1 + f, mu_f, gpc[g + 1 + f].kern.kdiag);
            }
          }
        }
      }

      if (wcur & SW_TILE_END) {
        if (conf) {
          if (p.conf.Q && p.G <= L::kQMaxG && !SGP_ABL(16)) {
            const int64_t row0 = int64_t(tile) * kTilePts + wave * 16;
            const int64_t rows_left = p.pts.N - row0;
            const int nq = (rows_left >= 16 ? 16 : (rows_left > 0 ? int(rows_left) : 0)) * p.G;
            __builtin_amdgcn_wave_barrier();
            double2_t* dst = reinterpret_cast<double2_t*>(p.conf.Q) + row0 * p.G;
            for (int i = lane; i < nq; i += 64)
              __builtin_nontemporal_store(
                  *reinterpret_cast<const double2_t*>(kbw + L::qoff(i)), dst + i);
            __builtin_amdgcn_wave_barrier();
          }
          if (p.conf.S) {
            if (writer) p.conf.S[row] = safe ? 1 : 0;
            // running maximum of l0 over the safe rows (folded over the wave and
            // written once, when the wave has walked all its tiles)
            lmax = fmax(lmax, (writer && safe) ? l0 : -INFINITY);
          }
        }
        safe = true;
        l0 = 0.0;
        if constexpr (!kLeanX) {
#pragma unroll
          for (int k = 0; k < D; ++k) x[k] = xnext[k];
        }
        tile += tstep;
      }
      if (more) {   // hyper-parameters of the next GP
        const int g_n = int(wnext >> SW_G_SHIFT) & 7;
        if (g_n != g || R > 0) load_gp(g_n);
      }
      gp_start = true;
    }

    SGP_STAMP(5);   // chunk fold, row epilogue
    if (!more) break;
    if (!resident) {
      wait_dma();       // (asm copies: the compiler does not count them)
      SGP_STAMP(6);   // wait for this wave's LDS-DMA
      if (!SGP_ABL(1)) __syncthreads();
      SGP_STAMP(7);   // barrier
    }
    par ^= 1;
    wcur = wnext;
    s0cur = s0next;
    sicur = si1;
    e1 = e2;
    si1 = si2;
    --left;
  }
#ifdef SGP_STAMPS
  if (lane == 0) {
    unsigned long long* o = p.stamps + (size_t(blockIdx.x) * NW + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = stamp_acc[i];
  }
#endif
  if (conf && p.conf.S) {
    lmax = wave_max(lmax);
    if (lane == 0) p.conf.partial[int(blockIdx.x) * NW + wave] = lmax;
  }
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
        // (the smaller of the two bounds on the posterior covariance: k_expander_many)
        const double tn2c = ea.tn2[g * 16 + cand];
        const double cmax =
            fmin(fabs(kxc[r]) + sqrt(qx * tn2c),
                 sqrt((var + 1e-12 * kdiag) * (fmax(kdiag - tn2c, 0.0) + 1e-12 * kdiag))) *
            (1.0 + 1e-9);
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

// Many candidates in ONE pass over the unsafe rows (sgp_grid_expander_pass; the expander
// loop of gp_opt.py:557-612 where it has to visit hundreds or thousands of candidates): the
// candidates come in groups of 16 -- the operand blocks of k_expander, laid out
// [group][GP][...] -- and a wave keeps its 16 rows, their posterior and the GP's kernel while
// it walks the groups: pre-filter (one covariance per row and candidate), and only where a
// row could be lifted above fmin the n-term contraction on the matrix cores.  Same tests,
// same arithmetic per (row, candidate) as k_expander.
//
// Three launches.  The rows that pass the pre-filter for ANY candidate are few and they sit
// together (the band next to the safe set whose lower bound is just below fmin: 500 of 750000
// unsafe rows in the converged state of bench.py, in 290 of 47000 waves) -- and such a row
// passes it for nearly EVERY candidate, so a scan that tested and contracted in place left the
// pair tests and the matrix products of a whole pass to a few hundred waves, one block after
// the other (all but 0.1 ms of a pass).  So
//   MODE 0, the scan of the grid, only looks for the waves with a block that passes the BLOCK
//           test (one group per lane, out at the first one) -> a list of waves per GP
//           (ea.wlist, ea.wcount);
//   MODE 2  takes the listed waves and the groups kManyTestChunk at a time -- an item per wave
//           of the launch, as many as the chip has -- and finds the ROWS that pass the block
//           test of the row alone for some group (kernels of several parts, which have no
//           block test: the pair test for some candidate) -> a list of rows per GP (ea.list,
//           ea.count; a row once: ea.wmask);
//   MODE 1  takes the listed rows 16 at a time and the groups kManyChunk at a time -- an item
//           per wave; with few items 4 at a time, an item per workgroup, its four waves a
//           quarter of the training points each -- and runs the same tests and the contraction
//           of what is left of them.
constexpr int kManyKbRow = 80;     // doubles between the k-rows of a wave's transpose buffer
constexpr int kManyChunk = 8;      // groups of an item of MODE 1
constexpr int kManyTestChunk = 32; //                   of MODE 2
constexpr int kManyListBlocks = 1024;   // workgroups of MODE 1 / 2 (4 items at a time each)
constexpr int kManyCoopItems = 1024;    // MODE 1: up to that many items of 4 groups go one per workgroup
#ifdef EXPM_STATS
__device__ unsigned long long g_expm_stats[32];
extern "C" void sgp_debug_expm_stats(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_expm_stats), sizeof(g_expm_stats));
  if (reset) { unsigned long long z[32] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_expm_stats), z, sizeof(z)); }
}
#define EXPM_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); \
    if (lane == 0) atomicAdd(&g_expm_stats[i], v_); } while (0)
#else
#define EXPM_STAT(i, v) do {} while (0)
#endif

// A wave's 16 rows (lane & 15) of one GP, what the tests need of them, and the tests.  SINGLE:
// a kernel of one part (resolved per GP at the top of the launch: the code of a pass is tens
// of KB and a wave at a time per SIMD runs it -- the paths it does not take stay out of the
// instruction cache, and the loops below are rolled where four independent chains are enough).
template <int D, bool SINGLE>
struct ManyRows {
  KernFast<D> kf;
  double x[D], xlo[D], xhi[D];
  double mu, var, sqx, svx, mu_hi, var_lo, sqx_hi, svx_hi, beta2, fmin;
  bool unsafe;
  int g, G, m_total;

  __device__ __forceinline__ ManyRows(const GpDev& gp, int g_, int G_, const ExpanderArgs& ea,
                                      const double (&x_)[D], double mu_, double var_, bool unsafe_)
      : kf(gp.kern), mu(mu_), var(var_), unsafe(unsafe_), g(g_), G(G_), m_total(ea.m) {
    const double kdiag = gp.kern.kdiag;
    // bounding box of the wave's unsafe rows (consecutive rows of the grid: a short segment)
#pragma unroll
    for (int k = 0; k < D; ++k) {
      x[k] = x_[k];
      xlo[k] = -wave_max(unsafe ? -x[k] : -INFINITY);
      xhi[k] = wave_max(unsafe ? x[k] : -INFINITY);
    }
    // ... and the extremes of their posterior: with the group's extremes (k_pass_agg) an upper
    // bound of what ANY pair of the block can reach
    mu_hi = wave_max(unsafe ? mu : -INFINITY);
    const double var_hi = wave_max(unsafe ? var : -INFINITY);
    var_lo = -wave_max(unsafe ? -var : -INFINITY);
    sqx_hi = sqrt(fmax(kdiag - var_lo, 0.0));
    svx_hi = sqrt(var_hi + 1e-12 * kdiag);
    // |L^-1 k_x| and the posterior standard deviation of the row (+ room for the rounding
    // of a variance that is a difference of O(k(x,x)) terms)
    sqx = sqrt(fmax(kdiag - var, 0.0));
    svx = sqrt(var + 1e-12 * kdiag);
    beta2 = ea.beta * ea.beta;
    fmin = ea.fmin[g];
  }
  static __device__ __forceinline__ double fmin2(double a, double b) { return ::fmin(a, b); }

  // mu + |delta| c - beta sqrt(var - c^2 / s2) + slack >= fmin for the largest |c| allowed
  __device__ __forceinline__ bool reach(double cmax, double mu_, double var_, double dl, double is2) const {
    const double mu2 = fma(dl, cmax, mu_);
    const double var2 = fmax(var_ - cmax * cmax * is2, 1e-15);
    const double room = mu2 + 1e-9 * (fabs(mu2) + 1.0) - fmin;
    return room >= 0.0 && room * room * (1.0 + 1e-9) >= beta2 * var2;
  }

  // The BLOCK test, one group per lane: the covariance of the closest points of the wave's
  // box and the group's box with the extremes of both sides bounds what any of the block's 256
  // pairs can reach -- most blocks are far apart and end here, for a fraction of an
  // instruction per pair (single-part kernels).
  __device__ __forceinline__ bool block(const ExpanderArgs& ea, int zz, const double* tab) const {
    return block_at(ea.box, ea.agg, zz, tab);
  }
  // (boxes / extremes: of the groups, or of the supergroups of 8 groups)
  __device__ __forceinline__ bool block_at(const double* boxes, const double* aggs, int zz,
                                           const double* tab) const {
    if (!SINGLE) return true;
    const double* bx = boxes + int64_t(zz) * 2 * D;
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double gap = fmax(fmax(bx[k] - xhi[k], xlo[k] - bx[D + k]), 0.0) * kf.sc[k];
      r2 = fma(gap, gap, r2);
    }
    const double kmax = kf.of_r2s(r2, tab);
    const double* ag = aggs + (int64_t(zz) * G + g) * 4;
    const double cmax = fmin2(fma(sqx_hi, ag[2], kmax), svx_hi * ag[3]) * (1.0 + 1e-9);
    return reach(cmax, mu_hi, var_lo, ag[0], ag[1]);
  }

  // The same bound for the lane's OWN row against group zz (its box, its extremes): what the
  // pair tests of the row with the group's 16 candidates -- neighbours along a grid line -- can
  // reach, for one covariance evaluation instead of 16.
  __device__ __forceinline__ bool row_block(const ExpanderArgs& ea, int zz, const double* tab) const {
    const double* bx = ea.box + int64_t(zz) * 2 * D;
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double gap = fmax(fmax(bx[k] - x[k], x[k] - bx[D + k]), 0.0) * kf.sc[k];
      r2 = fma(gap, gap, r2);
    }
    const double kmax = kf.of_r2s(r2, tab);
    const double* ag = ea.agg + (int64_t(zz) * G + g) * 4;
    const double cmax = fmin2(fma(sqx, ag[2], kmax), svx * ag[3]) * (1.0 + 1e-9);
    return unsafe && reach(cmax, mu, var, ag[0], ag[1]);
  }

  // The PAIR tests of four groups, the lane's row against the candidates (lane >> 4) + 4 r of
  // each: an upper bound of the updated lower bound from ONE covariance evaluation per (row,
  // candidate).  c(x) is the POSTERIOR covariance of the two:
  //   |c| <= |k(x, x_c)| + |L^-1 k_x| |L^-1 k_c|        (k_expander's bound: small far
  //                                                      from the data and the candidate)
  //   |c| <= sd(x) sd(x_c)                               (Cauchy-Schwarz on the posterior:
  //                                                      small NEXT to the data)
  // -- with the second one the rows an observation has pinned below fmin drop out for every
  // candidate.  Squares instead of square roots (|L^-1 k_c|, sd(x_c): k_pass_aux).  The four
  // groups' loads and evaluations overlap (no branch around them); p[b]: a pair of group b passes.
  __device__ __forceinline__ void pair4(const ExpanderArgs& ea, const int (&z)[4], int lane,
                                        const double* tab, bool (&p)[4]) const {
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) p[b4] = false;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      const int cand = (lane >> 4) + 4 * r;
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) {
        const int m = min(16, m_total - 16 * z[b4]);
        const int cc = min(cand, m - 1);
        const int64_t zo = (int64_t(z[b4]) * G + g) * 16 + cc;
        const double kxc = kf.template raw_t<SINGLE>(x, ea.xc + (int64_t(z[b4]) * 16 + cc) * D, tab);
        const double cmax = fmin2(fma(sqx, ea.stn[zo], fabs(kxc)), svx * ea.svc[zo]) * (1.0 + 1e-9);
        p[b4] = p[b4] || (cand < m && unsafe && reach(cmax, mu, var, fabs(ea.delta[zo]), ea.inv_s2[zo]));
      }
    }
  }
};

// The groups of `mask` (bit j: group zlo + j) whose pair test passes for some pair, four groups
// at a time; rows: the 16-bit mask of the wave's rows with such a pair.
template <int D, bool SINGLE>
__device__ __forceinline__ unsigned long long many_pairs(const ManyRows<D, SINGLE>& rw,
                                                         const ExpanderArgs& ea, int zlo,
                                                         unsigned long long mask, const double* tab,
                                                         int lane, unsigned& rows) {
  unsigned long long todo = 0ull;
#pragma unroll 1
  while (mask != 0ull) {
    int jb[4], zb[4];
    jb[0] = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask));
    mask &= mask - 1ull;
#pragma unroll
    for (int b4 = 1; b4 < 4; ++b4) {
      jb[b4] = mask != 0ull ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask)) : jb[0];
      mask &= mask - (mask != 0ull ? 1ull : 0ull);
    }
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) zb[b4] = zlo + jb[b4];
    bool p[4];
    rw.pair4(ea, zb, lane, tab, p);
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) {
      const unsigned long long pb = __ballot(p[b4]);
      if (pb != 0ull) {                                        // wave-uniform
        todo |= 1ull << jb[b4];
        rows |= unsigned((pb | (pb >> 16) | (pb >> 32) | (pb >> 48)) & 0xffffull);
      }
    }
  }
  return todo;
}

// MODE 1: the wave's 16 listed rows against the groups [zlo, zhi), tests and contraction.
template <int D, bool SINGLE>
__device__ __forceinline__ void many_rows(const GpDev& gp, const ManyRows<D, SINGLE>& rw,
                                          const ExpanderArgs& ea, int zlo, int zhi, const double* tab,
                                          double* kbw, double* rowbuf, double* red, int lane,
                                          int wave, bool coop) {
  const KernFast<D>& kf = rw.kf;
  const int g = rw.g, G = rw.G;
  const int m_total = ea.m;
  double xs[D];
  kf.template prep_t<SINGLE>(rw.x, xs);
  const int nsteps = gp.n_pad >> 2;
  gptr_t Xj = (gptr_t)gp.Xs + (lane >> 4) * D;
  // the wave's rows for the final test of a block: [row][x | mean | var | unsafe]
  __builtin_amdgcn_wave_barrier();
  if (lane < 16) {
    double* rb = rowbuf + lane * (D + 3);
#pragma unroll
    for (int k = 0; k < D; ++k) rb[k] = rw.x[k];
    rb[D] = rw.mu;
    rb[D + 1] = rw.var;
    rb[D + 2] = rw.unsafe ? 1.0 : 0.0;
  }
  __builtin_amdgcn_wave_barrier();
  // (16 rows x 16 candidates) blocks that passed the tests go kQ at a time: ONE evaluation of
  // the rows' covariances with the training points feeds the matrix products of all kQ blocks
  // (the evaluation, ~25 fp64 instructions per value, costs three times the four matrix
  // instructions it feeds)
  constexpr int kQ = 4;
  unsigned zpack = 0u;            // the queue: group zlo + byte q of zpack (an item has <= 64 groups)
  int nq = 0;
  auto zq = [&](int q) { return zlo + int((zpack >> (8 * q)) & 0xffu); };
  // The matrix products run on v_mfma_f64_4x4x4_4b_f64 -- the fp64 instruction that reaches
  // the chip's peak; the 16 x 16 x 4 form stops at two thirds of it --: per k-step four
  // instructions whose B operands are the rows' covariances with row quad m broadcast to
  // all four quads (the LDS transpose of the sweeps, broadcast_quads), A = the packed
  // Ky^-1 k_c as it is (lane 16 k + candidate).  D[blk][i][j] -> lane 16 i + 4 blk + j: a lane
  // ends with ONE candidate, 4 blk + i, at the four rows 4 m + j -- whose x, mean and
  // variance it reads from the wave's row buffer.
  // `coop` (few items in the launch: the pass waits for the longest of them): the four waves of
  // the workgroup hold the SAME rows and queue (same tests on the same data), each takes a
  // quarter of the training points, the partial products meet in LDS and wave j finishes block
  // j -- an item is a chain of n / 64 evaluations, not n / 16.  Otherwise a wave has its own
  // item and all training points.
  auto flush = [&]() {
    double acc[kQ][4];
#pragma unroll
    for (int j = 0; j < kQ; ++j)
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) acc[j][m4] = 0.0;
    const int nit = nsteps >> 2;                       // (n_pad is a multiple of 16)
    const int sbeg = coop ? 4 * ((nit * wave) >> 2) : 0;
    const int send = coop ? 4 * ((nit * (wave + 1)) >> 2) : nsteps;
    double xr[4][D], xn[4][D];
    auto fetch = [&](int s0, double (&xo)[4][D]) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < D; ++k) xo[q][k] = Xj[(s0 + q) * 4 * D + k];
    };
    if (sbeg < send) fetch(sbeg, xr);
#pragma unroll 1
    for (int s0 = sbeg; s0 < send; s0 += 4) {
      if (s0 + 4 < send) fetch(s0 + 4, xn);
      // the A operands of all queued blocks are requested in front of the evaluation: their
      // latency (L2) passes under it and under the other wave of the SIMD (no second set of
      // them a step ahead: its 32 registers are the second wave)
      double a[kQ][4];
#pragma unroll
      for (int j = 0; j < kQ; ++j) {
        if (j < nq) {                     // (wave-uniform)
          gptr_t W = (gptr_t)ea.Wpack + (int64_t(zq(j)) * G + g) * ea.wstride + lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) a[j][q] = W[(s0 + q) * 64];
        }
      }
      double kv[4];
      kf.template manyn_t<4, SINGLE>(xs, &xr[0][0], D, tab, kv);
      double kb[4][4];
      broadcast_quads<kManyKbRow>(kv, kbw, lane, kb);
#pragma unroll
      for (int j = 0; j < kQ; ++j) {
        if (j < nq) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4)
              acc[j][m4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][q], kb[m4][q], acc[j][m4], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < D; ++k) xr[q][k] = xn[q][k];
    }
#pragma unroll
    for (int j = 0; j < kQ; ++j)
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) red[((wave * kQ + j) * 4 + m4) * 64 + lane] = acc[j][m4];
    if (coop) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    // the final test of the blocks (coop: block j belongs to wave j), the products from LDS
    const int cand = 4 * ((lane >> 2) & 3) + (lane >> 4);
    const int jlo = coop ? wave : 0, jhi = coop ? min(wave + 1, nq) : nq;
#pragma unroll 1
    for (int j = jlo; j < jhi; ++j) {
      const int z = zq(j);
      const int m = min(16, m_total - 16 * z);
      const int64_t zo = (int64_t(z) * G + g) * 16;
      const double* xc = ea.xc + int64_t(z) * 16 * D;
      bool hit = false;
      if (cand < m) {
        const double dl = ea.delta[zo + cand], is2 = ea.inv_s2[zo + cand];
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
          const double* rb = rowbuf + (4 * m4 + (lane & 3)) * (D + 3);
          if (rb[D + 2] != 0.0) {                       // an unsafe row of the grid
            double dot = red[(((coop ? 0 : wave) * kQ + j) * 4 + m4) * 64 + lane];
            if (coop) {                                 // (in the order of the training points)
#pragma unroll
              for (int w = 1; w < 4; ++w) dot += red[((w * kQ + j) * 4 + m4) * 64 + lane];
            }
            const double cx = kf.template raw_t<SINGLE>(rb, xc + cand * D, tab) - dot;
            const double mu2 = rb[D] + cx * dl;
            const double var2 = fmax(rb[D + 1] - cx * cx * is2, 1e-15);
            hit = hit || (mu2 - ea.beta * sqrt(var2) >= ea.fmin[g]);
          }
        }
      }
      if (hit) atomicOr(&ea.flags[(int64_t(z) * 16 + cand) * G + g], 1);
    }
    if (coop) __syncthreads();
    nq = 0;
  };
  // The item's groups (at most 64): the block test one group per lane, a finer test for the
  // groups that are left, then the blocks that pass it, four per evaluation.
  static_assert(kManyChunk <= 64, "one block test per item");
  EXPM_STAT(8, 1);
  const unsigned long long mask = __ballot(zlo + lane < zhi && rw.block(ea, zlo + lane, tab));
  unsigned long long todo = 0ull;
  if (SINGLE && gp.n_pad <= 256) {
    // ... which for a kernel of one part and a small factor is the block test of the ROW
    // against the group, four groups at a time (lane = 16 group + row): one covariance
    // evaluation per (row, group) instead of 16 -- the candidates of a group are neighbours
    // along a grid line, their box and extremes bound nearly what the pairs themselves do.
    // (From n = 257 on a block that the pair test would have dropped costs more than the test:
    // full_sets at n = 637 5.3 ms against 3.6.)
#pragma unroll 1
    for (int z4 = zlo; z4 < zhi; z4 += 4) {
      const int zz = z4 + (lane >> 4);
      const bool p = zz < zhi && ((mask >> (zz - zlo)) & 1ull) != 0ull && rw.row_block(ea, zz, tab);
      const unsigned long long pb = __ballot(p);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((pb >> (16 * q)) & 0xffffull) todo |= 1ull << (z4 + q - zlo);     // wave-uniform
    }
  } else {
    unsigned rows = 0u;
    todo = many_pairs<D, SINGLE>(rw, ea, zlo, mask, tab, lane, rows);
  }
#pragma unroll 1
  while (todo != 0ull) {
    zpack = 0u;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      if (todo != 0ull) {
        zpack |= unsigned(__builtin_amdgcn_readfirstlane(__builtin_ctzll(todo))) << (8 * q);
        todo &= todo - 1ull;
        nq = q + 1;
        EXPM_STAT(9, 1);
      }
    }
    flush();
  }
}

// One GP of a launch (see k_expander_many).
template <int D, int MODE, bool SINGLE>
__device__ __forceinline__ void many_gp(const GpDev* gps, int g, int G, const SweepPoints& pts,
                                        const ExpanderArgs& ea, int ngroups, const double* tab,
                                        double* kbw, double* rowbuf, double* red, int lane, int wave) {
  const int64_t nwaves = (pts.N + 15) >> 4;        // 16-row segments of the shard
  auto rows_of = [&](int64_t rrow, bool unsafe) {
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
      x[k] = pts.base[rrow * pts.stride_row + k * pts.stride_col];
    return ManyRows<D, SINGLE>(gps[g], g, G, ea, x, ea.mean[int64_t(g) * pts.N + rrow],
                               ea.var[int64_t(g) * pts.N + rrow], unsafe);
  };
  if (MODE == 0) {
    const int64_t wid = int64_t(blockIdx.x) * 4 + wave;
    const int64_t row = wid * 16 + (lane & 15);
    const bool valid = row < pts.N;
    const int64_t rrow = valid ? row : pts.N - 1;
    const bool unsafe = valid && (ea.S[rrow] == 0);
    if (__ballot(unsafe) == 0ull) return;          // (wave-uniform)
    const ManyRows<D, SINGLE> rw = rows_of(rrow, unsafe);
    // (the block test alone, 64 groups at a time: no wave of the scan waits for a chain of
    // pair tests -- the waves next to the band pass the block test for most groups and the
    // pair test for none)
    // Two levels: the SUPERGROUPS of 8 consecutive groups first, one per lane (a far segment is
    // done after one test per 64 x 128 candidates), then the groups of the supergroups that
    // pass, eight supergroups at a time (lane = 8 supergroup + group).
    bool some = false;
    EXPM_STAT(0, 1); EXPM_STAT(1, __popcll(__ballot(unsafe) & 0xffffull));
    const int nsuper = (ngroups + 7) >> 3;
#pragma unroll 1
    for (int Z0 = 0; Z0 < nsuper && !some; Z0 += 64) {
      const int ZZ = Z0 + lane;
      unsigned long long sm = __ballot(ZZ < nsuper && rw.block_at(ea.sbox, ea.sagg, ZZ, tab));
      EXPM_STAT(2, min(64, nsuper - Z0));
#pragma unroll 1
      while (sm != 0ull && !some) {
        int mine = -1;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
          const int bpos = sm != 0ull ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(sm)) : -1;
          if ((lane >> 3) == sidx) mine = bpos;
          sm &= sm - (sm != 0ull ? 1ull : 0ull);
        }
        const int zz = 8 * (Z0 + mine) + (lane & 7);
        some = __ballot(mine >= 0 && zz < ngroups && rw.block(ea, zz, tab)) != 0ull;
        EXPM_STAT(4, 1);
      }
    }
    EXPM_STAT(3, some);
    if (some && lane == 0) {
      const int at = atomicAdd(&ea.wcount[g], 1);
      ea.wlist[int64_t(g) * nwaves + at] = int(wid);
      ea.wmask[int64_t(g) * nwaves + at] = 0u;
    }
  } else if (MODE == 2) {
    const int nch = (ngroups + kManyTestChunk - 1) / kManyTestChunk;
    const int64_t total = int64_t(ea.wcount[g]) * nch;
#pragma unroll 1
    for (int64_t item = int64_t(blockIdx.x) * 4 + wave; item < total; item += int64_t(gridDim.x) * 4) {
      const int hw = int(item / nch), gc = int(item - int64_t(hw) * nch);
      const int64_t row = int64_t(ea.wlist[int64_t(g) * nwaves + hw]) * 16 + (lane & 15);
      const bool valid = row < pts.N;
      const int64_t rrow = valid ? row : pts.N - 1;
      const bool unsafe = valid && (ea.S[rrow] == 0);
      const ManyRows<D, SINGLE> rw = rows_of(rrow, unsafe);
      const int zlo = gc * kManyTestChunk, zhi = min(ngroups, (gc + 1) * kManyTestChunk);
      unsigned hot = 0u;
      if (SINGLE) {
        // row against group, four groups at a time (lane = 16 group + row): no pair tests here
        bool some = false;
#pragma unroll 4
        for (int zz = zlo + (lane >> 4); zz < zhi; zz += 4) some = rw.row_block(ea, zz, tab) || some;
        const unsigned long long pb = __ballot(some);
        hot = unsigned((pb | (pb >> 16) | (pb >> 32) | (pb >> 48)) & 0xffffull);
      } else {
        static_assert(kManyTestChunk <= 64, "one mask per item");
        many_pairs<D, SINGLE>(rw, ea, zlo, (zhi - zlo >= 64) ? ~0ull : ((1ull << (zhi - zlo)) - 1ull),
                              tab, lane, hot);
      }
      EXPM_STAT(16, 1); EXPM_STAT(17, __popc(hot));
      if (hot != 0u) {                           // (wave-uniform) the rows no other item listed
        int base = 0;
        if (lane == 0) {
          hot &= ~atomicOr(&ea.wmask[int64_t(g) * nwaves + hw], hot);
          if (hot != 0u) base = atomicAdd(&ea.count[g], __popc(hot));
        }
        hot = __builtin_amdgcn_readfirstlane(hot);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < 16 && ((hot >> lane) & 1u))
          ea.list[int64_t(g) * pts.N + base + __popc(hot & ((1u << lane) - 1u))] = int(rrow);
      }
    }
  } else {
    // an item = 16 listed rows x a chunk of groups.  Few items (one round of the chip's
    // workgroups at 4 groups each): an item per WORKGROUP, its waves split the training points
    // (many_rows, coop); otherwise an item of kManyChunk groups per wave.
    const int cnt = ea.count[g];
    const int nrb = (cnt + 15) >> 4;
    const bool coop = int64_t(nrb) * ((ngroups + 3) >> 2) <= kManyCoopItems;
    const int chunk = coop ? 4 : kManyChunk;
    const int nch = (ngroups + chunk - 1) / chunk;
    const int64_t total = int64_t(nrb) * nch;
    // (coop: the item and the trip count are uniform over the workgroup -- barriers inside)
    // Items in chunk-major order, and a contiguous range of them per XCD (workgroup b runs on
    // XCD b % 8, each with its own L2): the waves in flight at a time then contract the SAME
    // few chunks against different rows -- the groups' A operands (n x 16 doubles each, 10 MB
    // a pass of 6000 candidates at n = 217) are fetched into an L2 once instead of once per
    // 16 rows.
    const int per = coop ? 1 : 4;                                  // items of a workgroup at a time
    const int64_t share = (total + 7) >> 3;                        // of an XCD
    const int64_t xend = min(total, share * ((blockIdx.x & 7) + 1));
    const int64_t step = int64_t(gridDim.x >> 3) * per;
#pragma unroll 1
    for (int64_t item = share * (blockIdx.x & 7) + int64_t(blockIdx.x >> 3) * per + (coop ? 0 : wave);
         item < xend; item += step) {
      const int gc = int(item / nrb), rb = int(item - int64_t(gc) * nrb);
      const int idx = rb * 16 + (lane & 15);
      const bool unsafe = idx < cnt;
      const int64_t rrow = ea.list[int64_t(g) * pts.N + (unsafe ? idx : cnt - 1)];
      const ManyRows<D, SINGLE> rw = rows_of(rrow, unsafe);
      many_rows<D, SINGLE>(gps[g], rw, ea, gc * chunk, min(ngroups, (gc + 1) * chunk), tab,
                           kbw, rowbuf, red, lane, wave, coop);
    }
  }
}

template <int D, int MODE>
__global__ __launch_bounds__(256) void k_expander_many(const GpDev* gps, int G,
                                                       SweepPoints pts, ExpanderArgs ea,
                                                       int ngroups) {
  __shared__ double tab[kExpTabSize];
  __shared__ __attribute__((aligned(16))) double kbuf[MODE == 1 ? 4 : 1][4 * kManyKbRow];   // B-operand transposes
  __shared__ double rows_sh[MODE == 1 ? 4 : 1][16 * (D + 3)];
  __shared__ double red[MODE == 1 ? 4 * 4 * 4 * 64 : 1];     // [wave][block][m4][lane] partial products
  exp_tab_init(tab);
  __syncthreads();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  double* kbw = kbuf[MODE == 1 ? wave : 0];
  double* rowbuf = rows_sh[MODE == 1 ? wave : 0];
  for (int g = 0; g < G; ++g) {
    if (!ea.active[g]) continue;
    if (__builtin_amdgcn_readfirstlane(int(gps[g].kern.n_parts == 1)))
      many_gp<D, MODE, true>(gps, g, G, pts, ea, ngroups, tab, kbw, rowbuf, red, lane, wave);
    else
      many_gp<D, MODE, false>(gps, g, G, pts, ea, ngroups, tab, kbw, rowbuf, red, lane, wave);
  }
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
  // (the row is fetched together with its S flag, not after it)
  const uint8_t sflag = ea.S[rrow];
  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k)
    x[k] = pts.base[rrow * pts.stride_row + k * pts.stride_col];
  const bool unsafe = valid && (sflag == 0);
  if (__ballot(unsafe) == 0ull) return;
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
      // |c(x)| <= |k(x, x_c)| + |L^-1 k_x| |L^-1 k_c| and, Cauchy-Schwarz on the POSTERIOR
      // covariance, |c(x)| <= sd(x) sd(x_c): the second bound drops the rows an observation
      // has pinned below fmin, which the first one lists for every candidate
      const double tn2c = ea.tn2[g * 16];
      const double cmax =
          fmin(fabs(kxc) + sqrt(qx * tn2c),
               sqrt((var + 1e-12 * kdiag) * (fmax(kdiag - tn2c, 0.0) + 1e-12 * kdiag))) *
          (1.0 + 1e-9);
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

constexpr int kExpLds = 6144;     // doubles of staged training data, at most (48 KB)

template <int D>
__global__ __launch_bounds__(256) void k_expander_list(const GpDev* gps, int G,
                                                       SweepPoints pts,
                                                       ExpanderArgs ea,
                                                       const int* count,
                                                       const int* list,
                                                       int stage_cap) {
  __shared__ double tab[kExpTabSize];
  extern __shared__ double stage[];      // [n_pad][D] scaled rows | [n_pad] w
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
    const bool staged = np * (D + 1) <= stage_cap;      // block-uniform
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
  // GPs that share the factor of the GP in front of them (GpDev::share: same inputs,
  // kernel, noise and fitting history -- the outputs of a multi-output GP, with their one
  // new observation at the same x*) have the same w and the same k(X, x), hence the same
  // c(x): it is computed for the first of them and kept; only (y* - mu) / s^2 differs.
  bool have_cx = false;
  double cx_keep = 0.0;
  for (int g = 0; g < G; ++g) {
    double mean = ra.mean[int64_t(g) * pts.N + rrow];
    double var = ra.var[int64_t(g) * pts.N + rrow];
    if (ra.which[g] && gps[g].share >= 0 && have_cx) {
      const GpDev& gp = gps[g];
      mean = fma(cx_keep, gp.upd[0], mean);
      var = fmax(var - cx_keep * cx_keep * gp.upd[1], 1e-15);
      if (writer) {
        ra.mean[int64_t(g) * pts.N + row] = mean;
        ra.var[int64_t(g) * pts.N + row] = var;
      }
    } else if (ra.which[g]) {
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
      cx_keep = cx;
      have_cx = true;
      mean = fma(cx, gp.upd[0], mean);
      var = fmax(var - cx * cx * gp.upd[1], 1e-15);
      if (writer) {
        ra.mean[int64_t(g) * pts.N + row] = mean;
        ra.var[int64_t(g) * pts.N + row] = var;
      }
    } else {
      have_cx = false;
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


// ---- launch -----------------------------------------------------------------------
// Waves per workgroup: 4 (two workgroups per CU).
int sweep_waves() { return 4; }

// The stage sequence of one tile: for every GP, for every chunk of 16 row blocks
// of L^-1, the j-blocks 0 .. bend-1 (only the slots at or below the diagonal are
// active).  Depends on the block counts only, so it is rebuilt (and uploaded)
// when a GP crosses a multiple of 16 training points, not per launch.
// Narrow 4-row groups the last row block of a GP is taken as (0: a full block).
// SGP_NO_NARROW=1: never (A/B runs); =2: only the one-group form of round 3.
int narrow_groups_of(const GpDev& gp) {
  static const int mode = getenv("SGP_NO_NARROW") ? atoi(getenv("SGP_NO_NARROW")) : 0;
  const int ng = (gp.last_rows + 3) / 4;
  if (mode == 1 || ng > kMaxNg) return 0;
  if (mode == 2 && ng > 1) return 0;
  return ng;
}

int stage_table(sgp_ctx* ctx, const GpDev* gh, int Geff, int d, bool sep, int kIB,
                const bool* rides, const StageEnt** dev, int* nstages) {
  // (entries hold absolute addresses: rebuilt when a block count OR a buffer address
  // changes; the buffers are sized for the pitch of L^-1, so one-row appends keep them)
  std::vector<uint64_t> sig(1, uint64_t(Geff));
  sig.push_back(uint64_t(kIB) | (uint64_t(d) << 8) | (uint64_t(sep) << 16));
  int last_staged = 0;
  for (int g = 0; g < Geff; ++g) {
    sig.push_back(uint64_t(gh[g].nblk));
    sig.push_back(uint64_t(narrow_groups_of(gh[g])) | (uint64_t(rides[g]) << 2));
    sig.push_back(reinterpret_cast<uint64_t>(gh[g].Apack));
    sig.push_back(reinterpret_cast<uint64_t>(gh[g].XA));
    if (!rides[g]) last_staged = g;
  }
  if (sig == ctx->stage_sig && ctx->stage_tab.p) {
    *dev = static_cast<const StageEnt*>(ctx->stage_tab.p);
    *nstages = ctx->stage_count;
    return 0;
  }
  std::vector<StageEnt> tab;
  const uint64_t xa_block = uint64_t(16 * d + 16) * sizeof(double);
  uint32_t slots_total = 0;       // 2-KB positions of all stages (resident mode)
  for (int g = 0; g < Geff; ++g) {
    if (rides[g]) continue;       // (its alpha . k is formed in its leader's stages)
    const int nblk = gh[g].nblk, nsteps = gh[g].n_pad / 4;
    const int nchunks = (nblk + kIB - 1) / kIB;
    const uint64_t apack = reinterpret_cast<uint64_t>(gh[g].Apack);
    const uint64_t xa = reinterpret_cast<uint64_t>(gh[g].XA) + (sep ? 128u * uint64_t(d) : 0u);
    for (int c = 0; c < nchunks; ++c) {
      const int b0 = c * kIB, nib = std::min(kIB, nblk - b0), bend = b0 + nib;
      for (int jb = 0; jb < bend; ++jb) {
        StageEnt e{};
        e.a_src = apack + (uint64_t(bend - 1) * nsteps + 4 * uint64_t(jb)) * 512;
        e.xa = xa + uint64_t(jb) * xa_block;
        e.rs_bytes = uint32_t(nsteps) * 512u;
        e.jb = uint32_t(jb);
        e.word = uint32_t(std::min(nib, bend - jb)) | (uint32_t(g) << SW_G_SHIFT);
        e.slot0 = slots_total;
        slots_total += uint32_t(std::min(nib, bend - jb));
        if (jb == 0) e.word |= SW_FIRST;
        e.word |= uint32_t(nib) << SW_NSL_SHIFT;
        if (jb == bend - 1) e.word |= SW_CHUNK_END;
        if (c == nchunks - 1) e.word |= SW_MEAN;
        // slot 0 of the last chunk = the last row block
        if (c == nchunks - 1 && narrow_groups_of(gh[g]) > 0)
          e.word |= SW_NARROW | (uint32_t(narrow_groups_of(gh[g])) << SW_NGRP_SHIFT);
        if (c == nchunks - 1 && jb == bend - 1) {
          e.word |= SW_GP_END;
          if (g == last_staged) e.word |= SW_TILE_END;
        }
        tab.push_back(e);
      }
    }
  }
  SGP_TRY(sgp_reserve(ctx, &ctx->stage_tab, tab.size() * sizeof(StageEnt)));
  SGP_TRY(sgp_h2d(ctx, ctx->stage_tab.p, tab.data(), tab.size() * sizeof(StageEnt)));
  ctx->stage_sig = sig;
  ctx->stage_count = int(tab.size());
  ctx->stage_slots = int(slots_total);
  *dev = static_cast<const StageEnt*>(ctx->stage_tab.p);
  *nstages = ctx->stage_count;
  return 0;
}

// persistent: as many workgroups as are resident at once (256 VGPRs per thread
// -> 8 waves per CU) walk over the tiles
int sweep_grid_blocks(int num_cu, int64_t N, int nw, int slots) {
  (void)slots;
  const int64_t ntiles = (N + 16 * nw - 1) / (16 * nw);
  const int64_t resident = int64_t(num_cu) * (8 / nw);
  return int(ntiles < resident ? ntiles : resident);
}

template <int D, int NW, int SL, int MODE, bool SINGLE, int R = 0, int SEP = 0>
int launch_sweep_v(sgp_ctx* ctx, const SweepParams& p, double flops) {
  static bool attr_set = false;
  if (!attr_set) {
    SGP_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(&k_sweep<D, NW, SL, MODE, SINGLE, R, SEP, false>),
                     hipFuncAttributeMaxDynamicSharedMemorySize,
                     int(Lay<SL, D, R>::bytes(NW))));
    SGP_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(&k_sweep<D, NW, SL, MODE, SINGLE, R, SEP, true>),
                     hipFuncAttributeMaxDynamicSharedMemorySize,
                     int(Lay<SL, D, R>::bytes(NW))));
    attr_set = true;
  }
  const int nblocks = sweep_grid_blocks(ctx->num_cu, p.pts.N, NW, SL);
  SweepTimer timer;
  SGP_TRY(timer.begin(ctx, flops));
  SweepParams pp = p;
  {
    // small factors: every stage's positions stay in LDS for the whole launch
    // (SGP_NO_RESIDENT=1 / sgp_ctx_set_sweep(+ 16): stream them, A/B runs)
    typedef Lay<SL, D, R> L;
    static const bool off = getenv("SGP_NO_RESIDENT") != nullptr;
    const size_t xblk = size_t(L::kXTile + kJC * (1 + R)) * 8;
    const size_t need = size_t(ctx->stage_slots) * 2048 + size_t(p.nstages) * xblk;
    pp.resident = (!off && !(ctx->sweep_choice & 16) && need <= size_t(2 * L::kBuf) * 8) ? 1 : 0;
    pp.res_xbase = unsigned(ctx->stage_slots) * 2048u;
  }
#ifdef SGP_INSTRUMENT
  static const int ablate = getenv("SGP_ABLATE") ? atoi(getenv("SGP_ABLATE")) : 0;
  pp.ablate = ablate;
#endif
  const size_t lds_bytes = Lay<SL, D, R>::bytes(NW);
#ifdef SGP_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  if (!stamps_dev) SGP_HIP(ctx, hipMalloc(&stamps_dev, size_t(4096) * NW * 8 * 8));
  pp.stamps = stamps_dev;
#endif
  if (pp.resident)
    hipLaunchKernelGGL((k_sweep<D, NW, SL, MODE, SINGLE, R, SEP, true>), dim3(nblocks),
                       dim3(64 * NW), lds_bytes, ctx->stream, pp);
  else
    hipLaunchKernelGGL((k_sweep<D, NW, SL, MODE, SINGLE, R, SEP, false>), dim3(nblocks),
                       dim3(64 * NW), lds_bytes, ctx->stream, pp);
  SGP_HIP(ctx, hipGetLastError());
#ifdef SGP_STAMPS
  {
    std::vector<unsigned long long> h(size_t(nblocks) * NW * 8);
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGP_HIP(ctx, hipMemcpy(h.data(), stamps_dev, h.size() * 8, hipMemcpyDeviceToHost));
    static const char* names[8] = {"prefetch", "evaluate", "narrow", "transpose", "slots",
                                   "fold+epilogue", "dma-wait", "barrier"};
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
    for (size_t w = 0; w < size_t(nblocks) * NW; ++w)
      for (int i = 0; i < 8; ++i) sum[i] += double(h[w * 8 + i]);
    for (int i = 0; i < 8; ++i) tot += sum[i];
    fprintf(stderr, "stamps (ticks per wave, %% of loop):");
    for (int i = 0; i < 8; ++i)
      fprintf(stderr, "  %s %.0f (%.1f%%)", names[i], sum[i] / (double(NW) * nblocks),
              100.0 * sum[i] / tot);
    fprintf(stderr, "  | loop %.0f\n", tot / (double(NW) * nblocks));
  }
#endif
  return timer.end(ctx);
}

template <int D, int NW, int SL>
int launch_sweep_w(sgp_ctx* ctx, const SweepParams& p, double flops) {
  if constexpr (D <= 3) {       // (the d = 4 instance with riders would spill)
    bool riders = false;
    for (int g = 0; g < SGP_MAX_GPS; ++g) riders = riders || p.nride[g] > 0;
    if (riders) return launch_sweep_v<D, NW, SL, MODE_CONF, true, kSweepRide>(ctx, p, flops);
  }
  return p.single ? launch_sweep_v<D, NW, SL, MODE_CONF, true>(ctx, p, flops)
                  : launch_sweep_v<D, NW, SL, MODE_CONF, false>(ctx, p, flops);
}

template <int D>
int launch_sweep_d(sgp_ctx* ctx, const SweepParams& p, double flops) {
  return launch_sweep_w<D, 4, 16>(ctx, p, flops);
}

// tensor grid + RBF kernels: the instances that read factor tables (SEP axes; D = 1:
// they do not read the rows).  Riders as in the generic instances.
template <int SEP>
int launch_sweep_sep(sgp_ctx* ctx, const SweepParams& p, double flops) {
  bool riders = false;
  for (int g = 0; g < SGP_MAX_GPS; ++g) riders = riders || p.nride[g] > 0;
  if (riders) return launch_sweep_v<1, 4, 16, MODE_CONF, true, kSweepRide, SEP>(ctx, p, flops);
  return launch_sweep_v<1, 4, 16, MODE_CONF, true, 0, SEP>(ctx, p, flops);
}

int launch_posterior(sgp_ctx* ctx, const SweepParams& p, const GpDev* gh, int d, int Geff,
                     double flops, const SepLaunch* sep, bool rows_sharded);

// rows_sharded: the rows are a rank's shard of a grid (sgp_grid_*) -- the kernel is then
// chosen by the GPs alone, never by the number of rows (same kernel on every rank)
int launch_sweep(sgp_ctx* ctx, const SweepParams& p, const GpDev* gh, int d,
                 const SepLaunch* sep = nullptr, bool rows_sharded = false) {
  // algorithmic flops: G * (n^2 + 2n) per row
  double flops = 0.0;
  const int Geff =
      (p.mode == MODE_FITNESS && p.fit.swarm_type == SGP_SWARM_GREEDY) ? 1
                                                                       : p.G;
  for (int g = 0; g < Geff; ++g)
    flops += (double(gh[g].n) * gh[g].n + 2.0 * gh[g].n) * double(p.pts.N);
  if (p.pts.N <= 0) return 0;
  SweepParams q = p;
  const bool fitness = p.mode == MODE_FITNESS;
  if (fitness) {
    // SafeOptSwarm._compute_particle_fitness (gp_opt.py:901-1013) = the posterior of
    // the swarm's GPs (the sweep, mean / var only) + the shaping of fitness.h on
    // those (k_fitness_small: one thread per particle).  The particles are few next
    // to a grid: the 16 bytes per (GP, particle) in between cost nothing, and the
    // sweep kernels need no second set of instances (which spilled registers).
    const size_t np = size_t(Geff) * size_t(p.pts.N);
    SGP_TRY(sgp_reserve(ctx, &ctx->pair_post, 2 * np * sizeof(double)));
    q.mode = MODE_CONF;
    q.conf = ConfOut{};
    q.conf.mean = static_cast<double*>(ctx->pair_post.p);
    q.conf.var = q.conf.mean + np;
    q.G = Geff;
    for (int i = 0; i < SGP_MAX_GPS; ++i) q.conf.fmin[i] = -INFINITY;
  }
  int rc = launch_posterior(ctx, q, gh, d, Geff, flops, sep, rows_sharded);
  if (rc != 0 || !fitness) return rc;
  return launch_fitness_small(ctx, p.G, p.pts.N, q.conf.mean, q.conf.var, p.fit);
}

// The confidence sweep proper: the paired-wave kernel from 257 rows of L^-1 on, the 4-wave
// kernel below, the resident-factor kernel (sweep_mid.hip) for 49 .. 128 observations of
// single-part kernels, the VALU kernel (sweep_tiny.hip) up to 48 observations.
int launch_posterior(sgp_ctx* ctx, const SweepParams& p, const GpDev* gh, int d, int Geff,
                     double flops, const SepLaunch* sep, bool rows_sharded) {
  if (tiny_sweep_wanted(ctx, gh, Geff, p.pts.N, rows_sharded)) {
    ctx->last_sweep = 3;
    SweepArgs a{p.gps, p.G, p.mode, p.pts, p.conf, p.fit};
    return launch_sweep_tiny(ctx, a, gh, d, Geff, flops);    // (sets ctx->sweep_partials)
  }
  if (mid_sweep_wanted(ctx, gh, Geff, d) || mid_passes_wanted(ctx, gh, Geff, sep, p.conf)) {
    ctx->last_sweep = 6;
    SweepArgs a{p.gps, p.G, p.mode, p.pts, p.conf, p.fit};
    return launch_sweep_mid(ctx, a, gh, d, Geff, flops, sep);   // (sets ctx->sweep_partials)
  }
  if (pair_sweep_wanted(ctx, gh, Geff)) {
    ctx->last_sweep = 2;
    SweepArgs a{p.gps, p.G, p.mode, p.pts, p.conf, p.fit};
    return launch_sweep_pair(ctx, a, gh, d, Geff, flops, sep);   // (sets ctx->sweep_partials)
  }
  ctx->last_sweep = 1;
  ctx->sweep_partials = sweep_grid_blocks(ctx->num_cu, p.pts.N, sweep_waves(), 16) *
                        sweep_waves();
  SweepParams q = p;
  q.slots = 16;            // accumulator slots per wave
  q.single = 1;
  for (int g = 0; g < Geff; ++g) q.single = q.single && gh[g].kern.n_parts == 1;
  // followers of a shared factor ride in their leader's stages (sweep_shared.h)
  bool rides[SGP_MAX_GPS] = {};
  static const bool no_ride = getenv("SGP_PAIR_RIDE") && atoi(getenv("SGP_PAIR_RIDE")) == 0;
  for (int g = 0; g < SGP_MAX_GPS; ++g) q.nride[g] = 0;
  // (tensor-grid instances: one set of tables per leader, whatever the input dimension)
  if (no_ride || !sweep_riders(gh, Geff, sep ? 1 : d, q.single != 0 || sep != nullptr,
                               kSweepRide, 3, rides, q.nride))
    for (int g = 0; g < SGP_MAX_GPS; ++g) {
      rides[g] = false;
      q.nride[g] = 0;
    }
  for (int g = 0, leader = 0; g < Geff; ++g) {
    q.ride_delta[g] = 0;
    if (!rides[g]) {
      leader = g;
      continue;
    }
    q.ride_delta[g] = (long long)(reinterpret_cast<intptr_t>(gh[g].XA) -
                                  reinterpret_cast<intptr_t>(gh[leader].XA));
  }
  SGP_TRY(stage_table(ctx, gh, Geff, d, sep != nullptr, q.slots, rides, &q.stages, &q.nstages));
  if (sep) {
    q.sep = *sep;
    switch (sep->naxes) {
      case 1: return launch_sweep_sep<1>(ctx, q, flops);
      case 2: return launch_sweep_sep<2>(ctx, q, flops);
      case 3: return launch_sweep_sep<3>(ctx, q, flops);
      // (4 axes: 32 registers of factors in flight -- the instance would spill; the
      // covariances are evaluated, below)
    }
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

// Number of partials of max l0[S] the LAST confidence sweep left in
// ConfOut::partial (one per wave / wave pair of every launched workgroup).
int sweep_num_partials(const sgp_ctx* ctx, int64_t N) {
  (void)N;
  return ctx->sweep_partials;
}

int launch_sweep_conf(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                      int G, int d, SweepPoints pts, ConfOut out, const SepLaunch* sep,
                      bool rows_sharded) {
  SweepParams p{};
  p.gps = gps_dev;
  p.G = G;
  p.mode = MODE_CONF;
  p.pts = pts;
  p.conf = out;
  p.fit = FitnessArgs{};
  return launch_sweep(ctx, p, gps_host, d, sep, rows_sharded);
}

int launch_sweep_fitness(sgp_ctx* ctx, const GpDev* gps_dev,
                         const GpDev* gps_host, int G, int d, SweepPoints pts,
                         FitnessArgs fa) {
  SweepParams p{};
  p.gps = gps_dev;
  p.G = G;
  p.mode = MODE_FITNESS;
  p.pts = pts;
  p.conf = ConfOut{};
  p.fit = fa;
  return launch_sweep(ctx, p, gps_host, d);
}


// |L^-1 k_c| and sd(x_c) of every candidate of a pass, from tn2 = |L^-1 k_c|^2
__global__ void k_pass_aux(const GpDev* gps, int G, const double* tn2, double* stn, double* svc,
                           int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int g = (e >> 4) % G;
  const double t = tn2[e];
  stn[e] = sqrt(fmax(t, 0.0)) * (1.0 + 1e-12);
  svc[e] = sqrt(fmax(gps[g].kern.kdiag - t, 0.0) + 1e-12 * gps[g].kern.kdiag);
}

// Per group of 16 candidates: the largest |delta|, 1 / s2, |L^-1 k_c|, sd(x_c) of every GP
// (agg[(z G + g) 4 ..]) and the bounding box of the candidates' rows (box[z][2][d], raw
// coordinates) -- what lets a wave decide for a whole (16 rows x 16 candidates) block at once.
__global__ void k_pass_agg(int G, int d, int m_total, const double* xc, const double* delta,
                           const double* inv_s2, const double* stn, const double* svc,
                           double* agg, double* box, int ngroups) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ngroups * G) {
    const int z = e / G;
    const int m = min(16, m_total - 16 * z);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int c = 0; c < m; ++c) {
      a0 = fmax(a0, fabs(delta[int64_t(e) * 16 + c]));
      a1 = fmax(a1, inv_s2[int64_t(e) * 16 + c]);
      a2 = fmax(a2, stn[int64_t(e) * 16 + c]);
      a3 = fmax(a3, svc[int64_t(e) * 16 + c]);
    }
    agg[int64_t(e) * 4 + 0] = a0;
    agg[int64_t(e) * 4 + 1] = a1;
    agg[int64_t(e) * 4 + 2] = a2;
    agg[int64_t(e) * 4 + 3] = a3;
  }
  if (e < ngroups * d) {
    const int z = e / d, k = e - z * d;
    const int m = min(16, m_total - 16 * z);
    double lo = INFINITY, hi = -INFINITY;
    for (int c = 0; c < m; ++c) {
      const double v = xc[(int64_t(z) * 16 + c) * d + k];
      lo = fmin(lo, v);
      hi = fmax(hi, v);
    }
    box[(int64_t(z) * 2 + 0) * d + k] = lo;
    box[(int64_t(z) * 2 + 1) * d + k] = hi;
  }
}

// ... and the same per SUPERGROUP of 8 consecutive groups, from the groups' values (the scan of
// the grid tests those first: k_expander_many, MODE 0)
__global__ void k_pass_sagg(int G, int d, const double* agg, const double* box, double* sagg,
                            double* sbox, int ngroups) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int nsuper = (ngroups + 7) >> 3;
  if (e < nsuper * G) {
    const int Z = e / G, g = e - Z * G;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int z = 8 * Z; z < min(8 * Z + 8, ngroups); ++z)
      for (int q = 0; q < 4; ++q) a[q] = fmax(a[q], agg[(int64_t(z) * G + g) * 4 + q]);
    for (int q = 0; q < 4; ++q) sagg[int64_t(e) * 4 + q] = a[q];
  }
  if (e < nsuper * d) {
    const int Z = e / d, k = e - Z * d;
    double lo = INFINITY, hi = -INFINITY;
    for (int z = 8 * Z; z < min(8 * Z + 8, ngroups); ++z) {
      lo = fmin(lo, box[(int64_t(z) * 2 + 0) * d + k]);
      hi = fmax(hi, box[(int64_t(z) * 2 + 1) * d + k]);
    }
    sbox[(int64_t(Z) * 2 + 0) * d + k] = lo;
    sbox[(int64_t(Z) * 2 + 1) * d + k] = hi;
  }
}

int launch_expander_many(sgp_ctx* ctx, const GpDev* gps_dev, int G, int d, SweepPoints pts,
                         ExpanderArgs ea) {
  if (pts.N <= 0 || ea.m <= 0) return 0;
  const int nblocks = int((pts.N + 63) / 64);
  const int ngroups = (ea.m + 15) / 16;
  // the waves with a possible pair and the rows that pass the pair test for some candidate, per
  // GP: counts | [G][N / 16] waves | [G][N / 16] masks of their listed rows | [G][N] rows
  {
    const size_t nw = size_t((pts.N + 15) >> 4);
    int* hot = static_cast<int*>(sgp_scratch(ctx, 12, (64 + size_t(G) * (2 * nw + size_t(pts.N))) * sizeof(int)));
    SGP_CHECK(ctx, hot, "device allocation failed: %s", ctx->err.c_str());
    ea.count = hot;
    ea.wcount = hot + 32;
    ea.wlist = hot + 64;
    ea.wmask = reinterpret_cast<unsigned*>(hot + 64 + size_t(G) * nw);
    ea.list = hot + 64 + 2 * size_t(G) * nw;
    SGP_HIP(ctx, hipMemsetAsync(hot, 0, 64 * sizeof(int), ctx->stream));
  }
  {
    const int n = ngroups * G * 16;
    hipLaunchKernelGGL(k_pass_aux, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, gps_dev, G,
                       ea.tn2, const_cast<double*>(ea.stn), const_cast<double*>(ea.svc), n);
    const int na = ngroups * (G > d ? G : d);
    hipLaunchKernelGGL(k_pass_agg, dim3((na + 255) / 256), dim3(256), 0, ctx->stream, G, d, ea.m,
                       ea.xc, ea.delta, ea.inv_s2, ea.stn, ea.svc, const_cast<double*>(ea.agg),
                       const_cast<double*>(ea.box), ngroups);
    const int ns = ((ngroups + 7) / 8) * (G > d ? G : d);
    hipLaunchKernelGGL(k_pass_sagg, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, G, d, ea.agg,
                       ea.box, const_cast<double*>(ea.sagg), const_cast<double*>(ea.sbox), ngroups);
  }
#define EXPM_CASE(DD)                                                         \
  case DD:                                                                    \
    hipLaunchKernelGGL((k_expander_many<DD, 0>), dim3(nblocks), dim3(256), 0, \
                       ctx->stream, gps_dev, G, pts, ea, ngroups);            \
    hipLaunchKernelGGL((k_expander_many<DD, 2>), dim3(kManyListBlocks),       \
                       dim3(256), 0, ctx->stream, gps_dev, G, pts, ea,        \
                       ngroups);                                              \
    hipLaunchKernelGGL((k_expander_many<DD, 1>), dim3(kManyListBlocks),       \
                       dim3(256), 0, ctx->stream, gps_dev, G, pts, ea,        \
                       ngroups);                                              \
    break;
  switch (d) {
    EXPM_CASE(1) EXPM_CASE(2) EXPM_CASE(3) EXPM_CASE(4)
    EXPM_CASE(5) EXPM_CASE(6) EXPM_CASE(7) EXPM_CASE(8)
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
#undef EXPM_CASE
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_expander_check(sgp_ctx* ctx, const GpDev* gps_dev,
                          const GpDev* gps_host, int G, int d, SweepPoints pts,
                          ExpanderArgs ea) {
  if (pts.N <= 0) return 0;
  const int nblocks = int((pts.N + 63) / 64);
  const bool listed = ea.m == 1 && ea.count && ea.list;
  // the listed rows are a few per cent of the shard at most, and the loop over
  // the training points is a chain of LDS latencies: as many waves per CU as the
  // staged training data (LDS) allows
  int np_max = 0;
  for (int g = 0; g < G; ++g)
    if (ea.active[g]) np_max = std::max(np_max, gps_host[g].n_pad);
  int stage_cap = np_max * (d + 1);
  if (stage_cap > kExpLds) stage_cap = 0;              // read from L2 instead
  const int per_cu = std::max(1, std::min(8, int(144 * 1024 / (stage_cap * 8 + 1024))));
  const int nfilter = int((pts.N + 255) / 256), nlist = ctx->num_cu * per_cu;
#define EXP_CASE(DD)                                                          \
  case DD:                                                                    \
    if (listed) {                                                             \
      hipLaunchKernelGGL(k_expander_filter<DD>, dim3(nfilter), dim3(256), 0,  \
                         ctx->stream, gps_dev, G, pts, ea, ea.count, ea.list);\
      hipLaunchKernelGGL(k_expander_list<DD>, dim3(nlist), dim3(256),         \
                         size_t(stage_cap) * 8, ctx->stream, gps_dev, G, pts, \
                         ea, ea.count, ea.list, stage_cap);                   \
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


// ---- tensor grids: per-axis factor tables (SepLaunch) -------------------------------
// E_a[jb][i][4 k4 + q] = 2^(-W_k (axis_k[i] - X_{j k})^2 / 32), k = column of axis a,
// j = 16 jb + 4 q + k4, W_k = sum over the parts of their squared weights on column k
// (KernDesc::wsq: the exponents of RBF parts add).  Axis 0's table also carries the
// product of the variances and the factors of the constant columns (contexts).
struct SepCols {
  int naxes, d;
  int cols[4];
  uint32_t count[SGP_MAX_D];
  int off[SGP_MAX_D];
};
__global__ __launch_bounds__(256) void k_sep_table(GpDev gp, SepCols sc, int a,
                                                   const double* vals, double* out) {
  const int k = sc.cols[a];
  const uint32_t count = sc.count[k];
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t total = int64_t(gp.nblk > kSepMinBlocks ? gp.nblk : kSepMinBlocks) * count * 16;
  if (e >= total) return;
  const int pos = int(e & 15);
  const uint32_t i = uint32_t((e >> 4) % count);
  const int jb = int((e >> 4) / count);
  const int j = 16 * jb + 4 * (pos & 3) + (pos >> 2);
  if (jb >= gp.nblk) {            // (beyond the GP's own blocks: see kSepMinBlocks)
    out[e] = 0.0;
    return;
  }
  const double* xj = gp.Xpad + int64_t(j) * sc.d;
  auto weight = [&](int col) {
    double W = 0.0;
    for (int p = 0; p < gp.kern.n_parts; ++p) W += gp.kern.wsq[p][col];
    return W;
  };
  double dx = vals[sc.off[k] + i] - xj[k];
  double u = weight(k) * dx * dx;
  double scale = 1.0;
  if (a == 0) {
    scale = gp.kern.kdiag;
    for (int c = 0; c < sc.d; ++c) {
      if (sc.count[c] != 1) continue;
      dx = vals[sc.off[c]] - xj[c];
      u += weight(c) * dx * dx;
    }
  }
  out[e] = scale * exp2(-u * (1.0 / 32.0));
}

size_t sep_table_doubles(const GpDev& gp, uint32_t count) {
  return size_t(std::max(gp.nblk, kSepMinBlocks)) * count * 16;
}

int launch_sep_tables(sgp_ctx* ctx, const GpDev& gp, int d, const uint32_t* count,
                      const double* axis_vals, const int* axis_off, int naxes,
                      const int* cols, double* const* out) {
  SepCols sc{};
  sc.naxes = naxes;
  sc.d = d;
  for (int a = 0; a < naxes; ++a) sc.cols[a] = cols[a];
  for (int k = 0; k < d; ++k) {
    sc.count[k] = count[k];
    sc.off[k] = axis_off[k];
  }
  for (int a = 0; a < naxes; ++a) {
    const int64_t total = int64_t(sep_table_doubles(gp, count[cols[a]]));
    hipLaunchKernelGGL(k_sep_table, dim3(unsigned((total + 255) / 256)), dim3(256), 0,
                       ctx->stream, gp, sc, a, axis_vals, out[a]);
  }
  SGP_HIP(ctx, hipGetLastError());
  return 0;
}

// rows of the grid against the declared axes, bit for bit
__global__ __launch_bounds__(256) void k_verify_axes(const double* pts, int64_t N, int d,
                                                     int64_t goff, const double* vals,
                                                     const uint32_t* meta, int* mismatch) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= N) return;
  const uint32_t gi = uint32_t(goff + i);
  int bad = 0;
  for (int k = 0; k < d; ++k) {
    const uint32_t cnt = meta[3 * k], str = meta[3 * k + 1], off = meta[3 * k + 2];
    const double want = vals[off + (gi / str) % cnt];
    bad |= __double_as_longlong(want) != __double_as_longlong(pts[int64_t(k) * N + i]);
  }
  if (bad) atomicAdd(mismatch, 1);
}

int launch_verify_axes(sgp_grid* g, int* mismatch_dev) {
  sgp_ctx* ctx = g->ctx;
  uint32_t meta[3 * SGP_MAX_D];
  for (int k = 0; k < g->d; ++k) {
    meta[3 * k] = g->ax_count[k];
    meta[3 * k + 1] = g->ax_stride[k];
    meta[3 * k + 2] = uint32_t(g->ax_off[k]);
  }
  uint32_t* meta_dev = reinterpret_cast<uint32_t*>(mismatch_dev + 4);   // (same scratch slot)
  SGP_TRY(sgp_h2d(ctx, meta_dev, meta, sizeof(uint32_t) * 3 * g->d));
  hipLaunchKernelGGL(k_verify_axes, dim3(unsigned((g->N + 255) / 256)), dim3(256), 0,
                     ctx->stream, g->pts, g->N, g->d, g->goff,
                     static_cast<const double*>(g->ax_vals.p), meta_dev, mismatch_dev);
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

