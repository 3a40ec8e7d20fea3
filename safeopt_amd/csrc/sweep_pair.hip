// posterior_sweep, paired-wave form: the kernel for factors with more than 256
// rows (BASELINE.json configs 3, 4, 5).
//
// Same mathematics as sweep.hip (gp.predict_noiseless + the Q update of
// SafeOpt.update_confidence_intervals, safeopt/gp_opt.py:453-481; the posterior
// half of SafeOptSwarm._compute_particle_fitness, :901-1013): for every candidate row
//     v = L^-1 k(X, x),  var = k(x,x) - |v|^2,  mean = alpha . k(X, x)
// with k(X, x) evaluated on the fly and L^-1 k on the fp64 matrix cores.
//
// What is different: a covariance value is evaluated ONCE per (row, training
// point) up to n = 512 (sweep.hip re-evaluates the j-blocks for every chunk of
// 256 rows of L^-1: 48 instead of 32 evaluations per tile at n = 500, 159
// instead of 63 at n = 1000), and each wave evaluates only HALF of its tile's
// values:
//   * workgroup = 8 waves = 4 PAIRS; the two waves of a pair (w, w + 4: the two
//     waves of one SIMD) own the SAME 16 candidate rows and split the row blocks
//     of L^-1 between them (global slot t of a chunk of 32 row blocks belongs to
//     wave t & 1): 2 x 16 accumulator slots cover 512 rows in ONE pass;
//   * wave h of a pair evaluates the training points 8 h .. 8 h + 7 of a
//     j-block (two values per lane instead of four) and writes them to the
//     pair's LDS buffer in MFMA B-operand order; both waves read all 16 from
//     there.  The exchange is software pipelined: the values of stage s + 1 are
//     evaluated during stage s, so the one barrier per stage that the staged
//     L^-1 chunk needs anyway also orders the exchange;
//   * BOTH waves of a pair evaluate first, right behind the stage barrier, then
//     both multiply: fp64 VALU and fp64 MFMA instructions do not overlap on a
//     SIMD (a VALU instruction under the partner's MFMA stream gets one issue slot
//     in 45..72 cycles), so the two VALU bursts run together at full rate and the
//     two MFMA streams after them (profiles/r03/experiments.txt, sections 3, 5);
//   * the waves of half 0 copy the next L^-1 chunk by LDS-DMA piecewise between
//     their accumulator slots; half 1 finishes the pair's rows (adds the two
//     partial |v|^2 and alpha . k, runs the row epilogue one stage later);
//   * GPs with the factor of the GP in front of them (the outputs of a multi-
//     output GP) ride in its stages: their alpha . k comes from the covariances
//     the leader evaluates anyway (kMaxRide).
// The stage sequence comes from a host-built table with ABSOLUTE source
// addresses (one scalar load per stage, no pointer arithmetic on the device);
// the training rows and alpha of a j-block are one contiguous block
// (GpDev::XA) and arrive with a single LDS-DMA instruction two stages ahead.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "sweep_shared.h"
#include "small_path.h"

namespace {

constexpr int kJC = 16;                // training points per stage (one j-block)
constexpr int kSteps = kJC / 4;        // MFMA k-steps per stage
constexpr int kPairSlots = 32;         // row blocks of L^-1 per accumulator chunk
constexpr int kWaveSlots = 16;         // ... per wave (128 accumulator VGPRs)
constexpr int kPairs = 4;              // pairs per workgroup (64 rows per tile)
constexpr int kTileRows = 16 * kPairs;
constexpr int kKbRow = 64;             // doubles per k-row of a B buffer (odd k-rows are stored with
                                       // the two halves of a q pair's 32 doubles swapped: no padding)
constexpr int kDuoUnits = 28;          // 2 KB units of L^-1 in a stage of TWO j-blocks: the last four
                                       // units of its chunk buffer hold the B buffers of the second one

// The structure below is what survived the round-3 experiments (phase orders, pairs
// on adjacent waves, where and by whom the LDS-DMA is issued, priorities, a second
// barrier, ...: profiles/r03/experiments.txt; the switches they were built with are
// in the history of this file up to commit d1566ec; those of the merged stages of round 6:
// profiles/r06/attic/pair_experiment_switches.diff, results in profiles/r06/experiments.txt).
// Debug builds: -DSGP_INSTRUMENT (ablation mask, PGP_ABL) and -DPGP_STAMPS (cycle stamps per
// phase).

// One stage of a tile: ONE j-block of a (GP, chunk) -- or TWO (round 6, "merged stages").
// The stages of a triangular chunk shrink from 32 active row blocks to one; a thin stage
// pays the same barrier, table entry, operand fetch and exchange as a full one.  So j-block
// jb (32 - jb active slots) is put together with a j-block from the other end of the
// chunk while both fit kDuoUnits: 19 stages of 28 .. 32 units instead of 32 stages of
// 32 .. 1 at n = 500 (pair_stage_table).  The two j-blocks are the SEGMENTS A and B of the
// stage: each is a prefix of the accumulator slots (global slot t = row block bend-1-t,
// k-steps 4 jb .. 4 jb + 3); in the chunk buffer segment A's slots are the units
// 0 .. nA-1 and segment B's the units nA .. nA+nB-1.
struct PStage {
  uint64_t a_src;      // unit u < nA comes from a_src - u * rs_bytes
  uint64_t xa_next;    // device address of the [16 d | 16] block of the NEXT stage
                       // of the (cyclic) sequence (its first j-block)
  uint32_t xb_next;    // ... bytes from there to the block of its second j-block (0: none)
  uint32_t rs_bytes;   // bytes between consecutive row blocks in Apack
  uint32_t word;       // PW_*
  uint32_t info;       // PI_*: j-block of segment A (factor tables: row [jb] of every axis), how
                       // many j-blocks further segment B's is, GP of the NEXT stage (whose
                       // riders' alpha blocks go with xa_next)
  // Unit u >= nA comes from a_src + b_off - u * rs_bytes with b_off = 2 KB * (jb2 - jb) +
  // nA * rs_bytes (formed on the device: the entry stays ONE aligned 32-byte scalar load --
  // scalar loads count on lgkmcnt like the LDS reads around them, every further piece
  // of an entry couples another wait to them).
  __host__ __device__ uint32_t jb() const { return info & 1023u; }
  __host__ __device__ uint32_t djb() const { return (info >> 10) & 1023u; }
  __host__ __device__ uint32_t g_next() const { return (info >> 20) & 7u; }
};
enum : uint32_t {
  PW_NACT_MASK = 63u,       // active global slots 0 .. nact-1 (1..32)
  PW_CHUNK_END = 1u << 6,   // last j-block of an accumulator chunk: fold
  PW_GP_END = 1u << 7,      // last stage of a GP: row epilogue
  PW_TILE_END = 1u << 8,    // last stage of the tile
  PW_MEAN = 1u << 9,        // stage of the LAST chunk: accumulate alpha . k
  PW_NARROW = 1u << 10,     // global slot 0 holds a narrow row block (k_pack)
  PW_GP_FIRST = 1u << 11,   // first stage of a GP
  PW_LAST_GP = 1u << 15,    // stage of the last GP of the tile
  PW_CHUNK_SHIFT = 16,      // running number of the accumulator chunk (6 bits)
  PW_SHARED = 1u << 22,     // GP with the factor of the GP in front (GpDev::share): its
                            // stages have no active slots, they only evaluate the
                            // covariances for alpha . k; |L^-1 k|^2 is the leader's
  PW_DUO = 1u << 23,        // the stage has a second j-block (segment B)
  PW_NB_SHIFT = 24,         // active global slots of segment B (6 bits)
  PW_G_SHIFT = 12           // GP index (3 bits)
};

// LDS (doubles):  [2][A chunk 64 KB]  [2][2 j-blocks][16 D rows | 16 alpha]  exp table
//                 [4 pairs][2][B operands]  [4 pairs][2][32] pair exchange
//                 [4 pairs] staged Q rows
// The B operands of a stage's SECOND j-block live in the last four units of the chunk
// buffer that holds the stage's L^-1 units (kKb2Off; one 2 KB buffer per pair): they are
// written and read exactly when that buffer's units are (evaluated a stage ahead, while the
// chunk copy fills the units in front of them).
// R > 0 (riders, see kMaxRide): R more alpha blocks behind a training block, R more
// 16-double rows in a pair's exchange buffer.
template <int D, int R = 0>
struct LayP {
  static constexpr int kATile = kPairSlots * kSteps * 64;   // 8192 doubles
  static constexpr int kKb2Off = kDuoUnits * kSteps * 64;   // (inside a chunk buffer)
  static constexpr int kXBlk = kJC * D + kJC * (1 + R);     // one j-block's training block
  static constexpr int kXBuf = 2 * kXBlk;
  static constexpr int kXOff = 2 * kATile;
  static constexpr int kTabOff = kXOff + 2 * kXBuf;
  static constexpr int kKbOff = kTabOff + kExpTabSize;
  static constexpr int kKbBuf = 4 * kKbRow;                 // one B buffer
  static constexpr int kExOff = kKbOff + kPairs * 2 * kKbBuf;
  static constexpr int kExRow = 16 * (2 + R);               // [|L^-1 k|^2 | alpha.k | riders]
  static constexpr int kQOff = kExOff + kPairs * 2 * kExRow;
  static constexpr int kQMaxG = 6;                          // staged Q rows: G <= 6
  static constexpr int kQCap = 16 * 2 * kQMaxG;             // doubles per pair
  static constexpr int kTotal = kQOff + kPairs * kQCap;
  static constexpr size_t bytes() { return size_t(kTotal) * sizeof(double); }
};
// GPs that share the factor of the GP in front of them (GpDev::share: same inputs,
// kernel, noise -- the outputs of a multi-output GP) and RIDE with it: the covariances
// the leader evaluates are theirs as well, so their alpha . k is formed in the
// leader's stages and they have no stages of their own.  Up to kMaxRide per leader
// (what the LDS of a d <= 4 instance has room for); further followers keep stages
// without rows (PW_SHARED).
constexpr int kMaxRide = 2;
static_assert(sizeof(PStage) == 32, "one aligned scalar load per stage");
static_assert(LayP<8>::bytes() <= 160 * 1024, "LDS budget of one workgroup per CU");
static_assert(LayP<1>::kKb2Off + kPairs * LayP<1>::kKbBuf <= LayP<1>::kATile, "second B buffers");
static_assert(LayP<4, kMaxRide>::bytes() <= 160 * 1024, "LDS budget with riders");

struct PairParams {
  const GpDev* gps;
  int G;
  SweepPoints pts;
  ConfOut conf;
  FitnessArgs fit;
  const PStage* stages;   // [nstages] one tile's stage sequence (all GPs)
  int nstages;
  // Work split of the launch (pair_plan).  Workgroup w runs the tiles w, w + W, ..
  // below split_tile0 completely.  With split_parts > 0 the last split_count tiles
  // (a remainder that would keep a few workgroups busy for one more round while
  // the rest of the chip idles) are cut into split_parts runs of whole accumulator
  // chunks: workgroup w < split_count * split_parts takes the stages
  // [split_s0[i], split_s0[i + 1]), i = w % split_parts, of tile
  // split_tile0 + w / split_parts and leaves its PER-LANE chunk sums in split_t /
  // split_m; k_pair_split_finish adds them in the order of the unsplit loop
  // (bit-identical results) and runs the row epilogue.
  int geff;               // GPs in the stage table (1 for the greedy swarm)
  unsigned shared_mask;   // bit g: GP g takes |L^-1 k|^2 from the GP in front (PW_SHARED)
  int nride[SGP_MAX_GPS];             // riders of GP g (the GPs g + 1 .. g + nride[g])
  long long ride_delta[SGP_MAX_GPS];  // rider g: bytes from its leader's XA to its own
  int split_tile0, split_parts, split_count;
  int split_s0[9];
  int nchunks;            // chunks in the stage table (all GPs)
  int chunk_off[SGP_MAX_GPS + 1];   // first chunk number of every GP
  double* split_t;        // [split_count][nchunks][512]
  double* split_m;        // [split_count][geff][512]
  int split_partial0;     // first slot of the finish kernel in ConfOut::partial
  SepLaunch sep;          // tensor grid + factor tables (instances with SEP > 0)
#ifdef SGP_INSTRUMENT
  int ablate;
#endif
#ifdef PGP_STAMPS
  unsigned long long* stamps;   // [blocks][8 waves][8] cycles per phase (debug build)
#endif
};

// Per-phase cycle stamps of the stage loop (-DPGP_STAMPS, scripts/dev): s_memtime at
// the phase boundaries, summed per wave.  Perturbs the run (every stamp drains the
// LDS / scalar-load counter); for attribution only.
#ifdef PGP_STAMPS
#define PGP_STAMP(i)                                                   \
  do {                                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
    stamp_acc[i] += (unsigned int)(t_ - stamp_prev);                   \
    stamp_prev = t_;                                                   \
  } while (0)
#else
#define PGP_STAMP(i) do {} while (0)
#endif

// Timing experiments (results are wrong with any bit set).  -DPGP_CT_ABL=mask removes the
// parts at COMPILE time -- no run-time test next to what is being measured (the run-time
// mask of -DSGP_INSTRUMENT costs the paired kernel 15 %: profiles/r05/experiments.txt) --
// bits: 1 stage barrier, 2 LDS-DMA, 4 covariance evaluation, 8 MFMAs, 16 stores,
// 32 row epilogue, 64 A-operand reads of the slots, 128 B-operand reads.
#if defined(PGP_CT_ABL)
#define PGP_ABL(mask) (((PGP_CT_ABL) & (mask)) != 0)
#elif defined(SGP_INSTRUMENT)
#define PGP_ABL(mask) (p.ablate & (mask))
#else
#define PGP_ABL(mask) false
#endif

typedef const __attribute__((address_space(4))) PStage* pstage_ptr_t;
typedef const __attribute__((address_space(4))) GpDev* gpdev_cptr_t;

template <bool RIDE = false, bool SEP = false>
__device__ __forceinline__ PStage load_pstage(pstage_ptr_t t, int i) {
  PStage e;      // member-wise: scalar loads (only what the instance reads stays live)
  e.a_src = t[i].a_src;
  e.xa_next = t[i].xa_next;
  e.xb_next = t[i].xb_next;
  e.rs_bytes = t[i].rs_bytes;
  e.word = t[i].word;
  e.info = t[i].info;
  return e;
}

// (dma_2k / dma_1k / wait_dma / lds_addr_of: sweep_shared.h)

// [16 D training rows | 16 alpha] of one j-block: 128 D + 128 bytes, one wave.
template <int D>
__device__ __forceinline__ void xa_dma(uint64_t src, uint32_t dst, int lane,
                                       uint32_t voff) {
  constexpr int kLanes = 8 * D + 8;              // 16 bytes each
  if (lane < (kLanes < 64 ? kLanes : 64)) dma_1k(src, dst, voff);
  if (kLanes > 64) {
    if (lane < kLanes - 64) dma_1k(src + 1024, dst + 1024, voff);
  }
}

// ... and the alpha blocks (16 doubles each) of the riders of GP g behind it: the same
// j-block of every rider's own [16 d | 16 alpha] array.
template <int D>
__device__ __forceinline__ void rider_dma(const int (&nride)[SGP_MAX_GPS],
                                          const long long (&delta)[SGP_MAX_GPS], int g,
                                          uint64_t xa_block, uint32_t dst, int lane,
                                          uint32_t voff) {
  const int nr = nride[g];
  for (int f = 0; f < nr; ++f) {        // (wave-uniform)
    const uint64_t src = xa_block + uint64_t(delta[g + 1 + f]) + 128u * D;
    if (lane < 8) dma_1k(src, dst + uint32_t(16 * D + 16 + 16 * f) * 8u, voff);
  }
}

// The share of one wave of half 0 in the copy of the next stage's L^-1 units (LDS image
// [unit][k-step][lane], 2 KB per unit), issued piecewise between the slots of the running
// stage: wave w of the half copies the units w, w + 4, ..; every hook of the slot sequence
// sends the next one.
struct DmaPlan {
  uint64_t src;      // unit u < na: src - u * rs; u >= na: src + b_off - u * rs
  uint32_t b_off;
  uint32_t rs;
  uint32_t dst0;     // LDS byte address of unit 0 of the buffer being filled
  uint32_t voff;
  int u;             // the wave's next unit; >= ntot: nothing left (or the copy is off)
  int na, ntot;      // units of segment A, of the stage
  bool no_a;         // (timing experiments: the slots skip their A-operand reads)
};
__device__ __forceinline__ void dma_next(DmaPlan& d) {
  if (d.u < d.ntot) {
    const uint64_t off = uint64_t(uint32_t(d.u)) * d.rs - (d.u < d.na ? 0u : d.b_off);
    dma_2k(d.src - off, d.dst0 + uint32_t(d.u) * 2048u, d.voff);
    d.u += 4;
  }
}

// ---- matrix part (operand maps: see sweep.hip) ------------------------------------
// Local slot S of half h is unit slot_unit(h, S) of the segment (below); aT points at unit 0
// (+ lane).
// kGroups > 0: one group of the wave's LDS-DMA share goes out after every
// (16 / kGroups)-th slot.
//
// ASM_MFMA: the matrix instructions as inline asm with tied accumulators (see
// sweep_shared.h).  The compiler cannot pad hazards around instructions it does not
// see, and in instances that SPILL it stores accumulators right behind their last
// MFMA (needs 9 wait states, gets 0 -- scripts/dev/check_mfma_hazards.py finds
// such code).  Until round 4 the instances for d >= 6 spilled and used the builtin
// instead (a few register copies at the joins of the slot sequence); since no
// instance has scratch any more (d >= 7 without registers for the raw rows, kLeanX)
// all of them take the asm path, and the scanner runs over every one of them.
// Which unit of a segment is local slot S of half H: alternating, the halves have the same
// number of slots per stage (+-1).  (Half 0 -- which enters its matrix phase first -- with
// one slot more per stage, units 0, 1, 3, 5, ..: +0.7 % at config 3, profiles/r06/experiments.txt.)
__host__ __device__ constexpr int slot_unit(int H, int S) { return 2 * S + H; }
__device__ __forceinline__ int slots_of(int H, int n) { return (n - H + 1) >> 1; }

// One accumulator slot: the A operands of the NEXT slot are requested first, then the 16
// (narrow: 4) matrix instructions, then -- half 0 -- a group of the chunk copy.
template <int S, bool NARROW_OK, int kGroups, bool ASM_MFMA>
__device__ __forceinline__ void pair_slot_body(bool narrow0, double (&acc)[kWaveSlots][4],
                                               double& accx, const double* aT,
                                               const double (&kb)[4][4], const double (&kvn)[4],
                                               double (&cur)[4], double (&nxt)[4],
                                               DmaPlan& dma) {
  if (S + 1 < kWaveSlots) {
#if defined(SGP_INSTRUMENT) || defined(PGP_CT_ABL)
    if (dma.no_a) {
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt[q] = cur[q];
    } else
#endif
    {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        nxt[q] = aT[(slot_unit(kGroups > 0 ? 0 : 1, S + 1) * kSteps + q) * 64];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // The matrix instructions are inline asm: the compiler's hazard recogniser does not
  // see them.  A VALU write of an operand needs two wait states before an MFMA may read
  // the register: the wave's FIRST slot -- behind the evaluation and the operand fetch --
  // opens with s_nop 1.  Between two slots there are LDS reads (s_waitcnt), the chunk
  // copy and a branch, no VALU instruction: the guard in front of every slot (until
  // round 5) cost 0.9-1.2 % of the kernel (profiles/r05/experiments.txt, section 8);
  // scripts/dev/check_mfma_hazards.py (tests/test_abi.py) scans every instance's ISA for
  // a register copy the compiler might place there after all.
  // One asm statement per MFMA, on purpose: an accumulator is the addend again four
  // instructions later and needs 4 wait states -- three MFMAs and ONE more instruction.
  // The compiler supplies it (its hazard recogniser treats every inline asm as a
  // destination-forwarding producer and does not count asm statements as wait states:
  // an s_nop 0, or the slot's s_cmp, lands in front of MFMAs 5, 9 and 13), the scanner
  // checks it.  All 16 in one statement: no nops, two register copies per slot instead,
  // +5.8 % (profiles/r05/experiments.txt, section 9).
  if (!ASM_MFMA) {
    if (NARROW_OK && S == 0 && narrow0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        accx = __builtin_amdgcn_mfma_f64_4x4x4f64(cur[q], kvn[q], accx, 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[S][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(cur[q], kb[m][q], acc[S][m], 0, 0, 0);
      }
    }
  } else if (NARROW_OK && S == 0 && narrow0) {
    // four DEPENDENT MFMAs on one accumulator (4 wait states by hand)
    asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 4"
                 : "+v"(accx) : "v"(cur[0]), "v"(kvn[0]));
#pragma unroll
    for (int q = 1; q < 4; ++q)
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 4"
                   : "+v"(accx) : "v"(cur[q]), "v"(kvn[q]));
  } else {
    if (S == 0)
      asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0"
                   : "+v"(acc[S][0]) : "v"(cur[0]), "v"(kb[0][0]));
    else
      mfma_acc(acc[S][0], cur[0], kb[0][0]);
#pragma unroll
    for (int m = 1; m < 4; ++m) mfma_acc(acc[S][m], cur[0], kb[m][0]);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
#pragma unroll
      for (int m = 0; m < 4; ++m) mfma_acc(acc[S][m], cur[q], kb[m][q]);
    }
  }
  if (S + 1 < kWaveSlots)
    asm volatile("" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]));
  if constexpr (kGroups > 0) {
    constexpr int kEvery = kWaveSlots / kGroups;
    if (S % kEvery == 0) dma_next(dma);
  }
}

// (One compare-and-branch per slot.  Branching once per TWO slots -- two copies of a slot
// body on different paths -- makes the register allocator duplicate the tied accumulators:
// 430-560 B of scratch in every instance, as with every other second control-flow shape
// around them; profiles/r05/experiments.txt, section 8.)
template <int S, bool NARROW_OK, int kGroups, bool ASM_MFMA, int kMax = kWaveSlots>
__device__ __forceinline__ void pair_slots(int nw, bool narrow0,
                                           double (&acc)[kWaveSlots][4], double& accx,
                                           const double* aT,
                                           const double (&kb)[4][4],
                                           const double (&kvn)[4],
                                           double (&cur)[4], double (&nxt)[4],
                                           DmaPlan& dma) {
  if constexpr (S < kMax) {
    if (S < nw) {
      pair_slot_body<S, NARROW_OK, kGroups, ASM_MFMA>(narrow0, acc, accx, aT, kb, kvn, cur, nxt, dma);
      pair_slots<S + 1, false, kGroups, ASM_MFMA, kMax>(nw, false, acc, accx, aT, kb, kvn, nxt, cur, dma);
    }
  }
}

// The lane number, formed where it is used: addresses of the rare paths (the per-lane sums a
// run of a remainder tile leaves in HBM) are loop invariants the compiler would otherwise
// keep in registers -- 64-bit pairs -- across the whole stage loop, next to 128 accumulators.
__device__ __forceinline__ uint32_t cold_lane() {
  uint32_t l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// per-tile state of the row epilogue (only the finishing wave of a pair has it)
struct RowState {
  bool safe = true;
  double l0 = 0.0, values = 0.0, interest = 1.0, total_pen = 0.0, lower = 0.0;
  double lmax = -INFINITY;   // max l0 over the safe rows this wave has seen
};

// One GP's posterior at the wave's 16 rows -> confidence interval / fitness
// shaping; at the end of a tile the rows leave.  `mu`, `var` are complete
// (both halves of the pair); lane l works on row l & 15.
template <int D, int MODE>
__device__ __forceinline__ void row_epilogue(const PairParams& p, RowState& rs,
                                             uint32_t w, int tile, int pr, int lane,
                                             double mu, double var, double* qst) {
  // (no contraction: mu -+ beta sd must round the same way in every kernel this is
  // inlined into -- the sweep and k_pair_split_finish -- and as in the reference,
  // which multiplies, then adds)
#pragma clang fp contract(off)
  typedef LayP<D> L;
  constexpr bool conf = MODE == MODE_CONF;
  const double sd = sqrt(var);
  const int g = int(w >> PW_G_SHIFT) & 7;
  const int64_t row = int64_t(tile) * kTileRows + pr * 16 + (lane & 15);
  const bool writer = (row < p.pts.N) && (lane < 16);
  const int st = p.fit.swarm_type;
  if (conf) {
    // update_confidence_intervals + compute_safe_set (gp_opt.py:453-481)
    const double lo = mu - p.conf.beta * sd;
    const double up = mu + p.conf.beta * sd;
    if (g == 0) rs.l0 = lo;
    rs.safe = rs.safe && (lo > p.conf.fmin[g]);
    if (writer && !PGP_ABL(16)) {
      __builtin_nontemporal_store(mu, p.conf.mean + int64_t(g) * p.pts.N + row);
      __builtin_nontemporal_store(var, p.conf.var + int64_t(g) * p.pts.N + row);
    }
    // Q row = [l0, u0, l1, u1, ...] (gp_opt.py:375): collected in LDS, written
    // as ONE contiguous block per wave at the end of the tile
    if (p.conf.Q && lane < 16 && !PGP_ABL(16)) {
      if (p.G <= L::kQMaxG)
        *reinterpret_cast<double2_t*>(qst + (lane * p.G + g) * 2) = double2_t{lo, up};
      else if (writer)
        *reinterpret_cast<double2_t*>(p.conf.Q + (row * p.G + g) * 2) =
            double2_t{lo, up};
    }
  } else {
    // SafeOptSwarm._compute_particle_fitness, gp_opt.py:925-1013
    const FitnessArgs& f = p.fit;
    rs.lower = mu - f.beta * sd;
    if (g == 0) {
      rs.values = sd / f.scaling[0];
      if (st == SGP_SWARM_EXPANDERS) rs.interest = double(p.G);
      if (st == SGP_SWARM_MAXIMIZERS) {
        const double upper = mu + f.beta * sd;
        const double z = 10.0 * (upper - f.best_lower_bound) / f.scaling[0];
        rs.interest = 1.0 / (1.0 + exp(-z));  // scipy.special.expit
      }
    } else {
      rs.values = fmax(rs.values, sd / f.scaling[g]);
    }
    if (f.fmin[g] != -INFINITY) {
      double slack = rs.lower - f.fmin[g];
      rs.safe = rs.safe && (slack >= 0.0);
      if (st != SGP_SWARM_SAFE_SET) {
        slack = slack / f.scaling[g];
        rs.total_pen += swarm_penalty(slack);
        if (st == SGP_SWARM_EXPANDERS) {
          // scipy.stats.norm.pdf(slack, scale=0.2)
          const double z = slack / 0.2;
          rs.interest *= exp(-0.5 * z * z) / 2.5066282746310002 / 0.2;
        }
      }
    }
  }

  if (w & PW_TILE_END) {
    if (conf) {
      if (p.conf.Q && p.G <= L::kQMaxG && !PGP_ABL(16)) {
        const int64_t row0 = int64_t(tile) * kTileRows + pr * 16;
        const int64_t left = p.pts.N - row0;
        const int nq = (left >= 16 ? 16 : (left > 0 ? int(left) : 0)) * p.G;
        __builtin_amdgcn_wave_barrier();
        double2_t* dst = reinterpret_cast<double2_t*>(p.conf.Q) + row0 * p.G;
        for (int i = lane; i < nq; i += 64)
          __builtin_nontemporal_store(
              *reinterpret_cast<const double2_t*>(qst + 2 * i), dst + i);
        __builtin_amdgcn_wave_barrier();
      }
      if (p.conf.S) {
        if (writer) p.conf.S[row] = rs.safe ? 1 : 0;
        rs.lmax = fmax(rs.lmax, (writer && rs.safe) ? rs.l0 : -INFINITY);
      }
    } else if (writer) {
      double out;
      bool ok = rs.safe;
      if (st == SGP_SWARM_GREEDY) {
        out = rs.lower;
        ok = true;
      } else if (st == SGP_SWARM_SAFE_SET) {
        out = rs.lower;
      } else {
        out = (rs.values + rs.total_pen) * rs.interest;
      }
      p.fit.values[row] = out;
      p.fit.safe[row] = ok ? 1 : 0;
    }
    rs.safe = true;
    rs.l0 = rs.values = rs.total_pen = rs.lower = 0.0;
    rs.interest = 1.0;
  }
}

// The persistent stage loop of one wave; H = its half of the pair (compile time: half 0
// copies the A chunks and owns the narrow slot, half 1 finishes the pair's rows),
// R = riders per leader the instance can carry (0 or kMaxRide).
// SEP > 0: the rows are a tensor grid with SEP axes and every kernel a product of RBF
// parts (SepLaunch, sweep.hip): a covariance is the product of SEP table entries -- one
// 16-byte load per axis, lane and stage, requested a stage ahead -- instead of ~20 fp64
// instructions per value.  (Instantiated with D = 1: the rows themselves are not read.)
template <int D, int MODE, bool SINGLE, int H, int R, int SEP>
__device__ __forceinline__ void pair_loop(const PairParams& p, double* lds,
                                          const int lane, const int wave) {
  typedef LayP<D, R> L;
  constexpr bool conf = MODE == MODE_CONF;
  // (32 more live registers across the evaluation: instances that would spill for
  // it -- d >= 6, product kernels -- do without)
  // (instances with riders: up to d = 3 -- at d = 4 the fetch spills -- and without the row
  // prefetch below; config 3 with the shared factor 4.93 -> 4.87 ms)
  constexpr int kOpsEarly =
      (SINGLE && D <= 4 && (R == 0 || D <= 3)) ? 2 : 0;
  const int pr = wave & 3;
  const int k4 = lane >> 4, c16 = lane & 15;
  const double* tab = lds + L::kTabOff;
  const pstage_ptr_t stages = (pstage_ptr_t)(p.stages);
  const gpdev_cptr_t gpc = (gpdev_cptr_t)(p.gps);
  const int nstages = p.nstages;
  const int ntiles = int((p.pts.N + kTileRows - 1) / kTileRows);
  const int tstep = int(gridDim.x);
  // the work items of this workgroup: its whole tiles, then (split launches) one
  // run of chunks of a remainder tile.  (What describes them is re-read from the
  // kernel arguments where it is needed -- item set-up, chunk / GP ends -- rather
  // than kept in scalar registers across the stage loop.)
  int tile = int(blockIdx.x);          // tile of the stage being multiplied
  int left = 0;                        // stages of the current item still to go

  double* kbp = lds + L::kKbOff + pr * (2 * L::kKbBuf);   // the pair's B buffers
  double* exch = lds + L::kExOff + pr * (2 * L::kExRow);   // ... exchange [2][kExRow]
  double* qst = lds + L::kQOff + pr * L::kQCap;            // ... staged Q rows
  const uint32_t lds_a = lds_addr_of(lds);
  const uint32_t lds_xa = lds_addr_of(lds + L::kXOff);
  const uint32_t voff = uint32_t(lane) * 16u;

  // candidate rows of the tile being EVALUATED (one stage ahead of the
  // multiplication) and, prefetched, of the tile after it
  auto load_x = [&](int t, double (&xo)[D]) {
    int64_t r = int64_t(t) * kTileRows + pr * 16 + c16;
    r = r < p.pts.N ? r : p.pts.N - 1;
#pragma unroll
    for (int k = 0; k < D; ++k)
      xo[k] = __builtin_nontemporal_load(p.pts.base + r * p.pts.stride_row +
                                         k * p.pts.stride_col);
  };
  // x_raw: raw rows of the tile being evaluated -- until the scaled rows of its
  // LAST GP are formed, from then on already the rows of the next tile (the load
  // has a whole GP's stages to arrive)
  // (d >= 7: no registers for the raw row -- it is read, an L2 hit, where a GP's scaled
  // row is formed)
  constexpr bool kLeanX = D >= 7;
  double x_raw[kLeanX ? 1 : D], xs_e[D];
  int tile_e = tile;
  KernFast<D> kf;
  // SEP: byte offsets of this lane's row (c16) and training points (8 H + k4 and 4
  // further: positions 4 k4 + 2 H, + 1 of a block of 16) in the tables, for the tile of
  // the stage whose factors are being REQUESTED (tile_f); axis a's index =
  // (global row / stride_a) % count_a with stride_a = count_0 .. count_{a-1}
  constexpr int kAx = SEP > 0 ? SEP : 1;
  uint32_t soff[kAx];
  int tile_f = tile;
  auto sep_offsets = [&](int t) {
    int64_t r = int64_t(t) * kTileRows + pr * 16 + c16;
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
      soff[a] = idx * 128u + uint32_t(k4) * 32u + uint32_t(H) * 16u;
    }
  };
  double2_t efn[kAx], efn2[kAx];     // (second j-block of a stage: efn2)
  const char* sep_tab[kAx];
  uint32_t sep_pitch[kAx];
  int sep_g = -1;
  auto sep_fetch = [&](const PStage& e) {
    const int g = int(e.word >> PW_G_SHIFT) & 7;
    if (g != sep_g) {
      sep_g = g;
#pragma unroll
      for (int a = 0; a < kAx; ++a) {
        sep_tab[a] = reinterpret_cast<const char*>(uniform_ptr(p.sep.tab[g][a]));
        sep_pitch[a] = __builtin_amdgcn_readfirstlane(p.sep.count[a] * 128u);
      }
    }
    // (explicitly GLOBAL loads: a flat load also counts on lgkmcnt)
    typedef const __attribute__((address_space(1))) double2_t* gvec_t;
#pragma unroll
    for (int a = 0; a < kAx; ++a) {
      const char* src = sep_tab[a] + e.jb() * sep_pitch[a];    // (<= 256 MB per GP: sep_launch)
      efn[a] = *(gvec_t)(reinterpret_cast<const double2_t*>(src + soff[a]));
    }
    if (e.word & PW_DUO) {
#pragma unroll
      for (int a = 0; a < kAx; ++a) {
        const char* src = sep_tab[a] + (e.jb() + e.djb()) * sep_pitch[a];
        efn2[a] = *(gvec_t)(reinterpret_cast<const double2_t*>(src + soff[a]));
      }
    }
  };

  // covariances of one stage: this wave's half (training points 8 H .. 8 H + 7 of
  // the j-block) -> the pair's B buffer, [k][q pair][point][2]
  double mean = 0.0;
  // riders of the GP being evaluated (R > 0): their alpha . k, formed from the
  // leader's covariances
  double mean_r[R > 0 ? R : 1];
#pragma unroll
  for (int f = 0; f < (R > 0 ? R : 1); ++f) mean_r[f] = 0.0;
  int nr_e = 0;
  // the wave's two training rows (8 H + k4 and 4 further) and alpha entries of a
  // staged j-block: read FIRST in a stage, so that the evaluation does not queue
  // behind the operand reads of the matrix phase
  struct Rows {
    double y[2 * D];
  };
  auto load_rows = [&](const double* xa, Rows& r) {
    const double* ys = xa + (8 * H + k4) * D;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < D; ++i) r.y[q * D + i] = ys[q * 4 * D + i];
  };
  // (only where the 2 D + 2 registers are there for it; elsewhere the evaluation
  // reads its rows itself)
  constexpr bool kRowsFirst = kOpsEarly != 0 && SINGLE && D <= 2 && R == 0;
  // One j-block of a stage: J = 0 the first (its rows may have been fetched in front of
  // the evaluation: Rows), J = 1 the second of a merged stage.
  auto eval_block = [&](uint32_t w1, auto jtag, const Rows& r, const double* xa, double* kbw) {
    constexpr int J = decltype(jtag)::value;
    double kv[2];
    if constexpr (SEP > 0) {
      const double2_t (&ef)[kAx] = J == 0 ? efn : efn2;
      kv[0] = ef[0].x;
      kv[1] = ef[0].y;
#pragma unroll
      for (int a = 1; a < kAx; ++a) {
        kv[0] *= ef[a].x;
        kv[1] *= ef[a].y;
      }
    } else if (!PGP_ABL(4)) {
      if (kRowsFirst && J == 0)
        kf.template manyn_t<2, SINGLE>(xs_e, r.y, D, tab, kv);
      else
        kf.template manyn_t<2, SINGLE>(xs_e, xa + (8 * H + k4) * D, 4 * D, tab, kv);
    } else {
      kv[0] = xs_e[0];
      kv[1] = xs_e[0] + 1.0;
    }
    if (w1 & PW_MEAN) {
      // (alpha is read here, not with the rows: four registers the merged stages need)
      {
      const double* al = xa + kJC * D + 8 * H + k4;
      mean = fma(al[0], kv[0], mean);
      mean = fma(al[4], kv[1], mean);
      }
    }
    if (R > 0 && (w1 & PW_MEAN)) {
#pragma unroll
      for (int f = 0; f < R; ++f) {
        if (f < nr_e) {
          const double* al = xa + kJC * D + kJC * (1 + f) + 8 * H + k4;
          mean_r[f] = fma(al[0], kv[0], mean_r[f]);
          mean_r[f] = fma(al[4], kv[1], mean_r[f]);
        }
      }
    }
    // (odd k-rows: the halves of the q pair's 32 doubles swapped -- fetch_ops)
    *reinterpret_cast<double2_t*>(kbw + k4 * kKbRow + H * 32 + ((c16 * 2) ^ ((k4 & 1) * 16))) =
        double2_t{kv[0], kv[1]};
  };
  auto evaluate = [&](uint32_t w1, const Rows& r, const double* xa, double* kbw, double* kb2w) {
    if (SEP == 0 && __builtin_expect((w1 & PW_GP_FIRST) != 0, 0)) {
      kf.load_const(&p.gps[int(w1 >> PW_G_SHIFT) & 7].kern);
      if (R > 0) nr_e = p.nride[int(w1 >> PW_G_SHIFT) & 7];
      if constexpr (kLeanX) {
        double xr[D];
        load_x(tile_e, xr);
        kf.template prep_t<SINGLE>(xr, xs_e);
        if (w1 & PW_LAST_GP) tile_e += tstep;
      } else {
        kf.template prep_t<SINGLE>(x_raw, xs_e);
        if (w1 & PW_LAST_GP) {
          tile_e += tstep;
          load_x(tile_e, x_raw);
        }
      }
    }
    eval_block(w1, std::integral_constant<int, 0>{}, r, xa, kbw);
    // (no interleaving of the two blocks: four values in flight cost registers the
    // kernel does not have)
    __builtin_amdgcn_sched_barrier(0);
    if (w1 & PW_DUO)
      eval_block(w1, std::integral_constant<int, 1>{}, r, xa + L::kXBlk, kb2w);
  };

  PStage e1{};
  uint32_t wcur = 0;
  int si1 = 0;

  // accumulators
  double ssq_run = 0.0;             // folded squares of the finished chunks (per lane)
  double accx = 0.0;                // narrow slot 0 (H == 0 only)
  double acc[kWaveSlots][4];
#pragma unroll
  for (int b = 0; b < kWaveSlots; ++b)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[b][m] = 0.0;
  // the half whose wave finishes a pair's rows (sums the two partial |L^-1 k|^2 and
  // alpha . k, runs the row epilogue one stage later); the other hands its share over
  constexpr int kFin = 1;
  RowState rs;                       // (H == kFin: the finishing wave)
  double keep_ssq = 0.0, keep_mu = 0.0;
  double keep_mu_r[R > 0 ? R : 1];
#pragma unroll
  for (int f = 0; f < (R > 0 ? R : 1); ++f) keep_mu_r[f] = 0.0;
  uint32_t pend_w = 0;               // GP-end word waiting for its epilogue
  int pend_tile = 0;

  constexpr int kDmaGroups = H == 0 ? 8 : 0;     // (the waves of half 0 copy the A chunks)
  // B operands of a stage (operand (q, m) of lane (k, a, j) = value of training
  // point 4 q + k at row 4 m + j), the plain covariance register for a narrow slot
  // 0, and the A operands of the wave's first slot
  struct Ops {
    double kb[4][4];
    double kvn[4];
    double a0[4];
  };
  auto fetch_ops = [&](const double* abuf, const double* kbr, Ops& o, int part) {
    // part 1: B operands, part 2: the rest, 3: both
    if ((part & 1) && PGP_ABL(128)) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) o.kb[m][q] = 1.0 + m;
    } else if (part & 1) {
      // (row quads m = 0, 1 | 2, 3 sit in the two 16-double halves of a q pair's 32; odd
      // k-rows store them swapped: the k-rows of a 16-lane read group then hit different
      // banks without padding between them)
      const int sw = ((k4 & 1) * 16);
      const double2_t* r0 = reinterpret_cast<const double2_t*>(
          kbr + k4 * kKbRow + (lane & 3) * 2 + sw);
      const double2_t* r1 = reinterpret_cast<const double2_t*>(
          kbr + k4 * kKbRow + (lane & 3) * 2 + 16 - sw);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double2_t* r = m < 2 ? r0 + m * 4 : r1 + (m - 2) * 4;
        const double2_t a = r[0], b = r[16];
        o.kb[m][0] = a.x; o.kb[m][1] = a.y; o.kb[m][2] = b.x; o.kb[m][3] = b.y;
      }
    }
    if (!(part & 2)) return;
    if (H == 0) {
      // (read whether slot 0 is narrow or not: two LDS reads are cheaper than a
      // branch and four register moves next to the matrix instructions)
      const double2_t* rn = reinterpret_cast<const double2_t*>(
          kbr + k4 * kKbRow + ((c16 * 2) ^ ((k4 & 1) * 16)));
      const double2_t a = rn[0], b = rn[16];
      o.kvn[0] = a.x; o.kvn[1] = a.y; o.kvn[2] = b.x; o.kvn[3] = b.y;
    }
    const double* aT = abuf + slot_unit(H, 0) * (kSteps * 64) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) o.a0[q] = aT[q * 64];
  };
  auto multiply = [&](uint32_t w, const double* abuf, const double* kb2r, Ops& o,
                      DmaPlan& dma) {
    // segment A, then (merged stages) segment B through a second, shorter instance of the
    // slot sequence (segment B has at most kDuoUnits / 2 - 1 units).  Two sequences one
    // behind the other keep the tied accumulators where they are (no copies, no scratch:
    // scripts/resource_usage.py); ONE sequence inside a two-trip loop did too, but cost
    // 0.3 ms at config 3 (profiles/r06/experiments.txt, section 1).
    const int na = int(w & PW_NACT_MASK), nb = int(w >> PW_NB_SHIFT) & 63;
    const bool narrow0 = H == 0 && (w & PW_NARROW) != 0;
    {
      const int nw = slots_of(H, na);
      if (nw > 0 && !PGP_ABL(8)) {
        const double* aT = abuf + lane;
        double opsB[4];
        pair_slots<0, H == 0, kDmaGroups, true>(nw, narrow0, acc, accx, aT, o.kb,
                                                    o.kvn, o.a0, opsB, dma);
      }
      const int nw2 = slots_of(H, nb);
      if (nw2 > 0) {
        const double* aseg2 = abuf + na * (kSteps * 64);
        fetch_ops(aseg2, kb2r, o, 3);
        const double* aT = aseg2 + lane;
        double opsB[4];
        pair_slots<0, H == 0, kDmaGroups, true, (kDuoUnits / 2 + 1) / 2>(
            nw2, narrow0, acc, accx, aT, o.kb, o.kvn, o.a0, opsB, dma);
      }
    }
    if constexpr (kDmaGroups > 0) {
      // units that found no hook (the running stage has fewer slots than the next one units)
#pragma unroll 1
      while (dma.u < dma.ntot) dma_next(dma);
    }
    if (__builtin_expect((w & PW_CHUNK_END) != 0, 0)) {
      // squares of the chunk's accumulators, folded at once to this wave's share of
      // |L^-1 k|^2 per row (lane l: row l & 15): one register survives the chunk
      double sq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < kWaveSlots; ++b) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          sq[m] = fma(acc[b][m], acc[b][m], sq[m]);
          acc[b][m] = 0.0;
        }
      }
      // sq[m]: partial sums for column 4m + (lane & 3) over this lane's rows.
      // Transposing fold over the lanes that share (lane & 3): after the xor-4 and
      // xor-8 exchanges each lane holds the quad of its OWN column (lane & 15).
      const bool a0 = (lane & 4) != 0, a1 = (lane & 8) != 0;
      const double v0 = (a0 ? sq[1] : sq[0]) + __shfl_xor(a0 ? sq[0] : sq[1], 4, 64);
      const double v1 = (a0 ? sq[3] : sq[2]) + __shfl_xor(a0 ? sq[2] : sq[3], 4, 64);
      double t = (a1 ? v1 : v0) + __shfl_xor(a1 ? v0 : v1, 8, 64);
      if (H == 0) {
        t = fma(accx, accx, t);     // (narrow slot 0: rows l >> 4, row-of-tile l & 15)
        accx = 0.0;
      }
      if (p.split_parts > 0 && tile >= p.split_tile0)
        p.split_t[(size_t(tile - p.split_tile0) * p.nchunks +
                   (int(w >> PW_CHUNK_SHIFT) & 63)) * 512 + wave * 64 + cold_lane()] = t;
      else
        ssq_run += t;
    }
  };

  // this wave's share of |L^-1 k|^2 at the 16 rows (lane l: row l & 15)
  auto gp_partials = [&](double& ssq) {
    ssq = sum_lane_groups(ssq_run);
    ssq_run = 0.0;
  };

  double ssq_lead = 0.0;             // |L^-1 k|^2 of the last GP with a factor of its own
  auto finish = [&](int par_prev) {
    const double* ex = exch + par_prev * L::kExRow;
    double ssq = keep_ssq + ex[c16];
    if (pend_w & PW_SHARED)
      ssq = ssq_lead;
    else
      ssq_lead = ssq;
    const double mu = keep_mu + ex[16 + c16];
    const int g = int(pend_w >> PW_G_SHIFT) & 7;
    const double kdiag = gpc[g].kern.kdiag;
    const double var = fmax(kdiag - ssq, 1e-15);  // GPy clip
    const int nr = R > 0 ? p.nride[g] : 0;
    // (the tile ends behind the last rider)
    row_epilogue<D, MODE>(p, rs, nr > 0 ? pend_w & ~uint32_t(PW_TILE_END) : pend_w, pend_tile,
                          pr, lane, mu, var, qst);
    if (R > 0) {
#pragma unroll
      for (int f = 0; f < R; ++f) {
        if (f < nr) {
          // a rider: the leader's |L^-1 k|^2, its own alpha . k and prior variance
          const int gf = g + 1 + f;
          const double mu_f = keep_mu_r[f] + ex[16 * (2 + f) + c16];
          const double var_f = fmax(gpc[gf].kern.kdiag - ssq, 1e-15);
          uint32_t wf = (pend_w & ~uint32_t((7u << PW_G_SHIFT) | PW_TILE_END)) |
                        (uint32_t(gf) << PW_G_SHIFT);
          if (f == nr - 1) wf |= pend_w & PW_TILE_END;
          row_epilogue<D, MODE>(p, rs, wf, pend_tile, pr, lane, mu_f, var_f, qst);
        }
      }
    }
    pend_w = 0;
  };

#ifdef PGP_STAMPS
  unsigned int stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  int par = 0;
#pragma unroll 1
  for (int item = 0; item < 2; ++item) {
  int s_lo = 0;                        // first stage of the item
  if (item == 0) {
    const int whole_end = p.split_parts > 0 ? p.split_tile0 : ntiles;
    if (int(blockIdx.x) >= whole_end) continue;
    tile = int(blockIdx.x);
    left = ((whole_end - tile + tstep - 1) / tstep) * nstages;
  } else {
    if (!(p.split_parts > 0 && int(blockIdx.x) < p.split_count * p.split_parts)) break;
    const int part = int(blockIdx.x) % p.split_parts;
    s_lo = p.split_s0[part];
    tile = p.split_tile0 + int(blockIdx.x) / p.split_parts;
    left = p.split_s0[part + 1] - s_lo;
  }
  // ---- start of the item.  The pipeline is primed by an EMPTY stage in front of
  // the first one (no active slots, no flags): its iteration copies the first A
  // chunk and evaluates the first stage's covariances with the very code every
  // other stage runs (one copy of the evaluation in the kernel: a stage must give
  // the same bits wherever an item begins).  Only the training block of the first
  // stage has to be in LDS before it.
  par = 0;
  e1 = load_pstage<(R > 0), (SEP > 0)>(stages, s_lo);
  if (wave == 7) {
    const PStage el = load_pstage<(R > 0), (SEP > 0)>(stages, (s_lo == 0 ? nstages : s_lo) - 1);
    // block(s) of the first stage
    xa_dma<D>(el.xa_next, lds_xa + L::kXBuf * 8, lane, voff);
    if (R > 0)
      rider_dma<D>(p.nride, p.ride_delta, int(el.g_next()), el.xa_next, lds_xa + L::kXBuf * 8,
                   lane, voff);
    if (el.xb_next != 0) {
      xa_dma<D>(el.xa_next + el.xb_next, lds_xa + (L::kXBuf + L::kXBlk) * 8, lane, voff);
      if (R > 0)
        rider_dma<D>(p.nride, p.ride_delta, int(el.g_next()), el.xa_next + el.xb_next,
                     lds_xa + (L::kXBuf + L::kXBlk) * 8, lane, voff);
    }
  }
  wcur = 0;
  ++left;
  si1 = s_lo;
  tile_e = tile;
  if constexpr (SEP > 0) {
    tile_f = tile;
    sep_offsets(tile_f);
    sep_fetch(e1);
  }
  if constexpr (!kLeanX && SEP == 0) load_x(tile_e, x_raw);
  if (SEP == 0 && !(e1.word & PW_GP_FIRST)) {
    // the item begins inside a GP (a run of chunks of a remainder tile)
    kf.load_const(&p.gps[int(e1.word >> PW_G_SHIFT) & 7].kern);
    if (R > 0) nr_e = p.nride[int(e1.word >> PW_G_SHIFT) & 7];
    if constexpr (kLeanX) {
      double xr[D];
      load_x(tile_e, xr);
      kf.template prep_t<SINGLE>(xr, xs_e);
    } else {
      kf.template prep_t<SINGLE>(x_raw, xs_e);
    }
  }
  wait_dma();
  __syncthreads();

#pragma unroll 1
  while (true) {
    const bool more = left > 1;
    const uint32_t wnext = e1.word;

    if (H == kFin && pend_w != 0 && !PGP_ABL(32)) finish(par ^ 1);
    PGP_STAMP(0);   // deferred row epilogue

    // ---- prefetch: the training block of the stage after the next one (its A chunk
    // goes out piecewise between the slots of half 0, DmaPlan)
    // Which wave copies it: without riders wave 7, at the top of the stage (14.77 vs
    // 14.85 ms at config 3 for a wave of half 0); with riders -- three instructions --
    // wave 0, the half with slack at the barrier, behind the VALU bursts and in front of
    // its first slot (config 3 with the shared factor: 5.77 -> 5.65 ms).
    // (d = 4 with riders: wave 7 again -- the other placement spills there)
    constexpr bool kXaHalf0 = R > 0 && D <= 3;
    // (the training blocks of BOTH j-blocks of a merged stage)
    auto xa_prefetch = [&](const PStage& e, uint32_t dst) {
      xa_dma<D>(e.xa_next, dst, lane, voff);
      if (R > 0) rider_dma<D>(p.nride, p.ride_delta, int(e.g_next()), e.xa_next, dst, lane, voff);
      if (e.xb_next != 0) {
        xa_dma<D>(e.xa_next + e.xb_next, dst + L::kXBlk * 8, lane, voff);
        if (R > 0)
          rider_dma<D>(p.nride, p.ride_delta, int(e.g_next()), e.xa_next + e.xb_next,
                       dst + L::kXBlk * 8, lane, voff);
      }
    };
    if constexpr (R == 0 || !kXaHalf0) {
      if (more && !PGP_ABL(2) && wave == 7)
        xa_prefetch(e1, lds_xa + uint32_t(par) * (L::kXBuf * 8));
    }
    DmaPlan plan{};
    if constexpr (kDmaGroups > 0) {
      plan.src = e1.a_src;
      plan.b_off = e1.djb() * 2048u + uint32_t(wnext & PW_NACT_MASK) * e1.rs_bytes;
      plan.rs = e1.rs_bytes;
      plan.dst0 = lds_a + uint32_t(par ^ 1) * (L::kATile * 8);
      plan.voff = voff;
      plan.na = int(wnext & PW_NACT_MASK);
      plan.ntot = plan.na + (int(wnext >> PW_NB_SHIFT) & 63);
      if (!(more && !PGP_ABL(2))) plan.ntot = 0;
      plan.u = wave & 3;           // first unit of this wave in the copy
    }
    plan.no_a = PGP_ABL(64);
    PGP_STAMP(1);   // LDS-DMA issue
    int si2 = si1 + 1;
    if (si2 == nstages) si2 = 0;    // (a run of chunks ends before it would wrap)
    // (the table entry after the next one: a scalar load in flight here makes the
    // evaluation below wait for the B operand reads as well -- which is FASTER than
    // letting it start early, the waves of a SIMD then run their VALU bursts
    // together; experiments.txt section 10)
    PStage e2 = e1;
    if (left > 2) e2 = load_pstage<(R > 0), (SEP > 0)>(stages, si2);

    const double* abuf = lds + par * L::kATile;
    const double* kbr = kbp + par * L::kKbBuf;
    double* kbw = kbp + (par ^ 1) * L::kKbBuf;
    // (second j-block: in the tail of the chunk buffer of the stage it belongs to)
    const double* kb2r = abuf + L::kKb2Off + pr * L::kKbBuf;
    double* kb2w = lds + (par ^ 1) * L::kATile + L::kKb2Off + pr * L::kKbBuf;
    const double* xa = lds + L::kXOff + (par ^ 1) * L::kXBuf;

    Ops ops;
    Rows rows;
    // both halves evaluate first (the barrier starts the VALU bursts of a SIMD's two
    // waves together), then multiply; where the registers are there, the training rows
    // and the B operands of the matrix phase are fetched in front of the evaluation
    if (kRowsFirst && more) load_rows(xa, rows);
    if (kOpsEarly) fetch_ops(abuf, kbr, ops, 1);
    if (__builtin_expect((wcur & PW_GP_END) != 0, 0)) {
      // alpha . k of the GP that ends here (the evaluation below may already belong
      // to the next one): half 0 hands its share over, half 1 keeps it
      const int g_end = int(wcur >> PW_G_SHIFT) & 7;
      const int nr = R > 0 ? p.nride[g_end] : 0;
      if (p.split_parts > 0 && tile >= p.split_tile0) {
        // (a run of a remainder tile: per lane, summed by k_pair_split_finish)
        double* sm = p.split_m + (size_t(tile - p.split_tile0) * p.geff + g_end) * 512 +
                     wave * 64 + cold_lane();
        sm[0] = mean;
        if (R > 0) {
#pragma unroll
          for (int f = 0; f < R; ++f)
            if (f < nr) sm[size_t(1 + f) * 512] = mean_r[f];
        }
      } else {
        const double mu = sum_lane_groups(mean);
        if (H != kFin) {
          if (lane < 16) exch[par * L::kExRow + 16 + lane] = mu;
        } else {
          keep_mu = mu;
        }
        if (R > 0) {
#pragma unroll
          for (int f = 0; f < R; ++f) {
            if (f < nr) {
              const double mu_f = sum_lane_groups(mean_r[f]);
              if (H != kFin) {
                if (lane < 16) exch[par * L::kExRow + 16 * (2 + f) + lane] = mu_f;
              } else {
                keep_mu_r[f] = mu_f;
              }
            }
          }
        }
      }
      mean = 0.0;
      if (R > 0) {
#pragma unroll
        for (int f = 0; f < R; ++f) mean_r[f] = 0.0;
      }
    }
    if (more) evaluate(wnext, rows, xa, kbw, kb2w);
    if constexpr (SEP > 0) {
      // the factors of the stage after the next one (its entry, e2, was requested
      // above: it has arrived under the evaluation); they have the matrix phase to come
      if (left > 2) {
        __builtin_amdgcn_sched_barrier(0);
        if (si2 == 0) {
          tile_f += tstep;
          sep_offsets(tile_f);
        }
        sep_fetch(e2);
      }
    }
    PGP_STAMP(3);     // covariance evaluation
    fetch_ops(abuf, kbr, ops, kOpsEarly ? 2 : 3);
    if constexpr (kXaHalf0 && H == 0) {
      if (more && !PGP_ABL(2) && wave == 0)
        xa_prefetch(e1, lds_xa + uint32_t(par) * (L::kXBuf * 8));
    }
    multiply(wcur, abuf, kb2r, ops, plan);
    PGP_STAMP(2);     // matrix phase (operand reads, slots, chunk fold)

    if (__builtin_expect((wcur & PW_GP_END) != 0, 0)) {
      if (!(p.split_parts > 0 && tile >= p.split_tile0)) {
        double ssq;
        gp_partials(ssq);
        if (H != kFin) {
          if (lane < 16) exch[par * L::kExRow + lane] = ssq;
        } else {
          keep_ssq = ssq;
          pend_w = wcur;
          pend_tile = tile;
        }
      }
      if (wcur & PW_TILE_END) tile += tstep;
    }

    PGP_STAMP(4);     // GP-end partials
    if (!more) break;
    wait_dma();
    PGP_STAMP(5);     // wait for this wave's LDS-DMA
    if (!PGP_ABL(1)) __syncthreads();
    PGP_STAMP(6);     // barrier
    par ^= 1;
    wcur = wnext;
    e1 = e2;
    si1 = si2;
    --left;
  }
  // (every wave is done with the buffers before the next item refills them)
  __syncthreads();
  if (H == kFin && pend_w != 0 && !PGP_ABL(32)) finish(par);
  }   // items
#ifdef PGP_STAMPS
  if (lane == 0) {
    unsigned long long* o = p.stamps + (size_t(blockIdx.x) * 8 + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = stamp_acc[i];
  }
#endif
  if (H == kFin) {
    if (conf && p.conf.S) {
      const double m = wave_max(rs.lmax);
      if (lane == 0) p.conf.partial[int(blockIdx.x) * kPairs + pr] = m;
    }
  }
}

template <int D, int MODE, bool SINGLE, int R = 0, int SEP = 0>
__global__ __launch_bounds__(512, 1) void k_sweep_pair(PairParams p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  exp_tab_init(lds + LayP<D, R>::kTabOff);   // visible after the first barrier
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4)
    pair_loop<D, MODE, SINGLE, 0, R, SEP>(p, lds, lane, wave);
  else
    pair_loop<D, MODE, SINGLE, 1, R, SEP>(p, lds, lane, wave);
}

// Remainder tiles that were cut into runs of chunks (PairParams::split_*): the
// per-lane chunk sums are added in the order of the unsplit loop -- chunk by chunk
// into one register, then over the lane groups, then half 1 + half 0 -- so that a
// row's posterior does not depend on whether its tile was split (same bits), and
// the row epilogue runs exactly as in k_sweep_pair.  One 512-thread workgroup per
// tile, wave / lane = the wave / lane of the sweep.
template <int MODE>
__global__ __launch_bounds__(512) void k_pair_split_finish(PairParams p) {
  constexpr bool conf = MODE == MODE_CONF;
  __shared__ double sh_ex[kPairs][32];
  __shared__ __attribute__((aligned(16))) double sh_q[kPairs][LayP<1>::kQCap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pr = wave & 3;
  const int half = wave >> 2;
  const int c16 = lane & 15;
  const int pt = int(blockIdx.x);
  const int tile = p.split_tile0 + pt;
  RowState rs;
  double ssq_lead = 0.0;
  for (int g = 0; g < p.geff; ++g) {
    double ssq_run = 0.0;
    for (int c = p.chunk_off[g]; c < p.chunk_off[g + 1]; ++c)
      ssq_run += p.split_t[(size_t(pt) * p.nchunks + c) * 512 + tid];
    const double ssq = sum_lane_groups(ssq_run);
    const double mu = sum_lane_groups(p.split_m[(size_t(pt) * p.geff + g) * 512 + tid]);
    if (half == 0 && lane < 16) {
      sh_ex[pr][lane] = ssq;
      sh_ex[pr][16 + lane] = mu;
    }
    __syncthreads();
    if (half == 1) {
      double ssq_t = ssq + sh_ex[pr][c16];
      if (p.shared_mask & (1u << g))
        ssq_t = ssq_lead;
      else
        ssq_lead = ssq_t;
      const double mu_t = mu + sh_ex[pr][16 + c16];
      const double var = fmax(p.gps[g].kern.kdiag - ssq_t, 1e-15);  // GPy clip
      uint32_t w = (uint32_t(g) << PW_G_SHIFT) | PW_GP_END;
      if (g == p.geff - 1) w |= PW_TILE_END;
      row_epilogue<1, MODE>(p, rs, w, tile, pr, lane, mu_t, var, sh_q[pr]);
    }
    __syncthreads();
  }
  if (half == 1 && conf && p.conf.S) {
    const double m = wave_max(rs.lmax);
    if (lane == 0) p.conf.partial[p.split_partial0 + pt * kPairs + pr] = m;
  }
}

// ---- host side ------------------------------------------------------------------
// The stage sequence of one tile: for every GP, for every chunk of 32 row blocks
// of L^-1, the j-blocks 0 .. bend-1.  Entries hold absolute addresses, so the
// table is rebuilt when a block count OR a buffer address changes (buffers are
// sized for the pitch of L^-1: one-row appends keep their addresses).
// sep: the launch reads factor tables -- the training block of a stage is its 16 alpha
// only (the D = 1 instance copies 256 bytes: the address is that of alpha - 128).
//
// Merged stages (round 6): inside a chunk the j-blocks are taken from both ends -- the
// widest one left with the narrowest one left while their active slots together fit
// kDuoUnits, alone otherwise.  32 row blocks: j-blocks 0 .. 4 alone (32 .. 28 slots), then
// (5, 31), (6, 30), .. (17, 19) with 28 slots each, then 18 (14): 19 stages instead of 32.
// An accumulator still meets only j-blocks at or below its row block, in a different
// order than 0, 1, 2, .. (the sums differ in the last bits from the unmerged schedule,
// which sgp_ctx_set_sweep(.. | 32) / SGP_PAIR_MERGE=0 keep for A/B runs).
bool pair_merge_wanted(const sgp_ctx* ctx) {
  static const bool off = getenv("SGP_PAIR_MERGE") && atoi(getenv("SGP_PAIR_MERGE")) == 0;
  return !off && !(ctx->sweep_choice & 32);
}

int pair_stage_table(sgp_ctx* ctx, const GpDev* gh, int Geff, int d, bool sep, const bool* rides,
                     const PStage** dev, int* nstages) {
  const bool merge = pair_merge_wanted(ctx);
  std::vector<uint64_t> sig(1, uint64_t(Geff));
  sig.push_back(uint64_t(d) | (uint64_t(sep) << 8) | (uint64_t(merge) << 9));
  int last_staged = 0;
  for (int g = 0; g < Geff; ++g)
    if (!rides[g]) last_staged = g;
  for (int g = 0; g < Geff; ++g) {
    sig.push_back(uint64_t(gh[g].nblk));
    sig.push_back(uint64_t(gh[g].narrow) | (uint64_t(gh[g].share >= 0) << 8) |
                  (uint64_t(rides[g]) << 9));
    sig.push_back(reinterpret_cast<uint64_t>(gh[g].Apack));
    sig.push_back(reinterpret_cast<uint64_t>(gh[g].XA));
  }
  if (sig == ctx->pstage_sig && ctx->pstage_tab.p) {
    *dev = static_cast<const PStage*>(ctx->pstage_tab.p);
    *nstages = ctx->pstage_count;
    return 0;
  }
  std::vector<PStage> tab;
  std::vector<uint64_t> xa, xb;   // training block(s) of every stage (xb: 0 = no second one)
  std::vector<uint32_t> gof;      // GP of every stage
  std::vector<int> chunk_start;
  const uint64_t xa_block = uint64_t(16 * d + 16) * sizeof(double);
  const uint64_t xa_skip = sep ? uint64_t(16 * d - 16) * sizeof(double) : 0;
  for (int g = 0; g < Geff; ++g) {
    const int nblk = gh[g].nblk, nsteps = gh[g].n_pad / 4;
    const int nchunks = (nblk + kPairSlots - 1) / kPairSlots;
    const uint64_t apack = reinterpret_cast<uint64_t>(gh[g].Apack);
    const uint64_t xa0 = reinterpret_cast<uint64_t>(gh[g].XA) + xa_skip;
    ctx->pstage_chunk_off[g] = int(chunk_start.size());
    if (rides[g]) continue;       // (its alpha . k is formed in its leader's stages)
    const bool rowless = gh[g].share >= 0;
    // same factor as the GP in front: one "chunk" without rows -- every j-block once,
    // for alpha . k
    for (int c = 0; c < (rowless ? 1 : nchunks); ++c) {
      const uint32_t chunk_id = uint32_t(chunk_start.size());
      chunk_start.push_back(int(tab.size()));
      const bool last_chunk = rowless || c == nchunks - 1;
      const int b0 = c * kPairSlots, nib = std::min(kPairSlots, nblk - b0),
                bend = rowless ? nblk : b0 + nib;
      auto active = [&](int jb) { return rowless ? 0 : std::min(nib, bend - jb); };
      auto src_of = [&](int jb) { return apack + (uint64_t(bend - 1) * nsteps + 4 * uint64_t(jb)) * 512; };
      const size_t first = tab.size();
      int lo = 0, hi = bend - 1;
      while (lo <= hi) {
        PStage e{};
        const int na = active(lo);
        e.a_src = rowless ? apack : src_of(lo);
        e.rs_bytes = uint32_t(nsteps) * 512u;
        e.info = uint32_t(lo);
        e.word = uint32_t(na) | (uint32_t(g) << PW_G_SHIFT) | ((chunk_id & 63u) << PW_CHUNK_SHIFT);
        uint64_t second = 0;
        if (merge && lo < hi && na + active(hi) <= kDuoUnits) {
          const int nb = active(hi);
          e.word |= PW_DUO | (uint32_t(nb) << PW_NB_SHIFT);
          e.info |= uint32_t(hi - lo) << 10;
          second = xa0 + uint64_t(hi) * xa_block;
          --hi;
        }
        if (rowless) e.word |= PW_SHARED;
        if (last_chunk) e.word |= PW_MEAN;
        if (!rowless && last_chunk && gh[g].narrow) e.word |= PW_NARROW;
        if (g == last_staged) e.word |= PW_LAST_GP;
        tab.push_back(e);
        xa.push_back(xa0 + uint64_t(lo) * xa_block);
        xb.push_back(second);
        gof.push_back(uint32_t(g));
        ++lo;
      }
      if (c == 0) tab[first].word |= PW_GP_FIRST;
      tab.back().word |= PW_CHUNK_END;
      if (last_chunk) {
        tab.back().word |= PW_GP_END;
        if (g == last_staged) tab.back().word |= PW_TILE_END;
      }
    }
  }
  for (size_t i = 0; i < tab.size(); ++i) {
    const size_t nx = (i + 1) % tab.size();
    tab[i].xa_next = xa[nx];
    tab[i].xb_next = xb[nx] ? uint32_t(xb[nx] - xa[nx]) : 0u;
    tab[i].info |= gof[nx] << 20;
  }
  ctx->pstage_chunk_off[Geff] = int(chunk_start.size());
  chunk_start.push_back(int(tab.size()));
  ctx->pstage_chunk_start = chunk_start;
  SGP_TRY(sgp_reserve(ctx, &ctx->pstage_tab, tab.size() * sizeof(PStage)));
  SGP_TRY(sgp_h2d(ctx, ctx->pstage_tab.p, tab.data(), tab.size() * sizeof(PStage)));
  ctx->pstage_sig = sig;
  ctx->pstage_count = int(tab.size());
  *dev = static_cast<const PStage*>(ctx->pstage_tab.p);
  *nstages = ctx->pstage_count;
  return 0;
}

// persistent: one workgroup per CU (512 threads, ~156 KB of LDS)
int pair_grid_blocks(int num_cu, int64_t N) {
  const int64_t ntiles = (N + kTileRows - 1) / kTileRows;
  return int(ntiles < num_cu ? ntiles : num_cu);
}

// How a launch spreads its tiles over the chip.  T tiles on C workgroups take
// ceil(T / C) rounds; a remainder of r < C tiles keeps r workgroups busy for a whole
// round while the others idle (config 5: 1563 tiles on 256 CUs = 6.1 rounds -> 7).
// When that costs more than ~1 %, the remainder tiles are cut into `parts` runs of
// whole accumulator chunks, balanced by stage count, one run per workgroup
// (PairParams::split_*); the same cut fills the chip when there are fewer tiles
// than CUs.  SGP_PAIR_SPLIT=0 switches it off (A/B runs).
struct PairPlan {
  int nblocks = 0;
  int tile0 = 0, parts = 0, count = 0;
  int s0[9] = {0};
};
PairPlan pair_plan(const sgp_ctx* ctx, int64_t N) {
  static const bool off = getenv("SGP_PAIR_SPLIT") && atoi(getenv("SGP_PAIR_SPLIT")) == 0;
  PairPlan pl;
  const int C = ctx->num_cu;
  const int64_t T = (N + kTileRows - 1) / kTileRows;
  pl.nblocks = int(T < C ? T : C);
  pl.tile0 = int(T);
  const std::vector<int>& cs = ctx->pstage_chunk_start;
  const int nchunks = int(cs.size()) - 1, nstages = ctx->pstage_count;
  const int64_t full = T / C;
  const int rem = int(T % C);
  if (off || (ctx->sweep_choice & 4) || rem == 0 || nchunks < 2 || nchunks > 64 || T >= (int64_t(1) << 30)) return pl;
  int parts = std::min(std::min(C / rem, nchunks), 8);
  if (parts < 2) return pl;
  // contiguous runs of chunks with the smallest possible longest run (the chunks
  // are few: try every bound)
  int best_max = nstages + 1, best_parts = 0, best_s0[9] = {0};
  for (int k = parts; k >= 2; --k) {
    // smallest bound B such that a greedy cut needs <= k runs
    int lo = 0, hi = nstages;
    for (int c = 0; c < nchunks; ++c) lo = std::max(lo, cs[c + 1] - cs[c]);
    auto runs_for = [&](int B, int* s0) {
      int runs = 0, start = 0;
      for (int c = 0; c < nchunks; ++c) {
        if (cs[c + 1] - cs[start] > B) {
          if (s0 && runs < 9) s0[runs] = cs[start];
          ++runs;
          start = c;
        }
      }
      if (s0 && runs < 9) s0[runs] = cs[start];
      return runs + 1;
    };
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (runs_for(mid, nullptr) <= k) hi = mid; else lo = mid + 1;
    }
    if (lo < best_max) {
      int s0[9] = {0};
      const int r = runs_for(lo, s0);
      s0[r] = nstages;
      best_max = lo;
      best_parts = r;
      for (int i = 0; i <= r; ++i) best_s0[i] = s0[i];
    }
  }
  if (best_parts < 2) return pl;
  // rounds with and without the cut (one stage ~ equal work; the cut pipeline
  // restarts once: count two stages for it)
  const double with = double(full) + double(best_max + 2) / nstages;
  const double without = double(full) + 1.0;
  if (with > 0.99 * without) return pl;
  pl.parts = best_parts;
  pl.count = rem;
  pl.tile0 = int(T) - rem;
  for (int i = 0; i <= best_parts; ++i) pl.s0[i] = best_s0[i];
  pl.nblocks = full > 0 ? C : rem * best_parts;
  return pl;
}

template <int D, int MODE, bool SINGLE, int R = 0, int SEP = 0>
int launch_pair_v(sgp_ctx* ctx, const PairParams& p, double flops) {
  static bool attr_set = false;
  if (!attr_set) {
    SGP_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(&k_sweep_pair<D, MODE, SINGLE, R, SEP>),
                     hipFuncAttributeMaxDynamicSharedMemorySize,
                     int(LayP<D, R>::bytes())));
    attr_set = true;
  }
  const PairPlan pl = pair_plan(ctx, p.pts.N);
  const int nblocks = pl.nblocks;
  PairParams pp = p;
  pp.split_tile0 = pl.tile0;
  pp.split_parts = pl.parts;
  pp.split_count = pl.count;
  for (int i = 0; i < 9; ++i) pp.split_s0[i] = pl.s0[i];
  pp.nchunks = int(ctx->pstage_chunk_start.size()) - 1;
  for (int g = 0; g <= SGP_MAX_GPS; ++g)
    pp.chunk_off[g] = g <= p.geff ? ctx->pstage_chunk_off[g] : ctx->pstage_chunk_off[p.geff];
  pp.split_partial0 = nblocks * kPairs;
  ctx->sweep_partials = nblocks * kPairs + (pl.parts > 0 ? pl.count * kPairs : 0);
  if (pl.parts > 0) {
    const size_t nt = size_t(pl.count) * pp.nchunks * 512, nm = size_t(pl.count) * p.geff * 512;
    SGP_TRY(sgp_reserve(ctx, &ctx->pair_split, (nt + nm) * sizeof(double)));
    pp.split_t = static_cast<double*>(ctx->pair_split.p);
    pp.split_m = pp.split_t + nt;
  }
  SweepTimer timer;
  SGP_TRY(timer.begin(ctx, flops));
#ifdef SGP_INSTRUMENT
  static const int ablate = getenv("SGP_ABLATE") ? atoi(getenv("SGP_ABLATE")) : 0;
  pp.ablate = ablate;
#endif
#ifdef PGP_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  if (!stamps_dev) SGP_HIP(ctx, hipMalloc(&stamps_dev, size_t(4096) * 64 * 8));
  pp.stamps = stamps_dev;
#endif
  hipLaunchKernelGGL((k_sweep_pair<D, MODE, SINGLE, R, SEP>), dim3(nblocks), dim3(512),
                     (LayP<D, R>::bytes()), ctx->stream, pp);
  if (pl.parts > 0)
    hipLaunchKernelGGL((k_pair_split_finish<MODE>), dim3(pl.count), dim3(512), 0,
                       ctx->stream, pp);
  SGP_HIP(ctx, hipGetLastError());
#ifdef PGP_STAMPS
  {
    std::vector<unsigned long long> h(size_t(nblocks) * 64);
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGP_HIP(ctx, hipMemcpy(h.data(), stamps_dev, h.size() * 8, hipMemcpyDeviceToHost));
    static const char* names[8] = {"epilogue", "dma-issue", "matrix", "evaluate",
                                   "gp-end", "dma-wait", "barrier", "-"};
    for (int half = 0; half < 2; ++half) {
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
      for (int b = 0; b < nblocks; ++b)
        for (int w = 4 * half; w < 4 * half + 4; ++w)
          for (int i = 0; i < 8; ++i) sum[i] += double(h[(size_t(b) * 8 + w) * 8 + i]);
      for (int i = 0; i < 8; ++i) tot += sum[i];
      fprintf(stderr, "stamps half %d (ticks per wave, %% of loop):", half);
      for (int i = 0; i < 7; ++i)
        fprintf(stderr, "  %s %.0f (%.1f%%)", names[i], sum[i] / (4.0 * nblocks),
                100.0 * sum[i] / tot);
      fprintf(stderr, "  | loop %.0f\n", tot / (4.0 * nblocks));
    }
  }
#endif
  return timer.end(ctx);
}

// (riders: single-part kernels up to d = 4 -- pair_riders)
template <int D>
int launch_pair_d(sgp_ctx* ctx, const PairParams& p, bool single, bool riders, double flops) {
  if constexpr (D <= 4) {
    if (riders) return launch_pair_v<D, MODE_CONF, true, kMaxRide>(ctx, p, flops);
  }
  return single ? launch_pair_v<D, MODE_CONF, true>(ctx, p, flops)
                : launch_pair_v<D, MODE_CONF, false>(ctx, p, flops);
}

}  // namespace

// The paired kernel pays off once a factor needs more than one pass of the
// 4-wave kernel (more than 256 rows); sgp_ctx_set_sweep or SGP_SWEEP=pair|classic
// force a choice (A/B runs of profiles/, tests).
bool pair_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff) {
  static const char* force = getenv("SGP_SWEEP");
  if ((ctx->sweep_choice & 3) == 2) return true;
  if ((ctx->sweep_choice & 3) == 1) return false;
  if (force && force[0] == 'p') return true;
  if (force && force[0] == 'c') return false;
  int np = 0;
  for (int g = 0; g < Geff; ++g) np = std::max(np, gh[g].n_pad);
  return np > 256;
}

int pair_sweep_partials(const sgp_ctx* ctx, int64_t N) {
  (void)N;
  return ctx->sweep_partials;     // set by the launch (pair_plan)
}

int launch_sweep_pair(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d,
                      int Geff, double flops, const SepLaunch* sep) {
  PairParams p{};
  p.gps = a.gps;
  p.G = a.G;
  p.pts = a.pts;
  p.conf = a.conf;
  p.fit = a.fit;
  p.geff = Geff;
  p.shared_mask = 0;
  for (int g = 0; g < Geff; ++g)
    if (gh[g].share >= 0) p.shared_mask |= 1u << g;
  // (a.mode is MODE_CONF: launch_sweep turns a fitness call into posterior + shaping)
  bool single = true;
  for (int g = 0; g < Geff; ++g) single = single && gh[g].kern.n_parts == 1;
  bool rides[SGP_MAX_GPS] = {};
  static const bool no_ride = getenv("SGP_PAIR_RIDE") && atoi(getenv("SGP_PAIR_RIDE")) == 0;
  for (int g = 0; g < SGP_MAX_GPS; ++g) {
    p.nride[g] = 0;
    p.ride_delta[g] = 0;
  }
  const bool riders = !no_ride && sweep_riders(gh, Geff, d, single, kMaxRide, 4, rides, p.nride);
  // (factor tables: the instances without riders; a launch with riders evaluates)
  if (riders) sep = nullptr;
  if (!riders)
    for (int g = 0; g < Geff; ++g) {
      rides[g] = false;
      p.nride[g] = 0;
    }
  for (int g = 0, leader = 0; g < Geff; ++g) {
    if (!rides[g]) {
      leader = g;
      continue;
    }
    p.ride_delta[g] = (long long)(reinterpret_cast<intptr_t>(gh[g].XA) -
                                  reinterpret_cast<intptr_t>(gh[leader].XA));
  }
  SGP_TRY(pair_stage_table(ctx, gh, Geff, d, sep != nullptr, rides, &p.stages, &p.nstages));
  if (sep) {
    p.sep = *sep;
    switch (sep->naxes) {
      case 1: return launch_pair_v<1, MODE_CONF, true, 0, 1>(ctx, p, flops);
      case 2: return launch_pair_v<1, MODE_CONF, true, 0, 2>(ctx, p, flops);
      case 3: return launch_pair_v<1, MODE_CONF, true, 0, 3>(ctx, p, flops);
    }
    sgp_set_error(ctx, "factor tables with %d axes", sep->naxes);
    return -2;
  }
  int rc = -2;
  switch (d) {
    case 1: rc = launch_pair_d<1>(ctx, p, single, riders, flops); break;
    case 2: rc = launch_pair_d<2>(ctx, p, single, riders, flops); break;
    case 3: rc = launch_pair_d<3>(ctx, p, single, riders, flops); break;
    case 4: rc = launch_pair_d<4>(ctx, p, single, riders, flops); break;
    case 5: rc = launch_pair_d<5>(ctx, p, single, riders, flops); break;
    case 6: rc = launch_pair_d<6>(ctx, p, single, riders, flops); break;
    case 7: rc = launch_pair_d<7>(ctx, p, single, riders, flops); break;
    case 8: rc = launch_pair_d<8>(ctx, p, single, riders, flops); break;
    default:
      sgp_set_error(ctx, "input dimension %d not in 1..%d", d, SGP_MAX_D);
      return -2;
  }
  return rc;
}
