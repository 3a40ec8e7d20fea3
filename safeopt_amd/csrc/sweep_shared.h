// Shared by the two posterior-sweep translation units (sweep.hip: the 4-wave
// kernel for small factors; sweep_pair.hip: the paired-wave kernel).
#pragma once

#include "kern_eval.h"
#include "fitness.h"

enum { MODE_CONF = 0, MODE_FITNESS = 1 };

// What a sweep launch works on (both kernels).
struct SweepArgs {
  const GpDev* gps;
  int G;
  int mode;
  SweepPoints pts;
  ConfOut conf;
  FitnessArgs fit;
};

// hipEvent pair around a sweep launch on the library's own stream
// (sgp_profile_enable): bench.py's roofline.achieved comes from these.
struct SweepTimer {
  hipEvent_t e1 = nullptr;
  int begin(sgp_ctx* ctx, double flops) {
    if (!ctx->profiling) return 0;
    if (ctx->prof_used + 2 > ctx->prof_events.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        SGP_HIP(ctx, hipEventCreate(&e));
        ctx->prof_events.push_back(e);
      }
    }
    hipEvent_t e0 = ctx->prof_events[ctx->prof_used];
    e1 = ctx->prof_events[ctx->prof_used + 1];
    ctx->prof_used += 2;
    ctx->prof_flops += flops;
    SGP_HIP(ctx, hipEventRecord(e0, ctx->stream));
    return 0;
  }
  int end(sgp_ctx* ctx) {
    if (e1) SGP_HIP(ctx, hipEventRecord(e1, ctx->stream));
    return 0;
  }
};

// (operand maps of v_mfma_f64_4x4x4_4b_f64: sweep.hip, "matrix part")
// Covariance values of a stage -> B operands.  Lane (k, c) = (l >> 4, l & 15)
// holds kv[q] = k(X_{4q+k}, x_c); operand (q, m) of lane (k, a, j) is kv[q] of
// lane (k, m, j).  Through a wave-private LDS buffer laid out [k][c][q]: two
// 16-byte stores and eight 16-byte loads per lane (LDS instructions of one wave
// execute in order; no barrier).
template <int kKbRow>
__device__ __forceinline__ void broadcast_quads(const double (&kv)[4], double* kbw,
                                                int lane, double (&kb)[4][4]) {
  double2_t* w = reinterpret_cast<double2_t*>(kbw + (lane >> 4) * kKbRow +
                                              (lane & 15) * 4);
  w[0] = double2_t{kv[0], kv[1]};
  w[1] = double2_t{kv[2], kv[3]};
  __builtin_amdgcn_wave_barrier();
  const double2_t* r = reinterpret_cast<const double2_t*>(
      kbw + (lane >> 4) * kKbRow + (lane & 3) * 4);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double2_t a = r[m * 8], b = r[m * 8 + 1];
    kb[m][0] = a.x; kb[m][1] = a.y; kb[m][2] = b.x; kb[m][3] = b.y;
  }
  __builtin_amdgcn_wave_barrier();
}

// Lane exchanges without the LDS (gfx950; semantics probed, scripts/dev/probe_dpp... in
// profiles/r05/experiments.txt, section 14): a row rotation by DPP -- lane i of a 16-lane
// row receives lane (i - n) mod 16 -- and the row / half swaps of v_permlane16_swap /
// v_permlane32_swap.  One or two VALU instructions per dword instead of a ds_bpermute
// round trip (~100 cycles of latency, two LDS instructions per double).
template <int ROR>
__device__ __forceinline__ double row_ror(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + ROR, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + ROR, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// what lane l ^ 4 / l ^ 8 holds in `v` (bit2: (l & 4) != 0)
__device__ __forceinline__ double take_xor4(double v, bool bit2) {
  const double up = row_ror<4>(v), down = row_ror<12>(v);   // from l - 4, from l + 4
  return bit2 ? up : down;
}
__device__ __forceinline__ double take_xor8(double v) { return row_ror<8>(v); }
// v + (v of lane l ^ 16), v + (v of lane l ^ 32); together = sum_lane_groups, same order
__device__ __forceinline__ double add_xor16(double v) {
  const unsigned lo = unsigned(__double2loint(v)), hi = unsigned(__double2hiint(v));
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double even = __hiloint2double(int(b[0]), int(a[0])), odd = __hiloint2double(int(b[1]), int(a[1]));
  // (even: rows 0 0 2 2 of v, odd: rows 1 1 3 3; the lane's own value first, as v + shfl(v))
  const bool own_even = (threadIdx.x & 16) == 0;
  return own_even ? even + odd : odd + even;
}
__device__ __forceinline__ double add_xor32(double v) {
  const unsigned lo = unsigned(__double2loint(v)), hi = unsigned(__double2hiint(v));
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double low = __hiloint2double(int(b[0]), int(a[0])), high = __hiloint2double(int(b[1]), int(a[1]));
  const bool own_low = (threadIdx.x & 32) == 0;
  return own_low ? low + high : high + low;
}
__device__ __forceinline__ double sum_lane_groups_valu(double v) { return add_xor32(add_xor16(v)); }

// MFMA with the accumulator tied to destination AND addend (the builtin lets the
// register allocator rename the destination, which costs v_mov_b64 copies at every
// join of a guarded slot sequence).
__device__ __forceinline__ void mfma_acc(double& c, double a, double b) {
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// ... with the accumulator in an ACCUMULATION register (the AccVGPR half of the
// unified register file, gfx90a+): the 128 registers of a wave's 16 slots then do not
// compete with anything the compiler allocates among the architectural VGPRs.
__device__ __forceinline__ void mfma_acc_a(double& c, double a, double b) {
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// global -> LDS without a VGPR round trip: "scalar base + 32-bit lane offset"
// (the builtin only produces the 64-bit-VGPR-address form, one VALU add per
// copy).  M0 carries the wave-uniform LDS byte address.
// (cache policy of the A-chunk copies: experiment switch, profiles/r04/experiments.txt)
#ifndef SGP_DMA_POLICY
#define SGP_DMA_POLICY ""
#endif
__device__ __forceinline__ void dma_2k(uint64_t src, uint32_t lds_addr, uint32_t voff) {
  asm volatile(
      "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2" SGP_DMA_POLICY "\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024" SGP_DMA_POLICY
      :: "s"(lds_addr), "v"(voff), "s"(src) : "memory", "m0");
}
__device__ __forceinline__ void dma_1k(uint64_t src, uint32_t lds_addr, uint32_t voff) {
  asm volatile(
      "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :: "s"(lds_addr), "v"(voff), "s"(src) : "memory", "m0");
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t lds_addr_of(const double* p) {
  return uint32_t(uintptr_t((const __attribute__((address_space(3))) void*)p));
}

// Which GPs of a launch RIDE with the GP in front of them: a follower (GpDev::share >= 0:
// same inputs, kernel, noise and fitting history -- the outputs of a multi-output GP) has
// the leader's L^-1 AND its covariances with every candidate, so its alpha . k is formed
// in the leader's stages and it needs no stages of its own.  Single-part kernels, d <= max_d
// (LDS room for the riders' alpha chunks, registers), the first `max_ride` followers of a
// leader.
// rides[g]: GP g rides; nride[g]: riders of leader g.  Returns whether any GP rides.
inline bool sweep_riders(const GpDev* gh, int Geff, int d, bool single, int max_ride,
                         int max_d, bool* rides, int* nride) {
  bool any = false;
  int leader = 0;
  for (int g = 0; g < Geff; ++g) {
    rides[g] = false;
    nride[g] = 0;
    if (gh[g].share < 0) {
      leader = g;
      continue;
    }
    if (single && d <= max_d && g - leader <= max_ride && nride[leader] == g - leader - 1) {
      rides[g] = true;
      ++nride[leader];
      any = true;
    }
  }
  return any;
}

// a wave-uniform pointer, pinned to scalar registers
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* q) {
  const uint64_t v = reinterpret_cast<uint64_t>(q);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
  return reinterpret_cast<const T*>((uint64_t(hi) << 32) | lo);
}

// sweep_tiny.hip: every GP of the launch has at most kTinyMaxN observations
constexpr int kTinyMaxN = 48;
bool tiny_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, int64_t rows,
                       bool rows_sharded);
int launch_sweep_tiny(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d, int Geff,
                      double flops);

// sweep_pair.hip
bool pair_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff);
int pair_sweep_partials(const sgp_ctx* ctx, int64_t N);
// sweep_mid.hip: the resident-factor kernel for 49 .. 128 observations (mid_sweep_wanted)
int launch_sweep_mid(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d, int Geff,
                     double flops, const SepLaunch* sep);
int launch_sweep_pair(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d,
                      int Geff, double flops, const SepLaunch* sep);
