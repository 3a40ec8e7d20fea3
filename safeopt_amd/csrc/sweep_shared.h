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
int launch_sweep_pair(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d,
                      int Geff, double flops, const SepLaunch* sep);
