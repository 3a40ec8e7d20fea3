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

// sweep_pair.hip
bool pair_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff);
int pair_sweep_partials(const sgp_ctx* ctx, int64_t N);
int launch_sweep_pair(sgp_ctx* ctx, const SweepArgs& a, const GpDev* gh, int d,
                      int Geff, double flops);
