// Round-2 probe: what does a wave cost its SIMD partner?  One 512-thread block:
// waves w and w+4 share a SIMD.  Waves 0-3 run stream A, waves 4-7 stream B,
// both for a fixed number of ticks; each wave reports how many 64-instruction
// blocks it completed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41"

enum { S_IDLE, S_MFMA, S_FMA, S_SWZ, S_DSREAD, S_DMA, S_SALU, S_IADD, S_MIX, S_NKINDS };
static const char* kNames[] = {"idle", "mfma", "v_fma_f64", "ds_swizzle", "ds_read_b64", "lds-dma 1KB", "salu", "v_add_u32", "mfma+fma 1:1"};

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

template <int KIND>
__device__ __forceinline__ void block64(const char* gsrc, double* lbuf) {
  if (KIND == S_MFMA)
    asm volatile(REP16("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n") ::: CLOB);
  if (KIND == S_FMA)
    asm volatile(REP16("v_fma_f64 v[10:11], v[10:11], v[18:19], v[20:21]\n v_fma_f64 v[12:13], v[12:13], v[18:19], v[20:21]\n v_fma_f64 v[14:15], v[14:15], v[18:19], v[20:21]\n v_fma_f64 v[16:17], v[16:17], v[18:19], v[20:21]\n") ::: CLOB);
  if (KIND == S_SWZ)
    asm volatile(REP16("ds_swizzle_b32 v26, v30 offset:swizzle(BITMASK_PERM, \"00p00\")\n ds_swizzle_b32 v27, v31 offset:swizzle(BITMASK_PERM, \"00p00\")\n ds_swizzle_b32 v28, v32 offset:swizzle(BITMASK_PERM, \"00p00\")\n ds_swizzle_b32 v29, v33 offset:swizzle(BITMASK_PERM, \"00p00\")\n") "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
  if (KIND == S_DSREAD) {
    const unsigned addr = unsigned(reinterpret_cast<uintptr_t>(lbuf)) + (threadIdx.x & 63) * 8;
    asm volatile("v_mov_b32 v30, %0\n" REP16("ds_read_b64 v[10:11], v30\n ds_read_b64 v[12:13], v30 offset:512\n ds_read_b64 v[14:15], v30 offset:1024\n ds_read_b64 v[16:17], v30 offset:1536\n") "s_waitcnt lgkmcnt(0)\n" :: "v"(addr) : CLOB);
  }
  if (KIND == S_DMA) {
    // 16 x 1 KB global -> LDS pieces (counts as 16 "instructions" x4 for the table)
    const char* src = gsrc + (threadIdx.x & 63) * 16;
    double* dst = lbuf;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + i * 128), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)");
  }
  if (KIND == S_SALU)
    asm volatile(REP16("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n") ::: "s20", "s21", "s22", "s23");
  if (KIND == S_IADD)
    asm volatile(REP16("v_add_u32 v26, v26, v30\n v_add_u32 v27, v27, v30\n v_add_u32 v28, v28, v30\n v_add_u32 v29, v29, v30\n") ::: CLOB);
  if (KIND == S_MIX)
    asm volatile(REP4(REP4(REP4("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n")) REP4(REP4("v_fma_f64 v[14:15], v[14:15], v[18:19], v[20:21]\n"))) ::: CLOB);
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k_pair(const char* gsrc, unsigned long long ticks, unsigned* out) {
  __shared__ double lbuf[2][4][2048];
  const int wave = threadIdx.x >> 6;
  const bool second = wave >= 4;
  asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n"
               "v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n"
               "v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n"
               "v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n"
               "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
               "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n" ::: CLOB);
  __syncthreads();
  const unsigned long long t0 = now();
  unsigned n = 0;
  if (!second) {
    if (KA != S_IDLE) while (now() - t0 < ticks) { block64<KA>(gsrc, &lbuf[0][wave & 3][0]); ++n; }
  } else {
    if (KB != S_IDLE) while (now() - t0 < ticks) { block64<KB>(gsrc, &lbuf[1][wave & 3][0]); ++n; }
  }
  if ((threadIdx.x & 63) == 0) out[wave] = n;
}

template <int KA, int KB>
void run(const char* gsrc, unsigned* dout) {
  const unsigned long long ticks = 4000000;
  hipMemset(dout, 0, 64);
  k_pair<KA, KB><<<1, 512>>>(gsrc, ticks, dout);
  unsigned h[8];
  hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
  auto per = [&](int kind, double nblocks) {
    if (kind == S_IDLE || nblocks == 0) return 0.0;
    const double ninstr = (kind == S_DMA) ? 16.0 : 64.0;
    return ticks / (nblocks * ninstr);
  };
  const double na = (h[0] + h[1] + h[2] + h[3]) / 4.0, nb = (h[4] + h[5] + h[6] + h[7]) / 4.0;
  printf("  A=%-13s B=%-13s  ticks/instr  A %7.2f   B %7.2f\n", kNames[KA], kNames[KB], per(KA, na), per(KB, nb));
}

int main() {
  char* gsrc; unsigned* dout;
  hipMalloc(&gsrc, 1 << 20); hipMemset(gsrc, 0, 1 << 20); hipMalloc(&dout, 64);
  printf("two waves per SIMD (waves w, w+4 of one 512-thread block), fixed duration:\n");
  run<S_MFMA, S_IDLE>(gsrc, dout);
  run<S_FMA, S_IDLE>(gsrc, dout);
  run<S_SWZ, S_IDLE>(gsrc, dout);
  run<S_DSREAD, S_IDLE>(gsrc, dout);
  run<S_DMA, S_IDLE>(gsrc, dout);
  run<S_SALU, S_IDLE>(gsrc, dout);
  run<S_IADD, S_IDLE>(gsrc, dout);
  run<S_MIX, S_IDLE>(gsrc, dout);
  run<S_MFMA, S_MFMA>(gsrc, dout);
  run<S_MFMA, S_FMA>(gsrc, dout);
  run<S_MFMA, S_IADD>(gsrc, dout);
  run<S_MFMA, S_SWZ>(gsrc, dout);
  run<S_MFMA, S_DSREAD>(gsrc, dout);
  run<S_MFMA, S_DMA>(gsrc, dout);
  run<S_MFMA, S_SALU>(gsrc, dout);
  run<S_FMA, S_FMA>(gsrc, dout);
  run<S_FMA, S_IADD>(gsrc, dout);
  run<S_MIX, S_MIX>(gsrc, dout);
  run<S_MFMA, S_MIX>(gsrc, dout);
  return 0;
}
