// Round-3 probe: what does ONE fp64 VALU instruction cost when it sits between the
// MFMAs of the issuing wave's own stream -- one per K matrix instructions, both waves
// of a SIMD running the same stream (the matrix phase of the paired sweep)?
// One 512-thread block, waves w and w+4 share a SIMD; fixed duration; reported:
// s_memtime ticks per MFMA per SIMD (= per wave / 2 when both waves stream).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v34","v35","v36","v37","v38","v39","v40","v41"
#define M0 "v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n"
#define M1 "v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n"
#define M2 "v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n"
#define M3 "v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n"
#define F0 "v_fma_f64 v[14:15], v[14:15], v[18:19], v[20:21]\n"
#define F1 "v_fma_f64 v[16:17], v[16:17], v[18:19], v[20:21]\n"
#define F2 "v_fma_f64 v[22:23], v[22:23], v[18:19], v[20:21]\n"
#define F3 "v_fma_f64 v[24:25], v[24:25], v[18:19], v[20:21]\n"

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

// a block = 64 MFMAs with VALU sprinkled in
template <int K>
__device__ __forceinline__ void block() {
  if (K == 0) asm volatile(REP16(M0 M1 M2 M3) ::: CLOB);                        // MFMA only
  if (K == 1) asm volatile(REP16(M0 F0 M1 F1 M2 F2 M3 F3) ::: CLOB);            // 1 : 1
  if (K == 2) asm volatile(REP16(M0 M1 F0 M2 M3 F1) ::: CLOB);                  // 1 : 2
  if (K == 4) asm volatile(REP16(M0 M1 M2 M3 F0) ::: CLOB);                     // 1 : 4
  if (K == 8) asm volatile(REP4(REP4(M0 M1 M2 M3) M0 M1 M2 M3 F0 M0 M1 M2 M3 F1) ::: CLOB);   // 1 : 8 (96 MFMA!)
  if (K == 16) asm volatile(REP4(M0 M1 M2 M3 M0 M1 M2 M3 M0 M1 M2 M3 M0 M1 M2 M3 F0) ::: CLOB);  // 1 : 16
  if (K == 44) asm volatile(REP4(M0 M1 M2 M3 F0 F1 F2 F3 M0 M1 M2 M3 M0 M1 M2 M3 M0 M1 M2 M3) ::: CLOB);  // bursts of 4 per 16
  if (K == 99) asm volatile(REP16(F0 F1 F2 F3) ::: CLOB);                       // VALU only (64 FMA)
}
template <int K> constexpr int mfmas() { return K == 8 ? 96 : (K == 99 ? 0 : 64); }
template <int K> constexpr int valus() {
  return K == 0 ? 0 : K == 1 ? 64 : K == 2 ? 32 : K == 4 ? 16 : K == 8 ? 8 : K == 16 ? 4 : K == 44 ? 16 : 64;
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k_pair(unsigned long long ticks, unsigned* out) {
  const int wave = threadIdx.x >> 6;
  asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n"
               "v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n"
               "v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n"
               "v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n"
               "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
               "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n" ::: CLOB);
  __syncthreads();
  const unsigned long long t0 = now();
  unsigned n = 0;
  if (wave < 4) {
    if (KA >= 0) while (now() - t0 < ticks) { block<KA>(); ++n; }
  } else {
    if (KB >= 0) while (now() - t0 < ticks) { block<KB>(); ++n; }
  }
  if ((threadIdx.x & 63) == 0) out[wave] = n;
}

template <int KA, int KB>
void run(unsigned* dout, const char* label) {
  const unsigned long long ticks = 4000000;
  hipMemset(dout, 0, 64);
  k_pair<KA, KB><<<1, 512>>>(ticks, dout);
  unsigned h[8];
  hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
  const double na = (h[0] + h[1] + h[2] + h[3]) / 4.0, nb = (h[4] + h[5] + h[6] + h[7]) / 4.0;
  const double m = na * mfmas<(KA < 0 ? 0 : KA)>() * (KA >= 0) + nb * mfmas<(KB < 0 ? 0 : KB)>() * (KB >= 0);
  const double v = na * valus<(KA < 0 ? 0 : KA)>() * (KA >= 0) + nb * valus<(KB < 0 ? 0 : KB)>() * (KB >= 0);
  // model: ticks = m * c_mfma + v * c_valu with c_mfma from the MFMA-only run
  printf("  %-44s MFMA/SIMD %9.0f  VALU/SIMD %9.0f  ticks per MFMA %6.2f", label, m, v, m > 0 ? ticks / m : 0.0);
  static double c_mfma = 0.0;
  if (v == 0 && m > 0 && KA >= 0 && KB >= 0) c_mfma = ticks / m;
  if (v > 0 && m > 0 && c_mfma > 0) printf("   -> %5.2f ticks per VALU on top of %.2f per MFMA", (ticks - m * c_mfma) / v, c_mfma);
  if (m == 0 && v > 0) printf("   %5.2f ticks per VALU", ticks / v);
  printf("\n");
}

int main() {
  unsigned* dout;
  hipMalloc(&dout, 64);
  printf("VALU between the MFMAs of a wave's own stream (both waves of a SIMD stream, 4 independent accumulators):\n");
  run<0, 0>(dout, "both: MFMA only");
  run<99, 99>(dout, "both: v_fma_f64 only");
  run<1, 1>(dout, "both: 1 v_fma_f64 per MFMA");
  run<2, 2>(dout, "both: 1 per 2");
  run<4, 4>(dout, "both: 1 per 4");
  run<8, 8>(dout, "both: 1 per 8");
  run<16, 16>(dout, "both: 1 per 16");
  run<44, 44>(dout, "both: burst of 4 per 16");
  run<0, 4>(dout, "A MFMA only, B 1 per 4");
  run<0, -1>(dout, "one wave: MFMA only");
  run<4, -1>(dout, "one wave: 1 per 4");
  run<16, -1>(dout, "one wave: 1 per 16");
  return 0;
}
