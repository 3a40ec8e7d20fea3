// Round-5 probe: what do the non-matrix instructions of one slot of the paired sweep
// (16 MFMAs; two LDS reads of the next slot's A operand, a compare, a branch, the two
// s_nop 0 the compiler's hazard recogniser puts between asm statements, a waitcnt)
// cost per slot -- when both waves of a SIMD stream slots, and when only one does
// (its partner evaluating, or waiting at the stage barrier)?
// One 512-thread block, waves w and w+4 share a SIMD; fixed duration; s_memtime ticks.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define CLOB "v10","v11","v12","v13","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","scc","memory"
#define M0 "v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n"
#define M1 "v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n"
#define M2 "v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n"
#define M3 "v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n"
#define G M0 M1 M2 M3
#define RD0 "ds_read2st64_b64 v[42:45], %0 offset0:4 offset1:5\n"
#define RD1 "ds_read2st64_b64 v[46:49], %0 offset0:6 offset1:7\n"
#define WT "s_waitcnt lgkmcnt(0)\n"
#define N0 "s_nop 0\n"
#define N1 "s_nop 1\n"
#define CMP "s_cmp_eq_u32 %1, 1\n"
#define BR "s_cbranch_scc1 1f\n1:\n"
#define FMA "v_fma_f64 v[42:43], v[10:11], v[12:13], v[10:11]\n"

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

// a block = 4 slots of 16 MFMAs
template <int K>
__device__ __forceinline__ void block(unsigned a, unsigned z) {
  if (K == 0) asm volatile(REP4(G G G G) :: "v"(a), "s"(z) : CLOB);
  if (K == 1) asm volatile(REP4(RD0 RD1 G G G G WT) :: "v"(a), "s"(z) : CLOB);
  if (K == 2) asm volatile(REP4(RD0 RD1 G CMP G N0 G N0 G WT) :: "v"(a), "s"(z) : CLOB);
  if (K == 3) asm volatile(REP4(RD0 RD1 G CMP G N0 G N0 G WT BR) :: "v"(a), "s"(z) : CLOB);   // the slot as compiled in round 5
  if (K == 4) asm volatile(REP4(G RD0 G RD1 G CMP G WT BR) :: "v"(a), "s"(z) : CLOB);          // reads between the groups
  if (K == 5) asm volatile(REP4(G RD0 G RD1 G G WT) :: "v"(a), "s"(z) : CLOB);                 // ... and no branch
  if (K == 6) asm volatile(REP4(N1 G G G G) :: "v"(a), "s"(z) : CLOB);
  if (K == 7) asm volatile(REP4(G G G G CMP BR) :: "v"(a), "s"(z) : CLOB);
  if (K == 8) asm volatile(REP4(RD0 RD1 N1 G CMP G N0 G N0 G WT BR) :: "v"(a), "s"(z) : CLOB);  // round 4: + s_nop 1 per slot
  if (K == 9) asm volatile(REP4(G G G G FMA) :: "v"(a), "s"(z) : CLOB);                          // one fp64 VALU per slot
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k_pair(unsigned long long ticks, unsigned* out) {
  __shared__ double lds[8192];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 1.0;
  asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n"
               "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
               "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n" ::: CLOB);
  __syncthreads();
  const unsigned a = (threadIdx.x & 63) * 8 + (unsigned)(size_t)lds;   // LDS byte address of the lane
  const unsigned z = __builtin_amdgcn_readfirstlane(out[15]);          // 0, not known to the compiler
  const unsigned long long t0 = now();
  unsigned n = 0;
  if (wave < 4) {
    if (KA >= 0) while (now() - t0 < ticks) { block<KA>(a, z); ++n; }
  } else {
    if (KB >= 0) while (now() - t0 < ticks) { block<KB>(a, z); ++n; }
  }
  if ((threadIdx.x & 63) == 0) out[wave] = n;
}

static double base2 = 0, base1 = 0;
template <int KA, int KB>
void run(unsigned* dout, const char* label) {
  const unsigned long long ticks = 4000000;
  hipMemset(dout, 0, 64);
  k_pair<KA, KB><<<1, 512>>>(ticks, dout);
  unsigned h[8];
  hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
  const double na = (h[0] + h[1] + h[2] + h[3]) / 4.0, nb = (h[4] + h[5] + h[6] + h[7]) / 4.0;
  const double slots = 4.0 * (na * (KA >= 0) + nb * (KB >= 0));   // per SIMD
  const double t = ticks / slots;
  double& base = (KB >= 0) ? base2 : base1;
  if (KA == 0) base = t;
  printf("  %-58s %7.1f ticks per slot per SIMD  (%+6.1f)\n", label, t, t - base);
}

int main() {
  unsigned* dout;
  hipMalloc(&dout, 64);
  printf("both waves of a SIMD stream slots:\n");
  run<0, 0>(dout, "16 MFMAs");
  run<1, 1>(dout, "+ 2 LDS reads in front, waitcnt behind");
  run<2, 2>(dout, "+ s_cmp, 2 x s_nop 0 between the groups");
  run<3, 3>(dout, "+ s_cbranch (not taken): the slot of round 5");
  run<8, 8>(dout, "+ s_nop 1 in front: the slot of round 4");
  run<4, 4>(dout, "reads between the groups, no s_nop, cmp + branch");
  run<5, 5>(dout, "reads between the groups, no s_nop, no branch");
  run<6, 6>(dout, "16 MFMAs + s_nop 1");
  run<7, 7>(dout, "16 MFMAs + cmp + branch");
  run<9, 9>(dout, "16 MFMAs + one v_fma_f64");
  printf("one wave of a SIMD streams, its partner idle:\n");
  run<0, -1>(dout, "16 MFMAs");
  run<1, -1>(dout, "+ 2 LDS reads in front, waitcnt behind");
  run<2, -1>(dout, "+ s_cmp, 2 x s_nop 0 between the groups");
  run<3, -1>(dout, "+ s_cbranch (not taken): the slot of round 5");
  run<8, -1>(dout, "+ s_nop 1 in front: the slot of round 4");
  run<4, -1>(dout, "reads between the groups, no s_nop, cmp + branch");
  run<5, -1>(dout, "reads between the groups, no s_nop, no branch");
  run<6, -1>(dout, "16 MFMAs + s_nop 1");
  run<7, -1>(dout, "16 MFMAs + cmp + branch");
  run<9, -1>(dout, "16 MFMAs + one v_fma_f64");
  return 0;
}
