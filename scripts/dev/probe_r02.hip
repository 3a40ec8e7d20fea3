// Round-2 hardware probes (run on the GPU box):
//   1. does v_mfma_f64_4x4x4_4b_f64 honour cbsz/abid (A-block broadcast)?
//   2. issue cost / dependent latency of the fp64 VALU ops of the covariance
//      evaluation (one wave per SIMD, s_memtime around unrolled chains).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/probe scripts/dev/probe_r02.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k_cbsz(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[0 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
  d[1 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 0, 0);
  d[2 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 1, 0);
  d[3 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 2, 0);
  d[4 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 3, 0);
  d[5 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 1, 0, 0);
  d[6 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 1, 1, 0);
}

__global__ __launch_bounds__(256) void k_calib(unsigned long long* out, int iters) {
  unsigned long long t0, t1;
  double a = threadIdx.x, b = 1.0;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
  }
  asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(c0), "v"(c1), "v"(c2), "v"(c3));
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (c0 + c1 + c2 + c3 == 12345.678) out[0] = 0;
}

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// one timed block: `body` is an asm string using v[10:11]..v[40:41] freely
#define TIMED(name, ninstr, body)                                              \
  {                                                                            \
    unsigned long long t0, t1;                                                 \
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)); \
    for (int it = 0; it < 8; ++it) { asm volatile(body ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45"); } \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = double(t1 - t0) / (8.0 * (ninstr)); \
    ++slot;                                                                    \
  }

__global__ __launch_bounds__(64) void k_time(double* out) {
  int slot = 0;
  // initialise registers with benign values
  asm volatile(
      "v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n"
      "v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n"
      "v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n"
      "v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n"
      "v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n"
      "v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n"
      "v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n"
      "v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n"
      "v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n"
      "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
      "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n"
      ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41");
  // 0/1: v_fma_f64 dependent / 4 independent
  TIMED("fma dep", 64, REP64("v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n"))
  TIMED("fma x4", 64, REP16("v_fma_f64 v[10:11], v[10:11], v[18:19], v[20:21]\n v_fma_f64 v[12:13], v[12:13], v[18:19], v[20:21]\n v_fma_f64 v[14:15], v[14:15], v[18:19], v[20:21]\n v_fma_f64 v[16:17], v[16:17], v[18:19], v[20:21]\n"))
  // 2/3: v_mul_f64
  TIMED("mul dep", 64, REP64("v_mul_f64 v[10:11], v[10:11], v[12:13]\n"))
  TIMED("mul x4", 64, REP16("v_mul_f64 v[10:11], v[10:11], v[18:19]\n v_mul_f64 v[12:13], v[12:13], v[18:19]\n v_mul_f64 v[14:15], v[14:15], v[18:19]\n v_mul_f64 v[16:17], v[16:17], v[18:19]\n"))
  // 4/5: v_add_f64
  TIMED("add dep", 64, REP64("v_add_f64 v[10:11], v[10:11], v[12:13]\n"))
  TIMED("add x4", 64, REP16("v_add_f64 v[10:11], v[10:11], v[18:19]\n v_add_f64 v[12:13], v[12:13], v[18:19]\n v_add_f64 v[14:15], v[14:15], v[18:19]\n v_add_f64 v[16:17], v[16:17], v[18:19]\n"))
  // 6/7: v_rsq_f64
  TIMED("rsq dep", 64, REP64("v_rsq_f64 v[10:11], v[10:11]\n"))
  TIMED("rsq x4", 64, REP16("v_rsq_f64 v[10:11], v[10:11]\n v_rsq_f64 v[12:13], v[12:13]\n v_rsq_f64 v[14:15], v[14:15]\n v_rsq_f64 v[16:17], v[16:17]\n"))
  // 8/9: v_rndne_f64
  TIMED("rndne dep", 64, REP64("v_rndne_f64 v[10:11], v[10:11]\n"))
  TIMED("rndne x4", 64, REP16("v_rndne_f64 v[10:11], v[10:11]\n v_rndne_f64 v[12:13], v[12:13]\n v_rndne_f64 v[14:15], v[14:15]\n v_rndne_f64 v[16:17], v[16:17]\n"))
  // 10: v_cvt_i32_f64 x4
  TIMED("cvt x4", 64, REP16("v_cvt_i32_f64 v26, v[10:11]\n v_cvt_i32_f64 v27, v[12:13]\n v_cvt_i32_f64 v28, v[14:15]\n v_cvt_i32_f64 v29, v[16:17]\n"))
  // 11/12: v_ldexp_f64
  TIMED("ldexp dep", 64, REP64("v_ldexp_f64 v[10:11], v[10:11], v26\n"))
  TIMED("ldexp x4", 64, REP16("v_ldexp_f64 v[10:11], v[10:11], v26\n v_ldexp_f64 v[12:13], v[12:13], v26\n v_ldexp_f64 v[14:15], v[14:15], v26\n v_ldexp_f64 v[16:17], v[16:17], v26\n"))
  // 13: v_max_f64 x4
  TIMED("max x4", 64, REP16("v_max_f64 v[10:11], v[10:11], v[18:19]\n v_max_f64 v[12:13], v[12:13], v[18:19]\n v_max_f64 v[14:15], v[14:15], v[18:19]\n v_max_f64 v[16:17], v[16:17], v[18:19]\n"))
  // 14: v_sqrt_f64 x4
  TIMED("sqrt x4", 64, REP16("v_sqrt_f64 v[10:11], v[10:11]\n v_sqrt_f64 v[12:13], v[12:13]\n v_sqrt_f64 v[14:15], v[14:15]\n v_sqrt_f64 v[16:17], v[16:17]\n"))
  // 15/16: 32-bit integer ops
  TIMED("add_u32 dep", 64, REP64("v_add_u32 v26, v26, v27\n"))
  TIMED("add_u32 x4", 64, REP16("v_add_u32 v26, v26, v30\n v_add_u32 v27, v27, v30\n v_add_u32 v28, v28, v30\n v_add_u32 v29, v29, v30\n"))
  // 17: v_mov_b32 x4
  TIMED("mov x4", 64, REP16("v_mov_b32 v26, v30\n v_mov_b32 v27, v31\n v_mov_b32 v28, v32\n v_mov_b32 v29, v33\n"))
  // 18: ds_swizzle x4 (independent), waited at the end of each group of 16
  TIMED("swizzle x4", 64, REP4(REP4("ds_swizzle_b32 v26, v30 offset:swizzle(BITMASK_PERM, \"00p00\")\n ds_swizzle_b32 v27, v31 offset:swizzle(BITMASK_PERM, \"00p00\")\n ds_swizzle_b32 v28, v32 offset:swizzle(BITMASK_PERM, \"00p00\")\n ds_swizzle_b32 v29, v33 offset:swizzle(BITMASK_PERM, \"00p00\")\n") "s_waitcnt lgkmcnt(0)\n"))
  // 19: swizzle dependent (latency)
  TIMED("swizzle dep", 16, REP16("ds_swizzle_b32 v26, v26 offset:swizzle(BITMASK_PERM, \"00p00\")\n s_waitcnt lgkmcnt(0)\n"))
  // 20/21: MFMA 4x4x4 x4 chains, plain and with cbsz:2
  TIMED("mfma x4", 64, REP16("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n"))
  TIMED("mfma cbsz x4", 64, REP16("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35] cbsz:2 abid:0\n v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37] cbsz:2 abid:1\n v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39] cbsz:2 abid:2\n v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41] cbsz:2 abid:3\n"))
  // 22: MFMA dependent chain
  TIMED("mfma dep", 64, REP64("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n"))
  // 23: MFMA + 1 independent v_fma per MFMA (single wave): cycles per pair
  TIMED("mfma+fma", 64, REP16("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n v_fma_f64 v[14:15], v[14:15], v[18:19], v[20:21]\n v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n v_fma_f64 v[16:17], v[16:17], v[18:19], v[20:21]\n v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n v_fma_f64 v[22:23], v[22:23], v[18:19], v[20:21]\n v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n v_fma_f64 v[24:25], v[24:25], v[18:19], v[20:21]\n"))
  // 24: MFMA + 1 v_add_u32 per MFMA
  TIMED("mfma+iadd", 64, REP16("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n v_add_u32 v26, v26, v30\n v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n v_add_u32 v27, v27, v30\n v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n v_add_u32 v28, v28, v30\n v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n v_add_u32 v29, v29, v30\n"))
  // 25: MFMA + 1 ds_read_b64 per MFMA
  TIMED("mfma+dsread", 64, REP16("v_mfma_f64_4x4x4_4b_f64 v[34:35], v[10:11], v[12:13], v[34:35]\n ds_read_b64 v[42:43], v30\n v_mfma_f64_4x4x4_4b_f64 v[36:37], v[10:11], v[12:13], v[36:37]\n ds_read_b64 v[44:45], v30\n v_mfma_f64_4x4x4_4b_f64 v[38:39], v[10:11], v[12:13], v[38:39]\n ds_read_b64 v[42:43], v30 offset:512\n v_mfma_f64_4x4x4_4b_f64 v[40:41], v[10:11], v[12:13], v[40:41]\n ds_read_b64 v[44:45], v30 offset:1024\n") "s_waitcnt lgkmcnt(0)\n")
  // 26: v_cvt_f32_f64 / v_cvt_f64_f32 x4
  TIMED("cvt_f32_f64 x4", 64, REP16("v_cvt_f32_f64 v26, v[10:11]\n v_cvt_f32_f64 v27, v[12:13]\n v_cvt_f32_f64 v28, v[14:15]\n v_cvt_f32_f64 v29, v[16:17]\n"))
  TIMED("cvt_f64_f32 x4", 64, REP16("v_cvt_f64_f32 v[10:11], v26\n v_cvt_f64_f32 v[12:13], v27\n v_cvt_f64_f32 v[14:15], v28\n v_cvt_f64_f32 v[16:17], v29\n"))
  TIMED("rsq_f32 x4", 64, REP16("v_rsq_f32 v26, v30\n v_rsq_f32 v27, v31\n v_rsq_f32 v28, v32\n v_rsq_f32 v29, v33\n"))
  out[63] = double(slot);
}

int main() {
  std::vector<double> a(64), b(64), d(7 * 64);
  srand(1);
  for (int l = 0; l < 64; ++l) { a[l] = (rand() % 17) - 8; b[l] = (rand() % 13) - 6; }
  double *da, *db, *dd, *dout;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 7 * 512); hipMalloc(&dout, 512);
  hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  k_cbsz<<<1, 64>>>(da, db, dd);
  hipMemcpy(d.data(), dd, 7 * 512, hipMemcpyDeviceToHost);
  // model: A[blk][i][k] <- lane 16k+4blk+i, B[blk][k][j] <- lane 16k+4blk+j,
  //        D[blk][i][j] -> lane 16i+4blk+j ; variant v uses A block sel(blk)
  auto model = [&](int cbsz, int abid, std::vector<double>& out) {
    out.assign(64, 0.0);
    for (int blk = 0; blk < 4; ++blk) {
      int ab = blk;
      if (cbsz == 2) ab = abid;
      if (cbsz == 1) ab = (blk & ~1) | (abid & 1);
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += a[16 * k + 4 * ab + i] * b[16 * k + 4 * blk + j];
        out[16 * i + 4 * blk + j] = s;
      }
    }
  };
  const int cb[7] = {0, 2, 2, 2, 2, 1, 1}, ab[7] = {0, 0, 1, 2, 3, 0, 1};
  for (int v = 0; v < 7; ++v) {
    std::vector<double> m;
    model(cb[v], ab[v], m);
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += (m[l] != d[v * 64 + l]);
    printf("cbsz %d abid %d: %s (%d lanes differ from the A-broadcast model)\n", cb[v], ab[v], bad ? "MISMATCH" : "ok", bad);
  }
  printf("A:"); for (int l = 0; l < 64; ++l) printf(" %g", a[l]); printf("\n");
  printf("B:"); for (int l = 0; l < 64; ++l) printf(" %g", b[l]); printf("\n");
  for (int v = 0; v < 7; ++v) { printf("D%d:", v); for (int l = 0; l < 64; ++l) printf(" %g", d[v * 64 + l]); printf("\n"); }
  {  // tick calibration: a long MFMA loop, ticks vs wall clock, 1 wave and a full chip
    unsigned long long* dt; hipMalloc(&dt, 8);
    for (int nb : {1, 2048}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      k_calib<<<nb, 256>>>(dt, 20000);
      hipDeviceSynchronize();
      hipEventRecord(e0); k_calib<<<nb, 256>>>(dt, 20000); hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long ticks; hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost);
      printf("calibration, %d blocks x 4 waves: %llu ticks in %.3f ms (kernel incl. launch) -> %.3f GHz tick rate, %.2f ticks per MFMA\n",
             nb, ticks, ms, ticks / (ms * 1e6), ticks / (20000.0 * 64));
    }
  }
  hipMemset(dout, 0, 512);
  k_time<<<1, 64>>>(dout);
  std::vector<double> t(64);
  hipMemcpy(t.data(), dout, 512, hipMemcpyDeviceToHost);
  const char* names[] = {"v_fma_f64 dep", "v_fma_f64 x4", "v_mul_f64 dep", "v_mul_f64 x4", "v_add_f64 dep", "v_add_f64 x4", "v_rsq_f64 dep", "v_rsq_f64 x4", "v_rndne_f64 dep", "v_rndne_f64 x4", "v_cvt_i32_f64 x4", "v_ldexp_f64 dep", "v_ldexp_f64 x4", "v_max_f64 x4", "v_sqrt_f64 x4", "v_add_u32 dep", "v_add_u32 x4", "v_mov_b32 x4", "ds_swizzle x4 (wait/16)", "ds_swizzle dep", "mfma 4x4x4 x4", "mfma 4x4x4 cbsz x4", "mfma 4x4x4 dep", "mfma + v_fma_f64 (per pair/2)", "mfma + v_add_u32 (per pair/2)", "mfma + ds_read_b64 (per pair/2)", "v_cvt_f32_f64 x4", "v_cvt_f64_f32 x4", "v_rsq_f32 x4"};
  printf("cycles per instruction, one wave (s_memtime ticks):\n");
  for (int i = 0; i < int(t[63]); ++i) printf("  %-34s %7.2f\n", names[i], t[i]);
  return 0;
}
