// Where are L2 misses served from -- Infinity Cache (MALL, 256 MB, memory side) or HBM?
// rocprofv3 has no MALL hit counter on gfx950; what it has is the average latency of the
// L2's read requests to the fabric: TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum (requests
// in flight, summed per cycle, over requests = cycles per request).  Two reference
// kernels pin the scale:
//   k_lat_hbm : 384 MB that nothing has touched since 5 GB of other traffic went by --
//               every request goes to HBM
//   k_lat_mall: the SAME 96 MB over and over -- far beyond the 8 x 4 MB of L2, well
//               inside the 256 MB Infinity Cache
// and the sweep kernels' own ratio (scripts/collect_profiles.sh, pmc_ealat_cfg*) is read
// against them.   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_mall scripts/dev/probe_mall.hip
#include <hip/hip_runtime.h>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#include <cstdio>
#include <cstdlib>

// one load in flight per wave (the next one is issued when the value is there): the
// fabric sees no queueing, the ratio is the latency of the level that serves the request
__global__ __launch_bounds__(64) void k_lat_hbm(const dbl2* p, size_t n, double* out) {
  double acc = 0.0;
  for (size_t i = size_t(blockIdx.x) * 64 + threadIdx.x; i < n; i += size_t(gridDim.x) * 64) {
    const dbl2 v = p[i];
    acc += v.x + v.y;
    asm volatile("" : "+v"(acc));
  }
  if (acc == 1.2345e-300) out[0] = acc;
}
__global__ __launch_bounds__(64) void k_lat_mall(const dbl2* p, size_t n, int passes,
                                                 double* out) {
  double acc = 0.0;
  for (int r = 0; r < passes; ++r)
    for (size_t i = size_t(blockIdx.x) * 64 + threadIdx.x; i < n; i += size_t(gridDim.x) * 64) {
      const dbl2 v = p[i];
      acc += v.x + v.y;
      asm volatile("" : "+v"(acc));
    }
  if (acc == 1.2345e-300) out[0] = acc;
}

int main() {
  const size_t big = size_t(6) << 30, small = size_t(96) << 20;
  dbl2 *a, *b; double* out;
  if (hipMalloc(&a, big) != hipSuccess || hipMalloc(&b, small) != hipSuccess) return 1;
  hipMalloc(&out, 8);
  hipMemset(a, 0, big); hipMemset(b, 0, small);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    // HBM: the first 384 MB of the 6 GB buffer (the memset behind them pushed them out
    // of every cache long ago; again before the second repetition)
    hipMemset(a + (size_t(1) << 30) / 16, 0, big - (size_t(1) << 30));
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_lat_hbm<<<256, 64>>>(a, (size_t(384) << 20) / 16, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("k_lat_hbm : %.3f ms, %.1f GB/s\n", ms, double(size_t(384) << 20) / ms / 1e6);
    // Infinity Cache: 96 MB (3 x the 8 L2s together), a warming pass, then 4 timed ones
    k_lat_mall<<<256, 64>>>(b, small / 16, 1, out);
    hipEventRecord(e0);
    k_lat_mall<<<256, 64>>>(b, small / 16, 4, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("k_lat_mall: %.3f ms, %.1f GB/s\n", ms, 4.0 * small / ms / 1e6);
  }
  return 0;
}
