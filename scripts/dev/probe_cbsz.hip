// Decode what cbsz does on v_mfma_f64_4x4x4_4b_f64: D[l] = sum_{p,q} T[l,p,q] A[p] B[q].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int CB, int AB>
__global__ void k(double* out) {
  const int l = threadIdx.x;
  for (int p = 0; p < 64; ++p)
    for (int h = 0; h < 2; ++h) {
      const double a = (l == p) ? 1.0 : 0.0;
      const double b = ((l >> 5) == h) ? double(1ull << (l & 31)) : 0.0;
      out[(p * 2 + h) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CB, AB, 0);
    }
}
template <int CB, int AB>
void run(double* d) {
  k<CB, AB><<<1, 64>>>(d);
  std::vector<double> h(64 * 2 * 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  printf("cbsz %d abid %d: output lane <- list of (A lane, B lane)\n", CB, AB);
  for (int l = 0; l < 64; ++l) {
    printf("  D[%2d]:", l);
    for (int p = 0; p < 64; ++p)
      for (int hh = 0; hh < 2; ++hh) {
        const unsigned long long bits = (unsigned long long)h[(p * 2 + hh) * 64 + l];
        for (int q = 0; q < 32; ++q)
          if (bits >> q & 1) printf(" (%d,%d)", p, hh * 32 + q);
      }
    printf("\n");
  }
}
int main() {
  double* d; hipMalloc(&d, 64 * 2 * 64 * 8);
  run<0, 0>(d); run<2, 0>(d); run<2, 1>(d); run<1, 0>(d);
  return 0;
}
