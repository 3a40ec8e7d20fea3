#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* o1, double* o2, double* o0, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = fmax(x[i], 1e-300);
  double y = __builtin_amdgcn_rsq(v);
  o0[i] = y;
  double g = v * y, h = 0.5 * y, r;
  r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h);
  // variant 1: one iteration + correction
  double r1 = fma(-g, g, v);
  o1[i] = fma(r1, h, g);
  // variant 2: two iterations + correction
  r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-g, g, v);
  o2[i] = fma(r, h, g);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> e(-60, 40);
  for (int i = 0; i < n; ++i) x[i] = std::exp2(e(rng)) * (1.0 + (rng() % 1000) * 1e-3);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d1, d2, d0, n);
  std::vector<double> o0(n), o1(n), o2(n);
  hipMemcpy(o0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(o1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(o2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) {
    double s = std::sqrt(x[i]);
    m0 = std::fmax(m0, std::fabs(o0[i] * s - 1.0));
    m1 = std::fmax(m1, std::fabs(o1[i] - s) / s);
    m2 = std::fmax(m2, std::fabs(o2[i] - s) / s);
  }
  printf("rsq rel err %.3g ; 1 iter + corr %.3g ; 2 iter + corr %.3g (eps %.3g)\n", m0, m1, m2, 2.2e-16);
  return 0;
}
