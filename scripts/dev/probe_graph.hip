// How much of a chain of small dependent kernels is launch latency, and does a hipGraph
// shorten it?  10 kernels of one workgroup each, every one reading what the previous wrote
// (the shape of the set passes on a small grid): stream launches vs a captured graph.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_graph scripts/dev/probe_graph.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_step(double* a, int i) { if (threadIdx.x == 0) a[i + 1] = a[i] + 1.0; }
int main() {
  double* a; hipMalloc(&a, 64 * 8); hipMemset(a, 0, 64 * 8);
  hipStream_t s; hipStreamCreate(&s);
  const int K = 10, reps = 300;
  auto chain = [&]() { for (int i = 0; i < K; ++i) k_step<<<1, 256, 0, s>>>(a, i); };
  for (int r = 0; r < 20; ++r) chain();
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { chain(); hipStreamSynchronize(s); }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("stream launches: %.1f us per chain of %d (%.1f us per kernel), launch + sync\n", us, K, us / K);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("graph launch   : %.1f us per chain of %d (%.1f us per kernel), launch + sync\n", us, K, us / K);
  // one kernel doing the ten steps behind barriers, for scale
  return 0;
}
