// Test harness (not part of the product): runs the two device-side merges of the
// N-rank certified step -- k_merge_front and k_merge_argmax, csrc/sets.hip -- on
// gathered blocks of MORE than one rank, which no single-GPU run of
// sgp_grid_sets_fused_comm can produce.  tests/test_gpu_nrank_control_flow.py writes the blocks
// to stdin and compares stdout with the NumPy merges of safeopt_amd/dist.py (the ones
// the gloo tests pin against unsharded runs).
//
//   stdin : int32 world, nfront, d, G | world * nfront doubles | world * 2 doubles
//   stdout: nfront doubles (merged front block) | d doubles xc | G doubles resid[g*16]
//           | value (double) | index (int64)
//
// Built by safeopt_amd/build.py next to the library (hipcc, gfx950).
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "sets.hip"

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));      \
      return 2;                                                      \
    }                                                                \
  } while (0)

int main() {
  int hdr[4];
  if (fread(hdr, sizeof(int), 4, stdin) != 4) return 1;
  const int world = hdr[0], nfront = hdr[1], d = hdr[2], G = hdr[3];
  std::vector<double> blocks(size_t(world) * nfront), pairs(size_t(world) * 2);
  if (fread(blocks.data(), 8, blocks.size(), stdin) != blocks.size()) return 1;
  if (fread(pairs.data(), 8, pairs.size(), stdin) != pairs.size()) return 1;
  const int n_xc = 16 * d + 16 * G;           // xc | resid[G][16] (as expander_bufs)
  const int n_fl = 16 * G + 16;
  double *dall, *dres, *dxc, *dpairs, *dout;
  int32_t* dfl;
  CK(hipMalloc(&dall, blocks.size() * 8));
  CK(hipMalloc(&dres, size_t(nfront) * 8));
  CK(hipMalloc(&dxc, size_t(n_xc) * 8));
  CK(hipMalloc(&dfl, size_t(n_fl) * 4));
  CK(hipMalloc(&dpairs, pairs.size() * 8));
  CK(hipMalloc(&dout, 16));
  CK(hipMemcpy(dall, blocks.data(), blocks.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpairs, pairs.data(), pairs.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dres, 0xff, size_t(nfront) * 8));
  CK(hipMemset(dxc, 0xff, size_t(n_xc) * 8));
  CK(hipMemset(dfl, 0xff, size_t(n_fl) * 4));
  hipLaunchKernelGGL(k_merge_front, dim3(1), dim3(64), 0, 0, dall, world, nfront, d, G,
                     dres, dxc, n_xc, dfl, n_fl);
  hipLaunchKernelGGL(k_merge_argmax, dim3(1), dim3(64), 0, 0, dpairs, world, dout,
                     reinterpret_cast<int64_t*>(dout + 1));
  CK(hipDeviceSynchronize());
  std::vector<double> res(nfront), xc(n_xc), out(2);
  std::vector<int32_t> fl(n_fl);
  CK(hipMemcpy(res.data(), dres, size_t(nfront) * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(xc.data(), dxc, size_t(n_xc) * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(fl.data(), dfl, size_t(n_fl) * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(out.data(), dout, 16, hipMemcpyDeviceToHost));
  for (int32_t f : fl)
    if (f != 0) {
      fprintf(stderr, "flags not zeroed\n");
      return 3;
    }
  fwrite(res.data(), 8, size_t(nfront), stdout);
  fwrite(xc.data(), 8, size_t(d), stdout);
  for (int g = 0; g < G; ++g) fwrite(&xc[n_xc - 16 * G + 16 * g], 8, 1, stdout);
  fwrite(out.data(), 8, 2, stdout);
  return 0;
}
