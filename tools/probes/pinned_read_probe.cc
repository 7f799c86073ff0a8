// GPU box: how fast does the CPU read pinned host memory a kernel has written?
// hipHostMalloc default (coherent) / NonCoherent / hipHostRegister'ed malloc, 1.5 MB (one batch's
// alignment rows). Build: hipcc -O2 tools/probes/pinned_read_probe.cc -o /tmp/pinned_read_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void fill(float *p, size_t n, float v) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)i;
}
static double ms_since(std::chrono::steady_clock::time_point t) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
}
int main() {
  const size_t n = 256 * 48 * 32;
  std::vector<float> dst(n);
  struct Case { const char *name; unsigned flags; int mode; } cases[] = {
      {"hipHostMalloc default", hipHostMallocDefault, 0},
      {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent, 0},
      {"hipHostMalloc Coherent", hipHostMallocCoherent, 0},
      {"malloc + hipHostRegister", 0, 1}};
  for (auto &c : cases) {
    float *h = nullptr, *d = nullptr;
    if (c.mode == 0) {
      if (hipHostMalloc((void **)&h, n * 4, c.flags) != hipSuccess) { std::printf("%s: alloc failed\n", c.name); continue; }
      d = h;
      void *dv = nullptr;
      if (hipHostGetDevicePointer(&dv, h, 0) == hipSuccess) d = (float *)dv;
    } else {
      h = (float *)aligned_alloc(4096, n * 4);
      if (hipHostRegister(h, n * 4, hipHostRegisterDefault) != hipSuccess) { std::printf("%s: register failed\n", c.name); continue; }
      void *dv = nullptr;
      hipHostGetDevicePointer(&dv, h, 0);
      d = (float *)dv;
    }
    double best_read = 1e9, best_sum = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, d, n, (float)rep);
      hipDeviceSynchronize();
      auto t = std::chrono::steady_clock::now();
      std::memcpy(dst.data(), h, n * 4);
      best_read = std::min(best_read, ms_since(t));
      hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, d, n, (float)rep + 0.5f);
      hipDeviceSynchronize();
      t = std::chrono::steady_clock::now();
      double s = 0;
      for (size_t i = 0; i < n; i += 8) s += h[i];  // strided small reads, like vector(row, row + len)
      best_sum = std::min(best_sum, ms_since(t));
      if (dst[5] != (float)rep + 5.0f || s == 0) std::printf("  (value check: %f)\n", dst[5]);
    }
    std::printf("%-28s memcpy of %.1f MB %.3f ms (%.2f GB/s), strided read %.3f ms\n", c.name, n * 4 / 1e6, best_read,
                n * 4 / 1e6 / best_read, best_sum);
  }
}
