// How many kernels run at once? N streams, one long single-workgroup kernel each
// (and a variant with 16 workgroups); each records begin / end wall clock (100 MHz).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/concurrency_probe.hip -o gpurun_out/concurrency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long ticks, unsigned long long *out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = wall_clock64();
  }
}

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 1;
  const int lds = argc > 2 ? atoi(argv[2]) : 0;
  std::vector<int> counts = {4, 8, 12, 16, 20, 24, 32, 48};
  if (argc > 3) {
    counts.clear();
    for (int i = 3; i < argc; ++i) counts.push_back(atoi(argv[i]));
  }
  for (int n : counts) {
    std::vector<hipStream_t> st(n);
    for (auto &s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *d;
    CHK(hipMalloc(&d, sizeof(unsigned long long) * 2 * wgs * n));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 2; ++rep) {  // first pass warms up queues
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(spin, dim3(wgs), dim3(1024), lds, st[i], 200000ull /* 2 ms */, d + 2 * wgs * i);
      CHK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(2 * wgs * n);
    CHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    // kernels overlapping the midpoint of the first kernel's first workgroup
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < n; ++i) {
      unsigned long long b = ~0ull, e = 0;
      for (int w = 0; w < wgs; ++w) { b = std::min(b, h[2 * (wgs * i + w)]); e = std::max(e, h[2 * (wgs * i + w) + 1]); }
      ev.push_back({b, +1});
      ev.push_back({e, -1});
    }
    std::sort(ev.begin(), ev.end());
    int cur = 0, peak = 0;
    for (auto &x : ev) { cur += x.second; peak = std::max(peak, cur); }
    const unsigned long long span = ev.back().first - ev.front().first;
    // workgroups running 1 ms after the first one began
    int resident = 0;
    const unsigned long long probe_t = ev.front().first + 100000ull;
    for (int i = 0; i < wgs * n; ++i) resident += (h[2 * i] <= probe_t && h[2 * i + 1] > probe_t);
    printf("streams %2d x %d workgroups (lds %d): peak concurrent kernels %2d, workgroups resident at +1 ms: %d of %d, span %.2f ms (one kernel: 2 ms)\n", n,
           wgs, lds, peak, resident, wgs * n, span * 1e-5);
    CHK(hipFree(d));
    for (auto &s : st) CHK(hipStreamDestroy(s));
  }
  return 0;
}
