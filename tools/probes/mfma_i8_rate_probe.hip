// Measured-sustained int8 MFMA rate of the chip ("report the fraction against
// both the nominal and the measured-sustained peak"). Register-resident operands, no memory
// traffic in the loop: every wave issues independent v_mfma_i32_16x16x64_i8 (or 32x32x32)
// chains back to back, on random operands (the chip clocks lower on random data than on
// zeros, MI355X_MICROARCH.md "DVFS give-back"). Sweeps waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_i8_rate_probe.hip -o gpurun_out/mfma_i8_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void mfma16(const int *seed, int iters, int *sink, unsigned long long *stamps) {
  const int s = seed[threadIdx.x & 63];
  v4i a = {s, s * 3 + 1, s * 5 + 2, s * 7 + 3}, b = {s ^ 0x55aa55aa, s * 11, s * 13 + 5, s * 17 + 7};
  v4i c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = v4i{i, 0, 0, 0};
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
  }
  const unsigned long long t1 = wall_clock64();
  int x = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) x += c[i].x + c[i].y + c[i].z + c[i].w;
  if (x == 0x7fffffff) sink[0] = x;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

__global__ void mfma32(const int *seed, int iters, int *sink, unsigned long long *stamps) {
  const int s = seed[threadIdx.x & 63];
  v4i a = {s, s * 3 + 1, s * 5 + 2, s * 7 + 3}, b = {s ^ 0x55aa55aa, s * 11, s * 13 + 5, s * 17 + 7};
  v16i c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c[i][j] = i + j;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
  }
  const unsigned long long t1 = wall_clock64();
  int x = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) x += c[i][j];
  if (x == 0x7fffffff) sink[0] = x;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

int main() {
  int *d_seed, *d_sink;
  unsigned long long *d_st;
  std::vector<int> seed(64);
  srand(7);
  for (auto &v : seed) v = rand() * 2654435761u;
  CHK(hipMalloc(&d_seed, 256));
  CHK(hipMemcpy(d_seed, seed.data(), 256, hipMemcpyHostToDevice));
  CHK(hipMalloc(&d_sink, 4));
  CHK(hipMalloc(&d_st, 16 * 4096));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  for (int shape = 0; shape < 2; ++shape) {
    for (int waves : {4, 8, 16}) {
      const int iters = 400000 / waves;
      const int per_it = shape == 0 ? 8 : 4;
      const double ops_per_mfma = shape == 0 ? 2.0 * 16 * 16 * 64 : 2.0 * 32 * 32 * 32;
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0));
      CHK(hipEventCreate(&e1));
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {  // the last repetition is the one reported (clock settled)
        CHK(hipEventRecord(e0, 0));
        if (shape == 0)
          hipLaunchKernelGGL(mfma16, dim3(cus), dim3(64 * waves), 0, 0, d_seed, iters, d_sink, d_st);
        else
          hipLaunchKernelGGL(mfma32, dim3(cus), dim3(64 * waves), 0, 0, d_seed, iters, d_sink, d_st);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
      }
      std::vector<unsigned long long> h(2 * cus);
      CHK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
      double in_kernel_us = 0;
      for (int i = 0; i < cus; ++i) in_kernel_us += (h[2 * i + 1] - h[2 * i]) * 0.01 / cus;
      const double total_ops = (double)cus * waves * iters * per_it * ops_per_mfma;
      printf("{\"probe\":\"mfma_i8_%s\",\"cus\":%d,\"waves_per_cu\":%d,\"ms\":%.3f,\"chip_TOPs\":%.0f,"
             "\"chip_TOPs_in_kernel\":%.0f,\"cycles_per_mfma_per_simd_at_2.4GHz\":%.2f}\n",
             shape == 0 ? "16x16x64" : "32x32x32", cus, waves, ms, total_ops / (ms * 1e-3) / 1e12,
             total_ops / (in_kernel_us * 1e-6) / 1e12,
             in_kernel_us * 1e-6 * 2.4e9 / ((double)iters * per_it * (waves / 4.0)));
      fflush(stdout);
    }
  }
  return 0;
}
