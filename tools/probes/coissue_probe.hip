// Do MFMA and ordinary VALU work of DIFFERENT waves on one SIMD overlap, and how much of an epilogue that depends on
// its own wave's MFMA results do four waves per SIMD hide? One workgroup of 1024 threads per CU (16 waves, 4 per
// SIMD), every wave runs `iters` iterations of
//   M: 16 x v_mfma_i32_16x16x64_i8 (4 independent accumulator chains of 4, operands in registers)
//   V: 76 VALU instructions per iteration (cvt, mul, add, fma: an epilogue's mix)
// modes: 0 = M only, 1 = V only (on constants), 2 = waves 0-7 M only + waves 8-15 V only (2 + 2 per SIMD),
//        3 = every wave M then V on that M's results (a GEMM with its epilogue), 4 = as 3, but V works on the
//        PREVIOUS iteration's results (software-pipelined inside the wave); 8 = as 4 with the interleave (1 MFMA,
//        5 VALU) given to the scheduler by sched_group_barrier.
// Prints microseconds per mode for the same `iters`; co-execution shows as t2 ~ max(t0, t1) / 2 and
// t3, t4 ~ max(t0, t1) instead of t0 + t1.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma16(v4i (&acc)[4], const v4i &a, const v4i &b) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[c], 0, 0, 0);
}
// 20 VALU instructions per accumulator (4 values): 4 cvt, then 4 x (mul, add, fma, max)
__device__ __forceinline__ float valu20(const v4i &c, float u, float p) {
  float f[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) f[r] = (float)c[r];
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = f[r] * u;
    t = t + p;
    t = __builtin_fmaf(t, u, p);
    s = __builtin_fmaxf(s, t);
  }
  return s;
}

// mode 5: every wave, per MFMA three independent VALU instructions right behind it (the "shadow" of a 4-pass MFMA:
// 16 cycles of matrix pipe for 4 cycles of issue); 48 VALU per iteration.
// modes 6, 7: modes 0 and 2 with v_mfma_i32_32x32x32_i8 (8 per iteration: the same MACs, half the instructions).
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(1024) void probe2(int iters, float u, float p, float *out, int seed) {
  const int wave = threadIdx.x >> 6;
  v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed, seed + 1, seed + 2, seed + 3};
  float sink = 0.0f;
  int keep = 0;
  if constexpr (MODE == 5) {
    v4i acc[4];
    float f[12];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = v4i{c, c + 1, c + 2, c + 3};
#pragma unroll
    for (int i = 0; i < 12; ++i) f[i] = (float)(seed + i) + u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[c], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            float &t = f[(3 * c + j) % 12];
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(u), "v"(p));
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) sink += f[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) keep += acc[c][0] + acc[c][3];
  } else {
    v16i acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = c + r;
    const bool do_m = MODE == 6 || wave < 8, do_v = MODE == 7 && wave >= 8;
    v4i va[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) va[c] = v4i{c, c + 1, c + 2, c + 3};
    for (int it = 0; it < iters; ++it) {
      if (do_m) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sink += valu20(va[c], u, p);
          va[c] += v4i{it, it + 1, it + 2, it + 3};
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) keep += acc[c][0] + acc[c][15];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = sink + (float)keep;
}

template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, float u, float p, float *out, int seed) {
  const int wave = threadIdx.x >> 6;
  v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed, seed + 1, seed + 2, seed + 3};
  v4i acc[4], prev[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = prev[c] = v4i{c, c + 1, c + 2, c + 3};
  float sink = 0.0f;
  const bool do_m = MODE == 0 || MODE >= 3 || (MODE == 2 && wave < 8);
  const bool do_v = MODE == 1 || MODE >= 3 || (MODE == 2 && wave >= 8);
  constexpr bool PREV = MODE == 4 || MODE == 8;  // mode 8: mode 4 + the interleave spelled out for the scheduler
  for (int it = 0; it < iters; ++it) {
    if (do_m) mfma16(acc, a, b);
    if (do_v) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const v4i &src = PREV ? prev[c] : acc[c];
        sink += valu20(src, u, p);
      }
    }
    if (PREV) {
#pragma unroll
      for (int c = 0; c < 4; ++c) prev[c] = acc[c];
    }
    if (MODE == 8) {  // one MFMA, then five VALU instructions in its shadow, sixteen times
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
    }
    if (MODE == 1 || (MODE == 2 && wave >= 8)) {  // keep the VALU-only work from being hoisted
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += v4i{it, it + 1, it + 2, it + 3};  // (+ 16 integer adds per iteration)
    }
  }
  int keep = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) keep += acc[c][0] + acc[c][3];
  out[blockIdx.x * 1024 + threadIdx.x] = sink + (float)keep;
}

template <int MODE>
static float run(int grid, int iters, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if constexpr (MODE >= 5 && MODE <= 7) {
    probe2<MODE><<<grid, 1024>>>(iters / 8, 1.0001f, 0.5f, out, 3);  // warm-up
    hipEventRecord(e0);
    probe2<MODE><<<grid, 1024>>>(iters, 1.0001f, 0.5f, out, 3);
  } else {
    probe<MODE><<<grid, 1024>>>(iters / 8, 1.0001f, 0.5f, out, 3);  // warm-up
    hipEventRecord(e0);
    probe<MODE><<<grid, 1024>>>(iters, 1.0001f, 0.5f, out, 3);
  }
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  float *out;
  if (hipMalloc(&out, (size_t)grid * 1024 * 4) != hipSuccess) return 1;
  const float t0 = run<0>(grid, iters, out), t1 = run<1>(grid, iters, out), t2 = run<2>(grid, iters, out);
  const float t3 = run<3>(grid, iters, out), t4 = run<4>(grid, iters, out);
  const float t5 = run<5>(grid, iters, out), t6 = run<6>(grid, iters, out), t7 = run<7>(grid, iters, out), t8 = run<8>(grid, iters, out);
  const double mf = 16.0 * iters, vi = 76.0 * iters;
  printf("{\"iters\": %d, \"workgroups\": %d, \"mfma_only_us\": %.1f, \"valu_only_us\": %.1f, \"half_mfma_half_valu_waves_us\": %.1f, "
         "\"mfma_then_own_epilogue_us\": %.1f, \"mfma_and_previous_epilogue_us\": %.1f, \"mfma_with_3_valu_in_its_shadow_us\": %.1f, "
         "\"mfma32_only_us\": %.1f, \"half_mfma32_half_valu_waves_us\": %.1f, \"mfma_and_previous_epilogue_interleaved_us\": %.1f, "
         "\"cycles_per_mfma_per_simd_at_2p4GHz\": %.1f, \"cycles_per_valu_per_simd_at_2p4GHz\": %.2f}\n",
         iters, grid, t0, t1, t2, t3, t4, t5, t6, t7, t8, t0 * 2400.0 / (mf * 4), t1 * 2400.0 / (vi * 4));
  return 0;
}
