// Is the shared-reciprocal quotient bit-identical to the compiler's correctly rounded
// f32 division inside a guard range (default |n|, d in [2^-40, 2^40], n may also be +0;
// argv: iters [n_exp_lo n_exp_hi d_exp_lo d_exp_hi] as powers of two)?
// The compiler expands n / d into v_div_scale x2, v_rcp, 2 fma (reciprocal refinement),
// mul + 3 fma (quotient refinement), v_div_fmas, v_div_fixup; inside the guard range the
// scale / fixup instructions are the identity, so the same fma chain with the reciprocal
// computed once per denominator must give the same bits. Counts mismatches over
// blocks * threads * iters pseudo-random pairs (plus boundary exponents).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float shared_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float shared_div(float n, float d, float r1) {
  const float q0 = n * r1;
  const float m0 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(m0, r1, q0);
  const float m1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(m1, r1, q1);
}

__device__ __forceinline__ unsigned mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (unsigned)x;
}

__global__ void probe(unsigned long long seed, int iters, unsigned long long *bad, float *example, int en_lo, int en_n,
                      int ed_lo, int ed_n) {
  const unsigned long long id = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long local = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned a = mix(seed + id * 0x9e3779b97f4a7c15ull + (unsigned long long)it * 0x632be59bd9b4e019ull);
    const unsigned b = mix(seed * 31 + id * 0xd1b54a32d192ed03ull + (unsigned long long)it * 0x2545f4914f6cdd1dull);
    // exponent in [127-40, 127+40], random mantissa; n: random sign, 1 in 64 exactly +0
    const unsigned ed = (unsigned)ed_lo + (a >> 8) % (unsigned)ed_n, en = (unsigned)en_lo + (b >> 8) % (unsigned)en_n;
    const float d = __uint_as_float((ed << 23) | (a & 0x7fffff) * ((a >> 31) ? 1u : 1u));
    float n = __uint_as_float(((b >> 30) & 1u) << 31 | (en << 23) | (b & 0x7fffff));
    if ((b & 0x3f000000u) == 0) n = 0.0f;
    const float want = n / d;
    const float got = shared_div(n, d, shared_rcp(d));
    if (__float_as_uint(want) != __float_as_uint(got)) {
      if (local == 0) { example[0] = n; example[1] = d; example[2] = want; example[3] = got; }
      ++local;
    }
  }
  if (local) atomicAdd(bad, local);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  const int nlo = argc > 5 ? atoi(argv[2]) : -40, nhi = argc > 5 ? atoi(argv[3]) : 40;
  const int dlo = argc > 5 ? atoi(argv[4]) : -40, dhi = argc > 5 ? atoi(argv[5]) : 40;
  unsigned long long *bad;
  float *ex;
  hipMalloc(&bad, 8);
  hipMalloc(&ex, 16);
  hipMemset(bad, 0, 8);
  hipMemset(ex, 0, 16);
  const int blocks = 4096, threads = 256;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull, iters, bad, ex, 127 + nlo, nhi - nlo + 1,
                     127 + dlo, dhi - dlo + 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  unsigned long long hb = 0;
  float he[4];
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
  printf("|n| in [2^%d, 2^%d] or +0, d in [2^%d, 2^%d]: pairs %.3g, mismatches %llu", nlo, nhi, dlo, dhi,
         (double)blocks * threads * iters, hb);
  if (hb) printf("  e.g. n=%a d=%a want=%a got=%a", he[0], he[1], he[2], he[3]);
  printf("\n");
  return hb != 0;
}
