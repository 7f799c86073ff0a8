// What v_cvt_pk_u8_f32 does with a float that is NOT integral: every one of the 2^32 bit patterns through
//   (a) cvt_pk_u8(min(x, 127))          -- the candidate short form of the FFN requantisation
//   (b) clamp(rint(x), 0, 127) -> u8    -- what the kernels compute today (round to nearest even, then clamp)
// and the counts of patterns where they differ, split by where x lies. If (a) == (b) everywhere, the
// instruction rounds to nearest even and saturates at 0, and the rint + lower clamp can go.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned long long *bad, unsigned *example) {
  const unsigned long long n = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long local[4] = {0, 0, 0, 0};
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += n) {
    const float x = __uint_as_float((unsigned)i);
    const float r = __builtin_rintf(x);
    const unsigned want = (unsigned)__builtin_fminf(__builtin_fmaxf(r, 0.0f), 127.0f);
    // SLIMT_PROBE_BYTES: the conversion alone, then the upper clamp on the packed byte (what a kernel would do on
    // four bytes at once): m = byte & 0x80; byte = (byte | (m - (m >> 7))) & 0x7f
#ifdef SLIMT_PROBE_BYTES
    unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(x, 0, 0u) & 0xffu;
    {
      const unsigned m = got & 0x80u;
      got = (got | (m - (m >> 7))) & 0x7fu;
    }
#elif defined(SLIMT_PROBE_MED3)
    // one v_med3_f32 clamps both ends (a NaN operand makes it MIN3 of the others: 0), then the conversion rounds
    const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(x, 0.0f, 127.0f), 0, 0u) & 0xffu;
#else
    const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(x, 127.0f), 0, 0u) & 0xffu;
#endif
    if (want != got) {
      const int k = x != x ? 0 : x < 0.0f ? 1 : x < 127.5f ? 2 : 3;  // NaN | negative | in range | above
      if (local[k] == 0 && atomicAdd(&bad[4 + k], 1ull) == 0) {
        example[3 * k] = (unsigned)i;
        example[3 * k + 1] = want;
        example[3 * k + 2] = got;
      }
      ++local[k];
    }
  }
  for (int k = 0; k < 4; ++k)
    if (local[k]) atomicAdd(&bad[k], local[k]);
}

int main() {
  unsigned long long *bad;
  unsigned *ex;
  hipMalloc(&bad, 64);
  hipMalloc(&ex, 48);
  hipMemset(bad, 0, 64);
  hipMemset(ex, 0, 48);
  probe<<<1024, 256>>>(bad, ex);
  unsigned long long hb[8];
  unsigned he[12];
  if (hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  hipMemcpy(he, ex, 48, hipMemcpyDeviceToHost);
  const char *names[4] = {"NaN", "negative", "0 <= x < 127.5", "x >= 127.5"};
  for (int k = 0; k < 4; ++k) {
    float f;
    __builtin_memcpy(&f, &he[3 * k], 4);
    printf("%-16s mismatches %llu", names[k], hb[k]);
    if (hb[k]) printf("   e.g. x = %.9g (0x%08x): rint+clamp %u, cvt_pk_u8(min(x,127)) %u", f, he[3 * k], he[3 * k + 1], he[3 * k + 2]);
    printf("\n");
  }
  return 0;
}
