// Exhaustive check (every one of the 2^32 float bit patterns) that the short forms the fused kernels use produce the same
// bits as the forms they replaced (device_common.h):
//   (0) quantize1_byte(x, aq) & 0xff        vs  quantize1(x, aq) & 0xff            for several multipliers aq (and x = the PRODUCT's operand)
//   (1) relu-in-the-clamp byte  med3(v aq, 0, 127) + magic   vs  quantize1(max(v, 0), aq) & 0xff
//   (2) exp_p_select(x)                      vs  exp_p(x)
//   (3) sigmoid_p_select(x)                  vs  sigmoid_p(x)
// A NaN result matches a NaN result (payloads are not compared: nothing downstream reads them).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I slimt_amd/csrc tools/probes/select_forms_probe.hip -o /tmp/select_forms_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "device_common.h"

using namespace slimt_hip;

__device__ __forceinline__ bool same_float(float a, float b) {
  return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b);
}

__global__ void probe(unsigned long long *bad, unsigned *example, const float *aqs, int n_aq) {
  const unsigned long long n = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long local[4] = {0, 0, 0, 0};
  auto note = [&](int k, unsigned i, unsigned want, unsigned got) {
    if (local[k] == 0 && atomicAdd(&bad[4 + k], 1ull) == 0) {
      example[3 * k] = i;
      example[3 * k + 1] = want;
      example[3 * k + 2] = got;
    }
    ++local[k];
  };
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += n) {
    const float x = __uint_as_float((unsigned)i);
    for (int a = 0; a < n_aq; ++a) {
      const float aq = aqs[a];
      const unsigned w0 = (unsigned)quantize1(x, aq) & 0xffu, g0 = (unsigned)quantize1_byte(x, aq) & 0xffu;
      if (w0 != g0) note(0, (unsigned)i, w0, g0);
      const float relu = x > 0.0f ? x : 0.0f;
      const unsigned w1 = (unsigned)quantize1(relu, aq) & 0xffu;
      const unsigned g1 = __float_as_uint(__builtin_amdgcn_fmed3f(x * aq, 0.0f, 127.0f) + 12582912.0f) & 0xffu;
      if (w1 != g1) note(1, (unsigned)i, w1, g1);
    }
    const float e0 = exp_p(x), e1 = exp_p_select(x);
    if (!same_float(e0, e1)) note(2, (unsigned)i, __float_as_uint(e0), __float_as_uint(e1));
    const float s0 = sigmoid_p(x), s1 = sigmoid_p_select(x);
    if (!same_float(s0, s1)) note(3, (unsigned)i, __float_as_uint(s0), __float_as_uint(s1));
  }
  for (int k = 0; k < 4; ++k)
    if (local[k]) atomicAdd(&bad[k], local[k]);
}

int main() {
  // multipliers: 127 / max|x| of typical activations, 1, tiny, huge, and one that makes halves (ties) common
  const float h_aq[6] = {127.0f / 6.0f, 1.0f, 0.5f, 3.0e-5f, 41234.5f, 127.0f / 0.37f};
  float *aqs;
  unsigned long long *bad;
  unsigned *ex;
  hipMalloc(&aqs, sizeof(h_aq));
  hipMemcpy(aqs, h_aq, sizeof(h_aq), hipMemcpyHostToDevice);
  hipMalloc(&bad, 64);
  hipMalloc(&ex, 48);
  hipMemset(bad, 0, 64);
  hipMemset(ex, 0, 48);
  probe<<<2048, 256>>>(bad, ex, aqs, 6);
  unsigned long long hb[8];
  unsigned he[12];
  if (hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  hipMemcpy(he, ex, 48, hipMemcpyDeviceToHost);
  const char *names[4] = {"quantize1_byte", "relu in the clamp", "exp_p_select", "sigmoid_p_select"};
  int rc = 0;
  for (int k = 0; k < 4; ++k) {
    float f;
    __builtin_memcpy(&f, &he[3 * k], 4);
    printf("%-18s mismatches %llu of 2^32 inputs%s", names[k], hb[k], k < 2 ? " x 6 multipliers" : "");
    if (hb[k]) {
      printf("   e.g. x = %.9g (0x%08x): want 0x%x, got 0x%x", f, he[3 * k], he[3 * k + 1], he[3 * k + 2]);
      rc = 2;
    }
    printf("\n");
  }
  return rc;
}
