"""GPU box: is a chain of v_mfma_f32_32x32x2_f32 (k ascending) bit-identical to the
k-ascending fmaf chain the attention kernels use? Builds a tiny probe with hipcc
and compares on random data of mixed magnitudes (incl. denormals)."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
SRC = r'''
#include <hip/hip_runtime.h>
typedef float v16f __attribute__((ext_vector_type(16)));
// A [32][K] row-major, B [K][32] row-major -> C [32][32]; K = 32
extern "C" __global__ void mfma_chain(const float* A, const float* B, float* Cm, float* Cv) {
  const int lane = threadIdx.x;  // one wave
  v16f acc = {0};
  for (int k0 = 0; k0 < 32; k0 += 2) {
    // 32x32x2: lane l holds A[m = l % 32][k = k0 + l / 32], B[k = k0 + l / 32][n = l % 32]
    const float a = A[(lane & 31) * 32 + k0 + (lane >> 5)];
    const float b = B[(k0 + (lane >> 5)) * 32 + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // C layout: lane l, reg r: n = l % 32, m = 8 * (r / 4) + 4 * (l / 32) + r % 4
  for (int r = 0; r < 16; ++r) {
    const int n = lane & 31, m = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
    Cm[m * 32 + n] = acc[r];
  }
  for (int idx = lane; idx < 1024; idx += 64) {
    const int m = idx >> 5, n = idx & 31;
    float s = 0.0f;
    for (int k = 0; k < 32; ++k) s = __builtin_fmaf(A[m * 32 + k], B[k * 32 + n], s);
    Cv[idx] = s;
  }
}
typedef float v4f __attribute__((ext_vector_type(4)));
// same product through v_mfma_f32_16x16x4_f32 (k ascending), top-left 16 x 16 of C only
extern "C" __global__ void mfma_chain16(const float* A, const float* B, float* Cm) {
  const int lane = threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < 32; k0 += 4) {
    // 16x16x4: lane l holds A[m = l % 16][k = k0 + l / 16], B[k = k0 + l / 16][n = l % 16]
    const float a = A[(lane & 15) * 32 + k0 + (lane >> 4)];
    const float b = B[(k0 + (lane >> 4)) * 32 + (lane & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  // C layout: lane l, reg r: n = l % 16, m = 4 * (l / 16) + r
  for (int r = 0; r < 4; ++r) Cm[(4 * (lane >> 4) + r) * 32 + (lane & 15)] = acc[r];
}
'''
MAIN = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
extern "C" __global__ void mfma_chain(const float*, const float*, float*, float*);
extern "C" __global__ void mfma_chain16(const float*, const float*, float*);
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); int n; fread(&n, 4, 1, f);
  float *hA = (float*)malloc(n * 4096), *hB = (float*)malloc(n * 4096);
  fread(hA, 4096, n, f); fread(hB, 4096, n, f); fclose(f);
  float *A, *B, *Cm, *Cv; hipMalloc(&A, 4096); hipMalloc(&B, 4096); hipMalloc(&Cm, 4096); hipMalloc(&Cv, 4096);
  float hm[1024], hv[1024], h16[1024]; long diff = 0, total = 0, diff16 = 0, total16 = 0;
  for (int i = 0; i < n; ++i) {
    hipMemcpy(A, hA + i * 1024, 4096, hipMemcpyHostToDevice); hipMemcpy(B, hB + i * 1024, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, A, B, Cm, Cv);
    hipMemcpy(hm, Cm, 4096, hipMemcpyDeviceToHost); hipMemcpy(hv, Cv, 4096, hipMemcpyDeviceToHost);
    for (int j = 0; j < 1024; ++j) { total++; if (memcmp(&hm[j], &hv[j], 4)) { diff++; if (diff <= 5) printf("case %d elem %d mfma %a valu %a\n", i, j, hm[j], hv[j]); } }
    hipLaunchKernelGGL(mfma_chain16, dim3(1), dim3(64), 0, 0, A, B, Cm);
    hipMemcpy(h16, Cm, 4096, hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { const int j = m * 32 + n; total16++;
      if (memcmp(&h16[j], &hv[j], 4)) { diff16++; if (diff16 <= 5) printf("16x16x4 case %d elem %d mfma %a valu %a\n", i, j, h16[j], hv[j]); } }
  }
  printf("32x32x2: elements %ld differing %ld\n", total, diff);
  printf("16x16x4: elements %ld differing %ld\n", total16, diff16);
  return 0;
}
'''
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "k.hip"), "w").write(SRC)
    open(os.path.join(d, "m.hip"), "w").write(MAIN)
    exe = os.path.join(d, "probe")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-ffp-contract=off",
                           os.path.join(d, "k.hip"), os.path.join(d, "m.hip"), "-o", exe])
    r = np.random.Generator(np.random.PCG64(1))
    cases = []
    for i in range(64):
        scale = [1.0, 1e-3, 1e3, 1e-20, 1e18, 1e-38][i % 6]
        A = (r.normal(0, 1, (32, 32)) * scale).astype(np.float32)
        B = (r.normal(0, 1, (32, 32)) * (1.0 if i % 2 else scale)).astype(np.float32)
        if i % 7 == 0:
            A[r.random((32, 32)) < 0.3] = 0
        cases.append((A, B))
    dat = os.path.join(d, "in.bin")
    with open(dat, "wb") as f:
        f.write(np.int32(len(cases)).tobytes())
        for A, _ in cases: f.write(A.tobytes())
        for _, B in cases: f.write(B.tobytes())
    print(subprocess.run([exe, dat], capture_output=True, text=True, timeout=120).stdout)
