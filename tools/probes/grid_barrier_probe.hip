// What would a grid-wide barrier cost a resident, weight-stationary decoder?
// One workgroup per CU (cooperative launch: the runtime refuses the launch if they cannot all be
// resident), K barriers in a row, three forms:
//   flat      every workgroup adds to ONE device-scope counter and polls it
//   twolevel  workgroups of an XCD meet on their XCD's counter, one of them carries the XCD to a
//             device counter and releases its XCD through a per-XCD flag (polls stay inside an L2)
//   payload   flat, with a 4 KB exchange per workgroup through global memory between barriers
//             (write own slot, barrier, read a neighbour's): what a phase of activations costs
// Every poll loop is BOUNDED (gives up after ~20 ms and flags the run as failed) so that a
// mistake cannot hang the device.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o gpurun_out/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

constexpr int kMaxSpins = 400000;  // x ~50 ns of s_sleep: ~20 ms

__device__ __forceinline__ bool wait_ge(const unsigned *p, unsigned target, unsigned *fail) {
  for (int i = 0; i < kMaxSpins; ++i) {
    if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  atomicAdd(fail, 1u);
  return false;
}

__global__ void flat_barriers(unsigned *counter, unsigned *fail, int iters, unsigned long long *stamps,
                              float *payload, int payload_floats) {
  const unsigned n = gridDim.x;
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  float acc = 0.0f;
  for (int k = 0; k < iters && ok; ++k) {
    if (payload_floats) {  // this phase's output of this workgroup
      for (int i = threadIdx.x; i < payload_floats; i += blockDim.x)
        payload[(size_t)blockIdx.x * payload_floats + i] = (float)(k + i);
      __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (!wait_ge(counter, (unsigned)(k + 1) * n, fail)) ok = 0;
    }
    __syncthreads();
    if (payload_floats) {  // the next phase reads what another workgroup wrote
      const unsigned src = (blockIdx.x + 37) % n;
      for (int i = threadIdx.x; i < payload_floats; i += blockDim.x)
        acc += __builtin_nontemporal_load(payload + (size_t)src * payload_floats + i);
      __syncthreads();  // (all reads done before the slot is rewritten two barriers later: one slot per phase parity would be the real thing)
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
  if (acc == 1.2345f) payload[0] = acc;
}

// counters: [0..7] per-XCD arrival, [8] device arrival, [16..23] per-XCD release generation
__global__ void twolevel_barriers(unsigned *c, unsigned *fail, int iters, unsigned long long *stamps,
                                  const unsigned *xcd_size) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;  // HW_REG_XCC_ID
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  unsigned nx = 0;  // XCDs that hold workgroups
  for (int i = 0; i < 8; ++i) nx += xcd_size[i] ? 1u : 0u;
  const unsigned mine = xcd_size[xcc];
  const unsigned long long t0 = wall_clock64();
  for (int k = 0; k < iters && ok; ++k) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned prev = __hip_atomic_fetch_add(c + xcc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1 == (unsigned)(k + 1) * mine) {  // last of this XCD: carry it to the device level
        __hip_atomic_fetch_add(c + 8, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_ge(c + 8, (unsigned)(k + 1) * nx, fail)) ok = 0;
        __hip_atomic_store(c + 16 + xcc, (unsigned)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else if (!wait_ge(c + 16 + xcc, (unsigned)(k + 1), fail)) {
        ok = 0;
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

__global__ void count_xcds(unsigned *xcd_size) {
  if (threadIdx.x == 0) atomicAdd(xcd_size + (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15), 1u);
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int iters = 2000;
  unsigned *counters, *fail, *xcd;
  unsigned long long *stamps;
  float *payload;
  CHK(hipMalloc(&counters, 64 * sizeof(unsigned)));
  CHK(hipMalloc(&fail, sizeof(unsigned)));
  CHK(hipMalloc(&xcd, 16 * sizeof(unsigned)));
  CHK(hipMalloc(&stamps, 2 * cus * sizeof(unsigned long long)));
  CHK(hipMalloc(&payload, (size_t)cus * 4096 * sizeof(float)));
  std::vector<unsigned long long> h(2 * cus);
  for (int threads : {64, 1024}) {
    for (int grid : {cus / 8, cus / 2, cus}) {
      // flat, without and with a payload
      for (int pf : {0, 1024, 4096}) {
        CHK(hipMemset(counters, 0, 64 * sizeof(unsigned)));
        CHK(hipMemset(fail, 0, sizeof(unsigned)));
        int it = iters;
        int pfl = pf;
        void *args[] = {&counters, &fail, &it, &stamps, &payload, &pfl};
        CHK(hipLaunchCooperativeKernel((void *)flat_barriers, dim3(grid), dim3(threads), args, 0, 0));
        CHK(hipDeviceSynchronize());
        unsigned f = 0;
        CHK(hipMemcpy(&f, fail, sizeof(f), hipMemcpyDeviceToHost));
        CHK(hipMemcpy(h.data(), stamps, 2 * grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double us = 0;
        for (int b = 0; b < grid; ++b) us += (double)(h[2 * b + 1] - h[2 * b]) / 100.0;
        printf("{\"form\": \"flat\", \"threads\": %d, \"workgroups\": %d, \"payload_bytes_per_wg\": %d, \"barriers\": %d, "
               "\"us_per_barrier\": %.3f, \"timeouts\": %u}\n", threads, grid, pf * 4, iters, us / grid / iters, f);
      }
      // two-level
      CHK(hipMemset(xcd, 0, 16 * sizeof(unsigned)));
      // the XCD population of a grid of this size (workgroups are dealt round-robin over the XCDs)
      {
        void *a0[] = {&xcd};
        CHK(hipLaunchCooperativeKernel((void *)count_xcds, dim3(grid), dim3(threads), a0, 0, 0));
        CHK(hipDeviceSynchronize());
      }
      CHK(hipMemset(counters, 0, 64 * sizeof(unsigned)));
      CHK(hipMemset(fail, 0, sizeof(unsigned)));
      int it = iters;
      void *args[] = {&counters, &fail, &it, &stamps, &xcd};
      CHK(hipLaunchCooperativeKernel((void *)twolevel_barriers, dim3(grid), dim3(threads), args, 0, 0));
      CHK(hipDeviceSynchronize());
      unsigned f = 0;
      CHK(hipMemcpy(&f, fail, sizeof(f), hipMemcpyDeviceToHost));
      CHK(hipMemcpy(h.data(), stamps, 2 * grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double us = 0;
      for (int b = 0; b < grid; ++b) us += (double)(h[2 * b + 1] - h[2 * b]) / 100.0;
      printf("{\"form\": \"twolevel\", \"threads\": %d, \"workgroups\": %d, \"barriers\": %d, \"us_per_barrier\": %.3f, "
             "\"timeouts\": %u}\n", threads, grid, iters, us / grid / iters, f);
    }
  }
  return 0;
}
