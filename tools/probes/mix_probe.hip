// Synthetic model of the bench's kernel mix, to study how the hardware packs it:
// W streams, each alternating an "encoder" (EG workgroups x ET us, 155 KB LDS) and a
// "decoder" (DG workgroups x DT us, 119 KB LDS) R times; both take a whole CU. Spin
// kernels (bounded by the wall clock). Prints the busy fraction of the CUs over the
// middle 60 % of the run and the rate of (encoder, decoder) pairs.
// usage: mix_probe W R EG ET_us DG DT_us [decoder stream priority: 0 same, 1 high] [encoder priority low: 1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long ticks, unsigned long long *log, unsigned long long *counter) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) {
    const unsigned long long slot = atomicAdd(counter, 1ull);
    log[2 * slot] = t0;
    log[2 * slot + 1] = wall_clock64();
  }
}

// one launch per batch: workgroups [0, eg) are the encoder's, the rest the decoder's
__global__ void spin2(int eg, unsigned long long et, unsigned long long dt, unsigned long long *log,
                      unsigned long long *counter) {
  const unsigned long long ticks = (int)blockIdx.x < eg ? et : dt;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) {
    const unsigned long long slot = atomicAdd(counter, 1ull);
    log[2 * slot] = t0;
    log[2 * slot + 1] = wall_clock64();
  }
}

// over-subscribed decoder: `over` x the workgroups are launched, the first `real` to start
// claim the work (atomic ticket), the rest exit at once
__global__ void spin3(unsigned real, unsigned *ticket, unsigned long long ticks, unsigned long long *log,
                      unsigned long long *counter) {
  __shared__ unsigned mine;
  if (threadIdx.x == 0) mine = atomicAdd(ticket, 1u);
  __syncthreads();
  if (mine >= real) return;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) {
    const unsigned long long slot = atomicAdd(counter, 1ull);
    log[2 * slot] = t0;
    log[2 * slot + 1] = wall_clock64();
  }
}

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 7) { printf("usage: mix_probe W R EG ET DG DT [dec_prio] [split]\n"); return 2; }
  const int W = atoi(argv[1]), R = atoi(argv[2]), EG = atoi(argv[3]), ET = atoi(argv[4]), DG = atoi(argv[5]),
            DT = atoi(argv[6]);
  const int dec_prio = argc > 7 ? atoi(argv[7]) : 0;
  const int split = argc > 8 ? atoi(argv[8]) : 0;  // 1: decoders on a second stream per worker (event-chained)
  const int chunks = argc > 9 ? atoi(argv[9]) : 1; // decoder as `chunks` consecutive launches of DT / chunks each
  int lo_p = 0, hi_p = 0;
  CHK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
  std::vector<hipStream_t> se(W), sd(W);
  std::vector<hipEvent_t> e1(W), e2(W);
  for (int w = 0; w < W; ++w) {
    CHK(hipStreamCreateWithPriority(&se[w], hipStreamNonBlocking, lo_p));
    if (split || dec_prio)
      CHK(hipStreamCreateWithPriority(&sd[w], hipStreamNonBlocking, dec_prio ? hi_p : lo_p));
    else
      sd[w] = se[w];
    CHK(hipEventCreateWithFlags(&e1[w], hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&e2[w], hipEventDisableTiming));
  }
  const size_t total = (size_t)W * R * (EG + DG * chunks);
  unsigned long long *log, *counter;
  CHK(hipMalloc(&log, total * 16));
  CHK(hipMalloc(&counter, 8));
  CHK(hipMemset(counter, 0, 8));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 158000));
  CHK(hipDeviceSynchronize());
  const int over = argc > 11 ? atoi(argv[11]) : 1;  // decoder launches over x DG workgroups, first DG claim the work
  unsigned *tickets = nullptr;
  CHK(hipMalloc(&tickets, sizeof(unsigned) * (size_t)W * R * chunks));
  CHK(hipMemset(tickets, 0, sizeof(unsigned) * (size_t)W * R * chunks));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin3), hipFuncAttributeMaxDynamicSharedMemorySize, 158000));
  CHK(hipDeviceSynchronize());
  const int fused = argc > 10 ? atoi(argv[10]) : 0;  // 1: one launch per batch (encoder + decoder workgroups)
  if (fused) {
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin2), hipFuncAttributeMaxDynamicSharedMemorySize, 158000));
    for (int r = 0; r < R; ++r)
      for (int w = 0; w < W; ++w)
        hipLaunchKernelGGL(spin2, dim3(EG + DG), dim3(1024), 158000, se[w], EG, (unsigned long long)ET * 100,
                           (unsigned long long)DT * 100, log, counter);
  } else
  for (int r = 0; r < R; ++r)
    for (int w = 0; w < W; ++w) {
      if (sd[w] != se[w] && r) CHK(hipStreamWaitEvent(se[w], e2[w], 0));  // next encoder after this worker's decoder
      hipLaunchKernelGGL(spin, dim3(EG), dim3(1024), 158000, se[w], (unsigned long long)ET * 100, log, counter);
      if (sd[w] != se[w]) {
        CHK(hipEventRecord(e1[w], se[w]));
        CHK(hipStreamWaitEvent(sd[w], e1[w], 0));
      }
      for (int c = 0; c < chunks; ++c) {
        if (over > 1)
          hipLaunchKernelGGL(spin3, dim3(DG * over), dim3(1024), 121000, sd[w], (unsigned)DG,
                             tickets + ((size_t)(r * W + w) * chunks + c), (unsigned long long)DT * 100 / chunks, log,
                             counter);
        else
          hipLaunchKernelGGL(spin, dim3(DG), dim3(1024), 121000, sd[w], (unsigned long long)DT * 100 / chunks, log, counter);
      }
      if (sd[w] != se[w]) CHK(hipEventRecord(e2[w], sd[w]));
    }
  CHK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * total);
  CHK(hipMemcpy(h.data(), log, total * 16, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < total; ++i) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); }
  const unsigned long long lo = t0 + (t1 - t0) / 5, hi = t1 - (t1 - t0) / 5;
  double busy = 0;
  for (size_t i = 0; i < total; ++i) {
    const unsigned long long b = std::max(h[2 * i], lo), e = std::min(h[2 * i + 1], hi);
    if (e > b) busy += (double)(e - b);
  }
  const double ideal = (double)W * R * ((double)EG * ET + (double)DG * DT) / 256.0 * 1e-3;  // ms at 100 % packing
  printf("W=%d R=%d enc %dx%dus dec %dx%dus (x%d launches, over %d) prio %d split %d: busy %.1f %% of 256 CUs, %.2f ms (ideal %.2f ms), %.2f pairs/ms\n",
         W, R, EG, ET, DG, DT / chunks, chunks, over, dec_prio, split, 100.0 * busy / ((double)(hi - lo) * 256.0), (t1 - t0) * 1e-5, ideal,
         (double)W * R / ((t1 - t0) * 1e-5));
  return 0;
}
