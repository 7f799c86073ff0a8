// NOTE: the end stamp is taken behind a workgroup barrier. Without it thread 0 reports wave 0's
// own finish time, and wave 0 -- the oldest wave, favoured by the arbiter -- finishes in a
// third of the workgroup's time (a first version of this probe "measured" 400 GB/s per CU and
// 95 TB/s for the chip that way; the L2 fabric gives ~35 TB/s).
// What does ONE CU pull from L2 when it streams MFMA-fragment-ordered weights the way the
// persistent decoder / encoder do (16 waves, 1 KiB per wave-instruction, tiles strided over
// the waves), and what changes it? Sweeps: workgroups (16 / 64 / 256 = CUs busy), loads in
// flight per wave, footprint (1 / 3.4 / 8 MB: L2-resident or not), access form (buffer_load
// b128 into registers; LDS-DMA b128), wave->tile mapping (strided tiles vs one contiguous
// range per wave), cache policy (default / nt).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/l2_stream_probe.hip -o gpurun_out/l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

// MODE 0: tiles strided over waves (tile = wave + 16 i, 4 KiB per tile = 4 loads)
// MODE 1: one contiguous range per wave
// AUX: cache policy immediate (0 default, 2 nt)
template <int NF, int MODE, int AUX>
__global__ __launch_bounds__(1024) void stream_regs(const char *w, unsigned bytes, int passes, int *sink,
                                                    unsigned long long *stamps) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(w), 0, bytes, 0x00020000);
  const int n_chunks = bytes / (NF * 1024);        // chunks of NF KiB
  const int per_wave = n_chunks / 16;
  v4i acc = {0, 0, 0, 0};
  const unsigned long long t0 = wall_clock64();
  for (int p = 0; p < passes; ++p) {
    asm volatile("" ::: "memory");  // the same addresses every pass: keep the loads inside the loop
    v4i f[2][NF];
    auto load = [&](v4i(&b)[NF], int c) {
      const int chunk = MODE == 0 ? (wave + 16 * c) : (wave * per_wave + c);
#pragma unroll
      for (int i = 0; i < NF; ++i)
        b[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (chunk * NF + i) * 1024, AUX));
    };
    load(f[0], 0);
    for (int c = 0; c < per_wave; c += 2) {
      load(f[1], c + 1);
#pragma unroll
      for (int i = 0; i < NF; ++i) acc += f[0][i];
      load(f[0], c + 2);  // past the end: zeros, no traffic
#pragma unroll
      for (int i = 0; i < NF; ++i) acc += f[1][i];
    }
  }
  __syncthreads();  // wave 0 (oldest, favoured by arbitration) finishes long before the others
  const unsigned long long t1 = wall_clock64();
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[0] = 1;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
  (void)smem;
}

// LDS-DMA: buffer_load ... lds, 16 B per lane (gfx950), ring of NS slots of 1 KiB per wave
template <int NS, int AUX>
__global__ __launch_bounds__(1024) void stream_lds(const char *w, unsigned bytes, int passes, int *sink,
                                                   unsigned long long *stamps) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(w), 0, bytes, 0x00020000);
  const int n_kib = bytes / 1024;
  const int per_wave = n_kib / 16;
  __attribute__((address_space(3))) char *base =
      (__attribute__((address_space(3))) char *)(smem) + wave * NS * 1024;
  int acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int p = 0; p < passes; ++p) {
    asm volatile("" ::: "memory");
    for (int c = 0; c < per_wave; c += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int chunk = wave + 16 * (c + s);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, base + s * 1024, 16, lane * 16, chunk * 1024, 0, AUX);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += *(volatile __attribute__((address_space(3))) int *)(base + lane * 4);
    }
  }
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (acc == 0x12345678) sink[0] = 1;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

// The persistent decoder's output-layer loop, stripped to its memory behaviour: NB chunks of
// one 4 KiB tile in flight per wave, four dependent MFMAs and an arg-max epilogue per tile.
// PHASED: the passes are separated by a workgroup barrier and ~2 us without memory traffic,
// like the phases of the persistent decoder (does a stream that starts and stops lose rate?)
template <int NB, bool PHASED = false, int STAGGER = 0>
__global__ __launch_bounds__(1024) void stream_mfma(const char *w, unsigned bytes, int passes, int *sink,
                                                    unsigned long long *stamps) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(w), 0, bytes, 0x00020000);
  const int n_tiles = bytes / 4096;
  const int per_wave = n_tiles / 16;
  v4i a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = v4i{lane * 3 + i, lane * 5 + i, lane * 7 + i, lane * 11 + i};
  float bv[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
  int bi[4] = {0, 0, 0, 0};
  const unsigned long long t0 = wall_clock64();
  unsigned long long busy = 0;
  for (int p = 0; p < passes; ++p) {
    asm volatile("" ::: "memory");
    if (PHASED) {
      __syncthreads();
      const unsigned long long w0 = wall_clock64();
      while (wall_clock64() - w0 < 200) __builtin_amdgcn_s_sleep(8);
      __syncthreads();
    }
    const unsigned long long p0 = wall_clock64();
    if (STAGGER > 0) {  // waves 4..7, 8..11, 12..15 start 1, 2, 3 x STAGGER x 64 cycles late
      for (int i = 0; i < (wave >> 2); ++i) __builtin_amdgcn_s_sleep(STAGGER);
    } else if (STAGGER < 0) {  // every wave its own delay
      for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(-STAGGER);
    }
    v4i f[NB][4];
    auto load = [&](v4i(&b)[4], int c) {
      const int tile = wave + 16 * c;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        b[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (tile * 4 + i) * 1024, 0));
    };
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      load(f[k], k);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int c = 0; c < per_wave; c += NB) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], f[k][i], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = (float)acc[q] * 1.5f + 0.25f;
          const bool better = v > bv[q];
          bv[q] = better ? v : bv[q];
          bi[q] = better ? c + k : bi[q];
        }
        load(f[k], c + k + NB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (PHASED) __syncthreads();
    busy += wall_clock64() - p0;
  }
  __syncthreads();
  unsigned long long t1 = wall_clock64();
  if (PHASED) t1 = t0 + busy;  // only the streaming parts count
  if (bv[0] + bv[1] + bv[2] + bv[3] + (float)(bi[0] + bi[1] + bi[2] + bi[3]) == 0.12345f) sink[0] = 1;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
  (void)smem;
}

template <class K>
int run(const char *name, K kernel, const char *d_w, unsigned bytes, int wgs, int lds, int *d_sink,
        unsigned long long *d_st) {
  const int passes = (int)(400ull * 1024 * 1024 / bytes);  // ~400 MB per workgroup
  CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(1024), lds, 0, d_w, bytes, passes, d_sink, d_st);
    CHK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(2 * wgs);
  CHK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0, worst = 0;
  for (int i = 0; i < wgs; ++i) {
    const double us = (h[2 * i + 1] - h[2 * i]) * 0.01;
    sum += us;
    worst = us > worst ? us : worst;
  }
  const double mean_us = sum / wgs;
  const double gbs = (double)bytes * passes / (mean_us * 1e-6) / 1e9;
  printf("{\"probe\":\"%s\",\"workgroups\":%d,\"footprint_MB\":%.2f,\"GBs_per_cu\":%.1f,\"B_per_clk_at_2.4GHz\":%.1f,"
         "\"chip_TBs\":%.2f,\"worst_over_mean\":%.2f}\n",
         name, wgs, bytes / 1048576.0, gbs, gbs / 2.4, gbs * wgs / 1e3, worst / mean_us);
  fflush(stdout);
  return 0;
}

int main() {
  const unsigned max_bytes = 16u << 20;
  char *d_w;
  int *d_sink;
  unsigned long long *d_st;
  CHK(hipMalloc(&d_w, max_bytes));
  {
    std::vector<unsigned> h(max_bytes / 4);
    unsigned x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }  // random bytes (no value a cache could fold)
    CHK(hipMemcpy(d_w, h.data(), max_bytes, hipMemcpyHostToDevice));
  }
  CHK(hipMalloc(&d_sink, 4));
  CHK(hipMalloc(&d_st, 8 * 2 * 1024));
  const int lds = 120 * 1024;  // one workgroup per CU, like the persistent decoder
  const unsigned sizes[] = {1u << 20, 3407872u /* 3.25 MiB = tiny11 decoder + shortlist */};
  for (int wgs : {16, 256}) {
    for (unsigned bytes : sizes) {
      run("mfma_argmax_4chunks", stream_mfma<4>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_3chunks", stream_mfma<3>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_4chunks_phased", stream_mfma<4, true>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_4chunks_phased_stagger_4x1", stream_mfma<4, true, 1>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_4chunks_phased_stagger_4x4", stream_mfma<4, true, 4>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_4chunks_phased_stagger_16x1", stream_mfma<4, true, -1>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_4chunks_phased_stagger_16x3", stream_mfma<4, true, -3>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_2chunks_phased", stream_mfma<2, true>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("mfma_argmax_6chunks_phased", stream_mfma<6, true>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("regs_strided_4inflightx2", stream_regs<4, 0, 0>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("regs_strided_8inflightx2", stream_regs<8, 0, 0>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("regs_strided_16inflightx2", stream_regs<16, 0, 0>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("regs_contig_8inflightx2", stream_regs<8, 1, 0>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("regs_strided_8inflightx2_nt", stream_regs<8, 0, 2>, d_w, bytes, wgs, lds, d_sink, d_st);
      run("ldsdma_ring8", stream_lds<8, 0>, d_w, bytes, wgs, 128 * 1024, d_sink, d_st);
    }
  }
  return 0;
}
