// Large-M int8 GEMM (M >= 1024: the encoder of a whole batch as one-launch-per-stage
// kernels, and slimt::qmm::affine on many rows): 128-row x 128/256-column block tiles.
//
// Measured (base, M = 8192, alone on the GPU; gemm_rows_kernel in brackets): FFN1
// 512 -> 2048 with the relu + requantise epilogue 22.9 us (34.6); Q/K/V/O 512 -> 512 from
// f32 rows 19.5 us (14.5); FFN2 2048 -> 512 29.9 us (25). These GEMMs move 21..50 MB of
// activations each through HBM / Infinity Cache behind a ~4.5 us launch floor and do 1.4 us
// of MFMA work: the tiling only wins where it removes re-reads and re-quantisation (FFN1:
// 8 column blocks). launch_gemm therefore routes only the relu + requantise epilogue here;
// the f32-output path is kept, tested, and reachable with rows_per_block == 0.
//
// Same arithmetic as gemm_rows_kernel (kernels.hip): q = clamp(rne(x a_quant)), exact
// int32 accumulation on v_mfma_i32_16x16x64_i8, y = float(acc + 127 colsum) u + pb
// (Intgemm.inl.cc:123-153) -- only the tiling differs, so results are bit-identical:
//   * the block's 128 activation rows are quantised ONCE per 512-deep K chunk into LDS
//     (int8, 66 KiB) and shared by all waves; gemm_rows re-stages 16..64 rows per 256
//     columns, i.e. reads and re-quantises A eight times for a 2048-column GEMM;
//   * a wave owns 64 rows x 64 columns (4 x 4 MFMA tiles, 64 accumulator registers):
//     every weight fragment it loads feeds 4 MFMAs and every A fragment 4, so a k-step
//     is 16 MFMAs for 4 LDS reads + 4 KiB of weights (gemm_rows: 4..16 MFMAs per 4 KiB);
//   * the next two k-steps' weight fragments are always in flight (two register sets,
//     loads issued behind scheduling barriers so that the compiler's s_waitcnt counts
//     stay exact), including across the barrier that separates two K chunks;
//   * the MFMA operands are swapped (weights as A, activations as B): the accumulator
//     lane then holds 4 consecutive COLUMNS of one row, so outputs leave as 16-byte
//     (f32) or 4-byte (int8) pieces and the epilogue constants arrive as one 16-byte
//     load each instead of per-column scalars.
#include "device_common.h"
#include "kernels.h"

namespace slimt_hip {

namespace {

constexpr int BM = 128;            // rows per block
constexpr int KC = 512;            // K chunk resident in LDS
constexpr int LDAB = KC + 16;      // LDS row stride (bytes): conflict-free ds_read_b128
constexpr size_t kTileLds = (size_t)BM * LDAB + 64;  // + slack: a zero-weight k-step may read past the last row

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

struct TileArgs {
  GemmArgs g;
  int KS;  // K / 64
  float u;
  int row_blocks, col_blocks;
};

// CG column groups of 64 columns per block; waves = 2 (row halves) x CG.
template <int CG, int EPI, bool AF32>
__global__ __launch_bounds__(128 * CG, 1) void gemm_tile_kernel(TileArgs ka) {
  extern __shared__ __attribute__((aligned(16))) char A_lds[];
  const GemmArgs &a = ka.g;
  constexpr int T = 128 * CG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int rg = wave & 1, cg = wave >> 1;
  // Block -> tile, XCD-aware. Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8)
  // and each XCD has its own L2. The column blocks of one row block read the same 128
  // activation rows (256 KiB in f32): they get ids 8 apart, i.e. the same XCD at about the
  // same time, so the rows come from HBM / Infinity Cache once and from that L2 afterwards.
  // (Measured with a plain 2-D grid: an 8192 x 512 x 512 GEMM took 20 us re-reading its
  // 16.8 MB of f32 activations once per column block.) Placement only changes speed.
  const int per_group = 8 * ka.col_blocks;
  const int group = blockIdx.x / per_group, in_group = blockIdx.x - group * per_group;
  const int row_block = group * 8 + (in_group & 7);
  const int col_block = in_group >> 3;
  if (row_block >= ka.row_blocks) return;  // whole workgroup
  const int m0 = row_block * BM;
  const int nt0 = (col_block * CG + cg) * 4;  // first 16-column tile of this wave
  const int K = a.w.K, KS = ka.KS, M = a.M;
  const int n_tiles = a.w.n_tiles;
  const rsrc_t rw = make_rsrc(a.w.Wp, (unsigned)n_tiles * KS * 1024u);

  v4i acc[4][4];  // [row tile][column tile]
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = v4i{0, 0, 0, 0};

  // ALL weight fragments of a K chunk for this wave's 4 column tiles (8 k-steps x 4 tiles =
  // 128 registers) are requested before the chunk's activations are staged: 32 KiB in
  // flight per wave under the staging and the barrier. (A first version kept two k-steps in
  // flight: with 1-2 us to L2 / Infinity Cache every k-step waited, 20-30 us per GEMM.)
  // Past the matrix the offset lies behind the descriptor: zeros, no traffic.
  constexpr int KSC = KC / 64;
  v4i bf[KSC][4];
  auto load_chunk = [&](int ks0) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int ntile = nt0 + ct, kstep = ks0 + ks;
        const int frag = (ntile < n_tiles && kstep < KS) ? ntile * KS + kstep : n_tiles * KS;
        bf[ks][ct] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, frag * 1024, 0));
      }
  };

  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kc = (K - k0) < KC ? (K - k0) : KC;
    if (k0) __syncthreads();  // every wave is done reading the previous chunk
    load_chunk(k0 >> 6);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage the chunk: quantise (or copy) 128 rows x kc into LDS ----------------
    if constexpr (AF32) {
      const int upr = kc >> 2;  // float4 units per row: a power of two (gemm_tile_supported)
      const int sh = 31 - __builtin_clz(upr);
      const int total = BM * upr;
      for (int u0 = 0; u0 < total; u0 += 8 * T) {
        float4 f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // 8 loads in flight per thread
          const int u = u0 + i * T + tid;
          const int r = u >> sh, c4 = u & (upr - 1);
          int row = m0 + r;
          row = row < M ? row : M - 1;
          f[i] = (u < total) ? *reinterpret_cast<const float4 *>(a.x_f32 + (size_t)row * a.lda + k0 + c4 * 4)
                             : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int u = u0 + i * T + tid;
          const int r = u >> sh, c4 = u & (upr - 1);
          int packed = pack4(quantize1(f[i].x, a.w.a_quant), quantize1(f[i].y, a.w.a_quant),
                             quantize1(f[i].z, a.w.a_quant), quantize1(f[i].w, a.w.a_quant));
          if (m0 + r >= M) packed = 0;
          if (u < total) *reinterpret_cast<int *>(A_lds + r * LDAB + c4 * 4) = packed;
        }
      }
    } else {
      const int upr = kc >> 4;  // 16-byte units per row: a power of two
      const int sh = 31 - __builtin_clz(upr);
      const int total = BM * upr;
      for (int u0 = 0; u0 < total; u0 += 8 * T) {
        v4i g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int u = u0 + i * T + tid;
          const int r = u >> sh, c = u & (upr - 1);
          int row = m0 + r;
          row = row < M ? row : M - 1;
          g[i] = (u < total) ? *reinterpret_cast<const v4i *>(a.x_i8 + (size_t)row * a.lda + k0 + c * 16)
                             : v4i{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int u = u0 + i * T + tid;
          const int r = u >> sh, c = u & (upr - 1);
          v4i v = g[i];
          if (m0 + r >= M) v = v4i{0, 0, 0, 0};
          if (u < total) *reinterpret_cast<v4i *>(A_lds + r * LDAB + c * 16) = v;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- the chunk's k-steps (a k-step past K multiplies zero weights) ---------------
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      v4i af[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        af[rt] = *reinterpret_cast<const v4i *>(A_lds + (rg * 64 + rt * 16 + lr) * LDAB + ks * 64 + lg * 16);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)  // weights as the A operand: acc lane = (row lr, columns 4 lg .. 4 lg + 3)
          acc[rt][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[ks][ct], af[rt], acc[rt][ct], 0, 0, 0);
    }
  }

  // ---- epilogue: y = float(acc + 127 colsum) * u + prepared_bias (Intgemm.inl.cc:146-153)
  // The block's output tile goes through LDS (the activation tile is dead) and leaves in a
  // second pass as whole contiguous rows: written straight from the accumulator layout (16
  // rows x 16 / 64 bytes per instruction) the stores were issue-bound -- 7-12 us of a 21-31 us
  // GEMM for 16.8 MB, 3 bytes per clock and CU.
  constexpr int BN = 64 * CG;
  constexpr int LDC = EPI == EPI_PLAIN ? (BN + 4) * 4 : BN + 16;  // bytes per tile row
  static_assert((size_t)BM * LDC <= kTileLds, "the output tile must fit the activation tile's LDS");
  const float u = ka.u;
  const rsrc_t rc = make_rsrc(a.w.colsum, (unsigned)n_tiles * 64u);
  const rsrc_t rp = make_rsrc(a.w.pb, (unsigned)n_tiles * 64u);
  __syncthreads();  // every wave is done with the activation tile
  {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int ntile = nt0 + ct;  // past the matrix: colsum / pb read zeros, the columns are dropped below
      const v4i cs4 = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rc, lg * 16, ntile * 64, 0));
      const float4 pb4 =
          __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp, lg * 16, ntile * 64, 0));
      const float pbv[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
      const int tc = cg * 64 + ct * 16 + lg * 4;  // column inside the tile
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int tr = rg * 64 + rt * 16 + lr;  // row inside the tile
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = (float)(acc[rt][ct][r] + 127 * cs4[r]) * u;
          v[r] = t + pbv[r];
        }
        if constexpr (EPI == EPI_PLAIN) {
          *reinterpret_cast<float4 *>(A_lds + tr * LDC + tc * 4) = float4{v[0], v[1], v[2], v[3]};
        } else {  // relu (TensorOps.cc:163-181), requantised for the next affine
          int q[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) q[r] = quantize1(v[r] > 0.0f ? v[r] : 0.0f, a.a_quant_out);
          *reinterpret_cast<int *>(A_lds + tr * LDC + tc) = pack4(q[0], q[1], q[2], q[3]);
        }
      }
    }
  }
  __syncthreads();
  const int n0 = col_block * BN;  // first column of the tile
  if constexpr (EPI == EPI_PLAIN) {
    constexpr int UPR = BN / 4;  // float4 units per tile row
    {
      for (int uidx = tid; uidx < BM * UPR; uidx += T) {
        const int tr = uidx / UPR, c4 = uidx - tr * UPR;
        const int row = m0 + tr, col = n0 + c4 * 4;
        if (row >= M || col >= a.w.N) continue;
        float4 v = *reinterpret_cast<const float4 *>(A_lds + tr * LDC + c4 * 16);
        if (a.res) {  // residual, Modules.cc:254,314
          const float4 rr = *reinterpret_cast<const float4 *>(a.res + (size_t)row * a.ldres + col);
          v.x = v.x + rr.x;
          v.y = v.y + rr.y;
          v.z = v.z + rr.z;
          v.w = v.w + rr.w;
        }
        *reinterpret_cast<float4 *>(a.y + (size_t)row * a.ldy + col) = v;
      }
    }
  } else {
    constexpr int UPR = BN / 16;  // 16-byte units per tile row
    for (int uidx = tid; uidx < BM * UPR; uidx += T) {
      const int tr = uidx / UPR, c16 = uidx - tr * UPR;
      const int row = m0 + tr, col = n0 + c16 * 16;
      if (row >= M || col >= a.w.N) continue;
      *reinterpret_cast<v4i *>(a.y_i8 + (size_t)row * a.ldy8 + col) =
          *reinterpret_cast<const v4i *>(A_lds + tr * LDC + c16 * 16);  // N % 16 == 0 (gemm_tile_supported)
    }
  }
}

template <int CG, int EPI>
hipError_t launch_tile_t(const TileArgs &ka, dim3 grid, hipStream_t st) {
  auto run = [&](auto kernel) {
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(kernel), (int)kTileLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, dim3(128 * CG), kTileLds, st, ka);
    return hipGetLastError();
  };
  if (ka.g.x_f32) return run(gemm_tile_kernel<CG, EPI, true>);
  return run(gemm_tile_kernel<CG, EPI, false>);
}

}  // namespace

bool gemm_tile_supported(const GemmArgs &a, int epilogue) {
  if (epilogue != EPI_PLAIN && epilogue != EPI_RELU_Q) return false;
  // K chunks of 512 (or one chunk of 128 / 256): the staging indexes rows by shifts
  const bool k_ok = a.w.K == 128 || a.w.K == 256 || (a.w.K > 0 && a.w.K % 512 == 0);
  if (a.M < 1024 || !k_ok || a.w.N < 64 || a.w.N % 16 != 0) return false;
  if ((a.x_f32 == nullptr) == (a.x_i8 == nullptr)) return false;
  if (a.x_f32 && (a.lda % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x_f32) & 15))) return false;
  if (a.x_i8 && (a.lda % 16 != 0 || (reinterpret_cast<uintptr_t>(a.x_i8) & 15))) return false;
  if (epilogue == EPI_PLAIN) {
    if (a.ldy % 4 != 0 || (reinterpret_cast<uintptr_t>(a.y) & 15)) return false;
    if (a.res && (a.ldres % 4 != 0 || (reinterpret_cast<uintptr_t>(a.res) & 15))) return false;
    if (a.kc_S || a.raw_acc) return false;  // the K/V-cache stores stay with gemm_rows_kernel
  } else {
    if (a.ldy8 % 16 != 0 || (reinterpret_cast<uintptr_t>(a.y_i8) & 15)) return false;
  }
  return true;
}

hipError_t launch_gemm_tile(const GemmArgs &a, int epilogue, hipStream_t st) {
  if (!gemm_tile_supported(a, epilogue)) return hipErrorInvalidValue;
  TileArgs ka;
  ka.g = a;
  ka.KS = a.w.K / 64;
  ka.u = a.w.u;
  const int row_blocks = (a.M + BM - 1) / BM;
  // 256-column blocks when that still fills the chip (>= 2 workgroups per CU's worth of
  // blocks), else 128-column blocks: a 512-column GEMM over 8192 rows = 64 x 4 blocks
  const int cb256 = (a.w.N + 255) / 256, cb128 = (a.w.N + 127) / 128;
  // (f32 outputs: 128-column blocks always -- the output tile is staged in the same 66 KiB of LDS)
  const bool wide = epilogue == EPI_RELU_Q && (long)row_blocks * cb256 >= 384;
  ka.row_blocks = row_blocks;
  ka.col_blocks = wide ? cb256 : cb128;
  const dim3 grid(((row_blocks + 7) / 8) * 8 * ka.col_blocks);
  if (wide) return launch_tile_t<4, EPI_RELU_Q>(ka, grid, st);
  return epilogue == EPI_PLAIN ? launch_tile_t<2, EPI_PLAIN>(ka, grid, st) : launch_tile_t<2, EPI_RELU_Q>(ka, grid, st);
}

}  // namespace slimt_hip
