// gfx950 (MI355X / CDNA4) kernels for slimt's int8 transformer-NMT hot path.
// Written for wave64 + v_mfma_i32_16x16x64_i8 only: no other target, no
// portability layer.
//
// Reference semantics (file:line in jerinphilip/slimt):
//   int8 affine / dot / select  slimt/qmm/Intgemm.inl.cc:7-226 (Int8Shift)
//   layer_norm                  slimt/TensorOps.cc:542-580
//   softmax                     slimt/TensorOps.cc:296-314
//   highway / sigmoid           slimt/TensorOps.cc:33-36,662-682
//   attention                   slimt/Modules.cc:24-86
//   SSRU                        slimt/Modules.cc:190-235
//   embedding transform         slimt/Transformer.cc:24-49,133-160
//   greedy sample + record      slimt/Transformer.cc:279-339, Model.cc:127-137
#include "kernels.h"

#include "device_common.h"

namespace slimt_hip {

// ---------------------------------------------------------------------------
// weight packing (load time / once per batch for the shortlist)
// ---------------------------------------------------------------------------

size_t packed_weight_bytes(int K, int N) {
  size_t n_tiles = (size_t)(N + 15) / 16;
  return n_tiles * (size_t)(K / 64) * 64 * 16;
}

// one block per 16-column tile of the logical [K, N] matrix
__global__ __launch_bounds__(256) void pack_weight_kernel(PackArgs a) {
  pack_weight_tile(a, blockIdx.x, threadIdx.x, 256);
}

hipError_t launch_pack_weight(const int8_t *W, int K, int N, const uint32_t *idx,
                              const float *bias, float a_quant, float b_quant, void *Wp,
                              int *colsum, float *pb, hipStream_t st) {
  PackArgs a;
  a.W = W;
  a.K = K;
  a.N = N;
  a.idx = idx;
  a.bias = bias;
  a.mult = pack_mult(a_quant, b_quant);
  a.Wp = Wp;
  a.colsum = colsum;
  a.pb = pb;
  const int n_tiles = (N + 15) / 16;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(n_tiles), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// int8 GEMM on v_mfma_i32_16x16x64_i8 with fused quantise prologue and
// dequant / activation / LayerNorm / argmax epilogues
// ---------------------------------------------------------------------------
//
// Block = 4 waves. Block tile = (16*RM rows) x (4 waves * NT tiles * 16 cols).
// A (activations) is quantised once per K-chunk into LDS as int8 and shared
// by the 4 waves; each wave streams its own B fragments (pre-tiled weights,
// one fully coalesced 1 KiB load per MFMA operand) straight into registers
// and reuses each across the RM row tiles.
//
// MFMA operand maps (v_mfma_i32_16x16x64_i8, checked by tests/test_gpu_qmm):
//   A: lane l holds A[row = l & 15][k = 16 (l >> 4) .. +15]   (16 int8 = v4i)
//   B: lane l holds B[k = 16 (l >> 4) .. +15][col = l & 15]
//   C: lane l, reg r holds C[row = 4 (l >> 4) + r][col = l & 15]

constexpr int KCH = 256;       // K chunk staged in LDS
constexpr int LDA = KCH + 16;  // padded LDS row stride (bytes): conflict-free b128 reads

struct GemmKArgs {
  GemmArgs g;
  int KS;           // K / 64
  float u;
};

template <int RM>
__device__ __forceinline__ void stage_A_f32(char *A_lds, const float *x, int lda, int M, int m0,
                                            int k0, int kc, float aq, int tid) {
  constexpr int R = 16 * RM;
  const int units_per_row = kc >> 2;  // float4 units
  for (int u = tid; u < R * units_per_row; u += 256) {
    const int r = u / units_per_row, c4 = u - r * units_per_row;
    const int row = m0 + r;
    int packed = 0;
    if (row < M) {
      const float4 f = *reinterpret_cast<const float4 *>(x + (size_t)row * lda + k0 + c4 * 4);
      packed = pack4(quantize1(f.x, aq), quantize1(f.y, aq), quantize1(f.z, aq),
                     quantize1(f.w, aq));
    }
    *reinterpret_cast<int *>(A_lds + r * LDA + c4 * 4) = packed;
  }
}

// Full-chunk staging in two halves: fetch_* issues a group of G global loads
// (row-clamped, unconditional) so that they are all in flight at once, put_*
// quantises / stores them to LDS. With the loop above each thread's 4..16
// loads were dependent round trips (load, quantise, store, next): an encoder
// GEMM over 8192 rows spent most of its time on that latency chain.
template <int G>
__device__ __forceinline__ void fetch_A_f32(float4 (&f)[G], const float *x, int lda, int M,
                                            int m0, int k0, int u0, int tid) {
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int u = u0 + tid + 256 * i, r = u >> 6, c4 = u & 63;  // KCH / 4 = 64 units per row
    int row = m0 + r;
    row = row < M ? row : M - 1;
    f[i] = *reinterpret_cast<const float4 *>(x + (size_t)row * lda + k0 + c4 * 4);
  }
}

template <int G>
__device__ __forceinline__ void put_A_f32(char *A_lds, const float4 (&f)[G], int M, int m0,
                                          float aq, int u0, int tid) {
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int u = u0 + tid + 256 * i, r = u >> 6, c4 = u & 63;
    int packed = pack4(quantize1(f[i].x, aq), quantize1(f[i].y, aq), quantize1(f[i].z, aq),
                       quantize1(f[i].w, aq));
    if (m0 + r >= M) packed = 0;
    *reinterpret_cast<int *>(A_lds + r * LDA + c4 * 4) = packed;
  }
}

template <int RM>
__device__ __forceinline__ void fetch_A_i8(v4i (&g)[RM], const int8_t *x, int lda, int M, int m0,
                                           int k0, int tid) {
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int u = tid + 256 * i, r = u >> 4, c = u & 15;  // KCH / 16 = 16 units per row
    int row = m0 + r;
    row = row < M ? row : M - 1;
    g[i] = *reinterpret_cast<const v4i *>(x + (size_t)row * lda + k0 + c * 16);
  }
}

template <int RM>
__device__ __forceinline__ void put_A_i8(char *A_lds, const v4i (&g)[RM], int M, int m0, int tid) {
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int u = tid + 256 * i, r = u >> 4, c = u & 15;
    v4i v = g[i];
    if (m0 + r >= M) v = v4i{0, 0, 0, 0};
    *reinterpret_cast<v4i *>(A_lds + r * LDA + c * 16) = v;
  }
}

template <int RM>
__device__ __forceinline__ void stage_A_i8(char *A_lds, const int8_t *x, int lda, int M, int m0,
                                           int k0, int kc, int tid) {
  constexpr int R = 16 * RM;
  const int units_per_row = kc >> 4;  // 16-byte units
  for (int u = tid; u < R * units_per_row; u += 256) {
    const int r = u / units_per_row, c = u - r * units_per_row;
    const int row = m0 + r;
    v4i v = {0, 0, 0, 0};
    if (row < M) v = *reinterpret_cast<const v4i *>(x + (size_t)row * lda + k0 + c * 16);
    *reinterpret_cast<v4i *>(A_lds + r * LDA + c * 16) = v;
  }
}

// One block tile of the GEMM: rows [row0 + bx * R, ...) below row_end, column
// block by. A kernel of its own (gemm_rows_kernel) or a stage of the persistent
// per-sentence encoder (encode_long_kernel), which walks the tiles of its
// sentence with a barrier between calls.
template <int RM, int NT, int EPI>
__device__ __forceinline__ void gemm_rows_body(const GemmKArgs &ka, int bx, int by, int row0,
                                               int row_end, char *smem) {
  const GemmArgs &a = ka.g;
  constexpr int R = 16 * RM;
  constexpr int BN = 64 * NT;  // columns per block
  char *A_lds = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform in the compiler's eyes
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = row0 + bx * R;
  const int nt0 = (by * 4 + wave) * NT;  // first 16-col tile of this wave
  const int K = a.w.K, KS = ka.KS;
  const int n_tiles = a.w.n_tiles;
  const v4i *Wp = reinterpret_cast<const v4i *>(a.w.Wp);

  v4i acc[RM][NT];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = v4i{0, 0, 0, 0};

  // The B fragments of a whole K chunk (KCH / 64 steps x NT tiles) are requested
  // at once, before A is staged: weights do not depend on the activations, so the
  // fetch overlaps the staging + barrier instead of stalling every k-step.
  constexpr int KSC = KCH / 64;
  v4i bf[KSC][NT];
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<v4i *>(Wp), 0, (unsigned)n_tiles * KS * 1024u, 0x00020000);
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      const int kstep = (k0 >> 6) + ks;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        // buffer load: descriptor and fragment offset are scalar, the only vector operand
        // is lane * 16 (no 64-bit per-lane addresses); past the matrix it returns zeros
        const int ntile = nt0 + nt;
        const int frag = (ntile < n_tiles && kstep < KS) ? ntile * KS + kstep : n_tiles * KS;
        bf[ks][nt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, frag * 1024, 0));
      }
    }
  };
  load_chunk(0);
  // K a multiple of the chunk: all of a chunk's A loads in flight at once
  // (fetch_* / put_*). Tried and measured, not kept: holding the NEXT chunk's
  // rows in registers across the MFMA loop, and a full software pipeline (two
  // LDS tiles, two sets of B fragments): 250-360 registers = one block per CU,
  // and M = 8192 GEMMs ran 1.5x SLOWER (31 vs 20 us) -- two resident blocks per
  // CU hide more latency than the deeper prefetch of one.
  const bool whole_chunks = (K % KCH) == 0;
  for (int k0 = 0; k0 < K; k0 += KCH) {
    const int kc = (K - k0) < KCH ? (K - k0) : KCH;
    if (k0) __syncthreads();
    if (whole_chunks) {
      if (a.x_f32) {
        constexpr int G = (NT >= 8 || RM <= 2) ? 4 : 8;  // loads in flight per thread (registers -> blocks per CU)
#pragma unroll
        for (int g = 0; g < 4 * RM / G; ++g) {
          float4 fa[G];
          fetch_A_f32<G>(fa, a.x_f32, a.lda, row_end, m0, k0, g * G * 256, tid);
          put_A_f32<G>(A_lds, fa, row_end, m0, a.w.a_quant, g * G * 256, tid);
        }
      } else {
        v4i ga[RM];
        fetch_A_i8<RM>(ga, a.x_i8, a.lda, row_end, m0, k0, tid);
        put_A_i8<RM>(A_lds, ga, row_end, m0, tid);
      }
    } else if (a.x_f32) {
      stage_A_f32<RM>(A_lds, a.x_f32, a.lda, row_end, m0, k0, kc, a.w.a_quant, tid);
    } else {
      stage_A_i8<RM>(A_lds, a.x_i8, a.lda, row_end, m0, k0, kc, tid);
    }
    __syncthreads();
    const int ksteps = kc >> 6;
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      if (ks < ksteps) {
        v4i af[RM];
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
          af[rm] = *reinterpret_cast<const v4i *>(A_lds + (rm * 16 + lr) * LDA + ks * 64 + lg * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int rm = 0; rm < RM; ++rm)
            acc[rm][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rm], bf[ks][nt], acc[rm][nt], 0, 0, 0);
      }
    }
    if (k0 + KCH < K) load_chunk(k0 + KCH);  // under the next chunk's staging
  }

  // ---- epilogue -----------------------------------------------------------
  // y = float(acc + 127 colsum) * u + prepared_bias   (Intgemm.inl.cc:146-153)
  const float u = ka.u;
  if constexpr (EPI == EPI_PLAIN || EPI == EPI_RELU_Q || EPI == EPI_ACC) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ntile = nt0 + nt;
      if (ntile >= n_tiles) continue;
      const int col = ntile * 16 + lr;
      const int cs = a.w.colsum[col];
      const float pb = a.w.pb[col];
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + rm * 16 + lg * 4 + r;
          if (row >= row_end || col >= a.w.N) continue;
          const int accS = acc[rm][nt][r] + 127 * cs;
          if constexpr (EPI == EPI_ACC) {
            a.acc_out[(size_t)row * a.w.N + col] = accS;
          } else {
            float v = (float)accS * u;
            v = v + pb;
            if constexpr (EPI == EPI_PLAIN) {
              if (a.raw_acc) v = (float)accS;
              if (a.kc_S) {
                const int sb = row / a.kc_S, j = row - sb * a.kc_S;
                const int h = col / a.kc_dh, d = col - h * a.kc_dh;
                const size_t chunk = ((size_t)sb * (a.w.N / a.kc_dh) + h) * (a.kc_dh >> 2) + (d >> 2);
                a.y[(chunk * a.kc_S + j) * 4 + (d & 3)] = v;
              } else {
                if (a.res) v = v + a.res[(size_t)row * a.ldres + col];  // residual, Modules.cc:254,314
                a.y[(size_t)row * a.ldy + col] = v;
              }
            } else {
              v = v > 0.0f ? v : 0.0f;  // relu, TensorOps.cc:163-181
              a.y_i8[(size_t)row * a.ldy8 + col] = (int8_t)quantize1(v, a.a_quant_out);
            }
          }
        }
    }
  } else if constexpr (EPI == EPI_RES_LN) {
    // the block owns complete rows (gridDim.y == 1, N == BN)
    float *rowbuf = reinterpret_cast<float *>(smem + R * LDA);
    constexpr int LDR = BN + 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = (nt0 + nt) * 16 + lr;
      const int cs = a.w.colsum[col];
      const float pb = a.w.pb[col];
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = rm * 16 + lg * 4 + r;
          const int row = m0 + rl;
          float v = 0.0f;
          if (row < row_end) {
            v = (float)(acc[rm][nt][r] + 127 * cs) * u;
            v = v + pb;
            v = v + a.res[(size_t)row * a.ldres + col];
          }
          rowbuf[rl * LDR + col] = v;
        }
    }
    __syncthreads();
    for (int rl = wave; rl < R; rl += 4) {
      const int row = m0 + rl;
      if (row < row_end)
        wave_layer_norm_row(rowbuf + rl * LDR, a.ln_scale, a.ln_bias, a.eps, BN,
                            a.y + (size_t)row * a.ldy, lane);
    }
  } else if constexpr (EPI == EPI_ARGMAX) {
    // first-max (strict >, lowest column wins ties): Transformer.cc:287-297
    float *red_v = reinterpret_cast<float *>(smem + R * LDA);
    int *red_i = reinterpret_cast<int *>(red_v + 4 * R);
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float bv = -3.402823466e+38f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int ntile = nt0 + nt;
          const int col = ntile * 16 + lr;
          if (ntile < n_tiles && col < a.w.N) {
            float v = (float)(acc[rm][nt][r] + 127 * a.w.colsum[col]) * u;
            v = v + a.w.pb[col];
            if (v > bv || (v == bv && col < bi)) {
              bv = v;
              bi = col;
            }
          }
        }
        // across the 16 lanes that share these rows
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          const float ov = __shfl_xor(bv, m, 64);
          const int oi = __shfl_xor(bi, m, 64);
          if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
          }
        }
        if (lr == 0) {
          const int rl = rm * 16 + lg * 4 + r;
          red_v[wave * R + rl] = bv;
          red_i[wave * R + rl] = bi;
        }
      }
    __syncthreads();
    if (tid < R) {
      const int row = m0 + tid;
      float bv = red_v[tid];
      int bi = red_i[tid];
      for (int w = 1; w < 4; ++w) {
        const float ov = red_v[w * R + tid];
        const int oi = red_i[w * R + tid];
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      if (row < row_end) {
        a.part_val[(size_t)row * a.n_parts + by] = bv;
        a.part_idx[(size_t)row * a.n_parts + by] = bi;
      }
    }
  }
}

template <int RM, int NT, int EPI>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmKArgs ka) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_rows_body<RM, NT, EPI>(ka, blockIdx.x, blockIdx.y, 0, ka.g.M, smem);
}

int gemm_col_blocks(int N, int epilogue, int *nt_out) {
  int nt;
  if (epilogue == EPI_RES_LN) {
    nt = N / 64;  // block must own the whole row
  } else {
    nt = N >= 256 ? 4 : (N >= 128 ? 2 : 1);
  }
  if (nt_out) *nt_out = nt;
  const int bn = 64 * nt;
  return (N + bn - 1) / bn;
}

template <int RM, int NT>
static hipError_t launch_gemm_t(const GemmKArgs &ka, int epi, dim3 grid, hipStream_t st) {
  constexpr int R = 16 * RM;
  size_t lds = (size_t)R * LDA;
  switch (epi) {
    case EPI_PLAIN:
      hipLaunchKernelGGL((gemm_rows_kernel<RM, NT, EPI_PLAIN>), grid, dim3(256), lds, st, ka);
      break;
    case EPI_RELU_Q:
      hipLaunchKernelGGL((gemm_rows_kernel<RM, NT, EPI_RELU_Q>), grid, dim3(256), lds, st, ka);
      break;
    case EPI_ACC:
      hipLaunchKernelGGL((gemm_rows_kernel<RM, NT, EPI_ACC>), grid, dim3(256), lds, st, ka);
      break;
    case EPI_RES_LN:
      lds += (size_t)R * (64 * NT + 4) * sizeof(float);
      hipLaunchKernelGGL((gemm_rows_kernel<RM, NT, EPI_RES_LN>), grid, dim3(256), lds, st, ka);
      break;
    case EPI_ARGMAX:
      lds += (size_t)R * 4 * (sizeof(float) + sizeof(int));
      hipLaunchKernelGGL((gemm_rows_kernel<RM, NT, EPI_ARGMAX>), grid, dim3(256), lds, st, ka);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs &a, int epilogue, int rows_per_block, hipStream_t st) {
  if (a.w.K % 64 != 0 || a.w.K <= 0 || a.M <= 0 || a.w.N <= 0) return hipErrorInvalidValue;
  // the 128-row tiling (gemm_tile.hip): where it measures faster (its header), or on request
  if ((epilogue == EPI_RELU_Q || rows_per_block == 0) && gemm_tile_supported(a, epilogue))
    return launch_gemm_tile(a, epilogue, st);
  if (rows_per_block == 0) rows_per_block = a.M >= 512 ? 32 : 16;
  if ((a.x_f32 == nullptr) == (a.x_i8 == nullptr)) return hipErrorInvalidValue;
  if (a.x_f32 && (a.lda % 4 != 0)) return hipErrorInvalidValue;
  if (a.x_i8 && (a.lda % 16 != 0)) return hipErrorInvalidValue;
  int nt;
  const int col_blocks = gemm_col_blocks(a.w.N, epilogue, &nt);
  if (epilogue == EPI_RES_LN && (a.w.N % 64 != 0 || a.w.N > 512)) return hipErrorInvalidValue;
  if (epilogue == EPI_ARGMAX && a.n_parts != col_blocks) return hipErrorInvalidValue;
  GemmKArgs ka;
  ka.g = a;
  ka.KS = a.w.K / 64;
  ka.u = a.w.u;
  int rm = rows_per_block / 16;
  if (nt == 8 && rm > 2) rm = 2;  // register budget
  const dim3 grid((a.M + 16 * rm - 1) / (16 * rm), col_blocks);
#define SLIMT_GEMM_CASE(RM_, NT_) \
  if (rm == RM_ && nt == NT_) return launch_gemm_t<RM_, NT_>(ka, epilogue, grid, st);
  SLIMT_GEMM_CASE(1, 1) SLIMT_GEMM_CASE(1, 2) SLIMT_GEMM_CASE(1, 4) SLIMT_GEMM_CASE(1, 8)
  SLIMT_GEMM_CASE(2, 1) SLIMT_GEMM_CASE(2, 2) SLIMT_GEMM_CASE(2, 4) SLIMT_GEMM_CASE(2, 8)
  SLIMT_GEMM_CASE(4, 1) SLIMT_GEMM_CASE(4, 2) SLIMT_GEMM_CASE(4, 4)
#undef SLIMT_GEMM_CASE
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// SSRU cell (Modules.cc:190-235): f = affine(Wf, bf)(x); Wx = dot(W)(x);
// c' = sigmoid(f) c + (1 - sigmoid(f)) Wx; h = LN(x + relu(c')); state <- c'
// One block = 16 sentences x the full D columns; both GEMMs share the block.
// ---------------------------------------------------------------------------

template <int NT>
__global__ __launch_bounds__(256) void ssru_kernel(SsruArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 64 * NT;
  constexpr int LDR = BN + 4;
  char *Af = smem;             // x quantised with Wf's multiplier
  char *Aw = smem + 16 * LDA;  // x quantised with W's multiplier
  float *rowbuf = reinterpret_cast<float *>(smem + 2 * 16 * LDA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int nt0 = wave * NT;
  const int D = a.D, KS = D / 64;
  const v4i *Wpf = reinterpret_cast<const v4i *>(a.wf.Wp);
  const v4i *Wpw = reinterpret_cast<const v4i *>(a.w.Wp);

  v4i accf[NT], accw[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) accf[nt] = accw[nt] = v4i{0, 0, 0, 0};

  for (int k0 = 0; k0 < D; k0 += KCH) {
    const int kc = (D - k0) < KCH ? (D - k0) : KCH;
    if (k0) __syncthreads();
    stage_A_f32<1>(Af, a.x, D, a.B, m0, k0, kc, a.wf.a_quant, tid);
    stage_A_f32<1>(Aw, a.x, D, a.B, m0, k0, kc, a.w.a_quant, tid);
    __syncthreads();
    const int ksteps = kc >> 6;
    for (int ks = 0; ks < ksteps; ++ks) {
      const v4i af = *reinterpret_cast<const v4i *>(Af + lr * LDA + ks * 64 + lg * 16);
      const v4i aw = *reinterpret_cast<const v4i *>(Aw + lr * LDA + ks * 64 + lg * 16);
      const int kstep = (k0 >> 6) + ks;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const size_t off = ((size_t)(nt0 + nt) * KS + kstep) * 64 + lane;
        accf[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, Wpf[off], accf[nt], 0, 0, 0);
        accw[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw, Wpw[off], accw[nt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = (nt0 + nt) * 16 + lr;
    const int csf = a.wf.colsum[col], csw = a.w.colsum[col];
    const float pbf = a.wf.pb[col], pbw = a.w.pb[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = lg * 4 + r;
      const int row = m0 + rl;
      float v = 0.0f;
      if (row < a.B) {
        float f = (float)(accf[nt][r] + 127 * csf) * a.wf.u;
        f = f + pbf;
        float wx = (float)(accw[nt][r] + 127 * csw) * a.w.u;
        wx = wx + pbw;
        const size_t o = (size_t)row * D + col;
        const float c = a.state[o];
        const float sg = sigmoid_p(f);  // highway(c, Wx, f), TensorOps.cc:674-678
        const float t1 = sg * c;
        const float t2 = (1.0f - sg) * wx;
        const float cn = t1 + t2;
        a.state[o] = cn;
        const float y = cn > 0.0f ? cn : 0.0f;
        v = a.x[o] + y;  // Modules.cc:230
      }
      rowbuf[rl * LDR + col] = v;
    }
  }
  __syncthreads();
  for (int rl = wave; rl < 16; rl += 4) {
    const int row = m0 + rl;
    if (row < a.B)
      wave_layer_norm_row(rowbuf + rl * LDR, a.ln_scale, a.ln_bias, a.eps, D,
                          a.h + (size_t)row * D, lane);
  }
}

hipError_t launch_ssru(const SsruArgs &a, hipStream_t st) {
  if (a.D % 64 != 0 || a.D > 512) return hipErrorInvalidValue;
  const int nt = a.D / 64;
  const dim3 grid((a.B + 15) / 16);
  const size_t lds = 2 * 16 * LDA + 16 * (size_t)(a.D + 4) * sizeof(float);
  switch (nt) {
    case 1: hipLaunchKernelGGL(ssru_kernel<1>, grid, dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL(ssru_kernel<2>, grid, dim3(256), lds, st, a); break;
    case 4: hipLaunchKernelGGL(ssru_kernel<4>, grid, dim3(256), lds, st, a); break;
    case 8: hipLaunchKernelGGL(ssru_kernel<8>, grid, dim3(256), lds, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// embeddings
// ---------------------------------------------------------------------------

// E[tok][d] = float(q) * (1/mult)   (Io.cc:275-283), then * sqrt(D), then
// + pos[s][d]  (Transformer.cc:24-49). Separate roundings, as in the reference.
__device__ __forceinline__ float embed1(const EmbedArgs &e, uint32_t tok, int d, const float *pos) {
  const float v = (float)e.wemb[(size_t)embed_row(e, tok) * e.D + d] * e.inv_mult;
  const float s = v * e.sqrt_d;
  return s + pos[d];
}

__global__ void embed_encoder_kernel(EmbedArgs e, const uint32_t *ids, int S, float *x) {
  const int row = blockIdx.x;  // b * S + s
  const int s = row % S;
  const uint32_t tok = embed_row(e, ids[row]);
  for (int d = threadIdx.x; d < e.D; d += blockDim.x)
    x[(size_t)row * e.D + d] = embed1(e, tok, d, e.pos + (size_t)s * e.D);
}

hipError_t launch_embed_encoder(const EmbedArgs &e, const uint32_t *ids, int B, int S, float *x,
                                hipStream_t st) {
  hipLaunchKernelGGL(embed_encoder_kernel, dim3(B * S), dim3(e.D < 256 ? 64 : 256), 0, st, e, ids,
                     S, x);
  return hipGetLastError();
}

// One block (64 threads) per sentence. Position is ALWAYS 0 in the decoder
// (Transformer.cc:160); step 0 embeds zeros (Transformer.cc:138-144).
__global__ __launch_bounds__(64) void decode_begin_step_kernel(EmbedArgs e, DecodeState s,
                                                               int first, int with_embed,
                                                               const float *part_val,
                                                               const int *part_idx, int n_parts,
                                                               float *x) {
  const int b = blockIdx.x;
  __shared__ uint32_t tok_s;
  if (!first && threadIdx.x == 0) {
    // finish the argmax: partials are in ascending column order
    float bv = part_val[(size_t)b * n_parts];
    int bi = part_idx[(size_t)b * n_parts];
    for (int p = 1; p < n_parts; ++p) {
      const float v = part_val[(size_t)b * n_parts + p];
      if (v > bv) {
        bv = v;
        bi = part_idx[(size_t)b * n_parts + p];
      }
    }
    if (bi == 0x7fffffff) bi = 0;  // every logit NaN or -inf: class 0 (Transformer.cc:287-298), never out of range
    if (s.pb0 && (*s.pb0 != *s.pb0 || s.u_out != s.u_out)) bi = 0;  // logit[0] is NaN: the reference's scan never leaves class 0
    const uint32_t tok = s.shortlist ? s.shortlist[bi] : (uint32_t)bi;
    s.prev[b] = tok;
    if (!s.finished[b]) {  // record(), Model.cc:127-137
      const uint32_t n = s.out_len[b];
      if ((int)n < s.Tmax) s.out_ids[(size_t)b * s.Tmax + n] = tok;
      s.out_len[b] = n + 1;
      if (tok == s.eos) {
        s.finished[b] = 1;
        atomicAdd(s.n_finished, 1);
      }
    }
    tok_s = tok;
  }
  __syncthreads();
  if (!with_embed) return;
  if (first) {
    for (int d = threadIdx.x; d < e.D; d += 64) {
      const float z = 0.0f * e.sqrt_d;
      x[(size_t)b * e.D + d] = z + e.pos[d];
    }
  } else {
    const uint32_t tok = tok_s;
    for (int d = threadIdx.x; d < e.D; d += 64) x[(size_t)b * e.D + d] = embed1(e, tok, d, e.pos);
  }
}

hipError_t launch_decode_begin_step(const EmbedArgs &e, const DecodeState &s, int B, int first,
                                    int with_embed, const float *part_val, const int *part_idx,
                                    int n_parts, float *x, hipStream_t st) {
  hipLaunchKernelGGL(decode_begin_step_kernel, dim3(B), dim3(64), 0, st, e, s, first, with_embed,
                     part_val, part_idx, n_parts, x);
  return hipGetLastError();
}

__global__ __launch_bounds__(64) void embed_decoder_kernel(EmbedArgs e, const uint32_t *prev,
                                                           int first, float *x) {
  const int b = blockIdx.x;
  if (first) {
    for (int d = threadIdx.x; d < e.D; d += 64) {
      const float z = 0.0f * e.sqrt_d;
      x[(size_t)b * e.D + d] = z + e.pos[d];
    }
  } else {
    const uint32_t tok = prev[b];
    for (int d = threadIdx.x; d < e.D; d += 64) x[(size_t)b * e.D + d] = embed1(e, tok, d, e.pos);
  }
}

hipError_t launch_embed_decoder(const EmbedArgs &e, const uint32_t *prev, int B, int first,
                                float *x, hipStream_t st) {
  hipLaunchKernelGGL(embed_decoder_kernel, dim3(B), dim3(64), 0, st, e, prev, first, x);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// attention (Modules.cc:24-86): one block per (sentence, head); K and V tiles
// of that head staged in LDS once, queries spread over the block's waves.
// f32 throughout; score = alpha * (k-ascending fmaf chain); softmax in the
// portable order; out = key-ascending fmaf chain.
// ---------------------------------------------------------------------------

template <int NW>
__device__ __forceinline__ void attention_body(const AttnArgs &a, int b, int h, char *smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S, dh = a.dh, LDK = dh + 1;
  float *Ks = reinterpret_cast<float *>(smem);
  float *Vs = Ks + S * LDK;
  for (int i = tid; i < S * dh; i += 64 * NW) {
    const int j = i / dh, d = i - j * dh;
    Ks[j * LDK + d] = a.k[(size_t)(b * S + j) * a.ldk + h * dh + d];
    Vs[j * LDK + d] = a.v[(size_t)(b * S + j) * a.ldv + h * dh + d];
  }
  __syncthreads();
  // additive mask of this sentence for this lane's keys (Input.cc:49-63)
  float mask0, mask1;
  {
    const int j0 = lane, j1 = lane + 64;
    if (a.mask) {
      mask0 = j0 < S ? a.mask[(size_t)b * S + j0] : 0.0f;
      mask1 = j1 < S ? a.mask[(size_t)b * S + j1] : 0.0f;
    } else {
      const int len = checked_length(a.lengths[b], S);
      const float minus_inf = -99999999.0f;
      mask0 = (1.0f - (j0 < len ? 1.0f : 0.0f)) * minus_inf;
      mask1 = (1.0f - (j1 < len ? 1.0f : 0.0f)) * minus_inf;
    }
  }
  const int j0c = lane < S ? lane : S - 1;
  const int j1c = (lane + 64) < S ? (lane + 64) : S - 1;
  const int dc = lane < dh ? lane : dh - 1;
  for (int qi = wave; qi < a.Tq; qi += NW) {
    const float qv = a.q[(size_t)(b * a.Tq + qi) * a.ldq + h * dh + dc];
    float s0 = 0.0f, s1 = 0.0f;
    for (int k = 0; k < dh; ++k) {
      const float qk = __shfl(qv, k, 64);
      s0 = __builtin_fmaf(qk, Ks[j0c * LDK + k], s0);
      s1 = __builtin_fmaf(qk, Ks[j1c * LDK + k], s1);
    }
    if (a.alpha != 1.0f) {
      s0 = a.alpha * s0;
      s1 = a.alpha * s1;
    }
    s0 = s0 + mask0;
    s1 = s1 + mask1;
    const float lowest = -3.402823466e+38f;
    if (lane >= S) s0 = lowest;
    if (lane + 64 >= S) s1 = lowest;
    const float m = wave_max(fmaxf(s0, s1));
    const float e0 = lane < S ? exp_p(s0 - m) : 0.0f;
    const float e1 = (lane + 64) < S ? exp_p(s1 - m) : 0.0f;
    const float sum = wave_sum(e0 + e1);
    const float p0 = e0 / sum, p1 = e1 / sum;
    if (a.attn) {
      float *ap = a.attn + ((size_t)(b * a.H + h) * a.Tq + qi) * S;
      if (lane < S) ap[lane] = p0;
      if (lane + 64 < S) ap[lane + 64] = p1;
    }
    if (a.align && h == 0 && !a.finished[b]) {  // update_alignment, Model.cc:84-108
      const int len = checked_length(a.lengths[b], S);
      const uint32_t t = a.out_len[b];
      if ((int)t < a.Tmax) {
        float *al = a.align + ((size_t)b * a.Tmax + t) * S;
        if (lane < len) al[lane] = p0;
        if (lane + 64 < len) al[lane + 64] = p1;
      }
    }
    float o = 0.0f;
    for (int j = 0; j < S; ++j) {
      const float pj = __shfl(j < 64 ? p0 : p1, j & 63, 64);
      o = __builtin_fmaf(pj, Vs[j * LDK + dc], o);
    }
    if (lane < dh) a.out[(size_t)(b * a.Tq + qi) * a.ldo + h * dh + lane] = o;
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void attention_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attention_body<NW>(a, blockIdx.x / a.H, blockIdx.x % a.H, smem);
}

// Self-attention with Tq == S <= 32 and d_head 32 / 64 on the f32 matrix cores:
// one wave per (sentence, head), the formulation of the persistent encoder
// (encode_fused.hip): S^T = K Q^T as a chain of v_mfma_f32_32x32x2_f32 over
// ascending d (bit-identical to the ascending fmaf chain of attention_body,
// tools/probe_mfma_f32.py), the canonical 32-key butterfly on the accumulator
// layout, O = P V over ascending keys. Q and K rows are staged through a
// per-wave LDS region (row stride DH + 2 floats: conflict-free operand reads);
// V operands come straight from global memory (128 B per half-wave).
template <int DH>
__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef float v16f __attribute__((ext_vector_type(16)));
  constexpr int LDH = DH + 2;
  constexpr int NDB = DH / 32;  // 32-column blocks of the output
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int job = blockIdx.x * 4 + wave;
  if (job >= a.B * a.H) return;  // waves are independent: no workgroup barrier below
  const int b = job / a.H, h = job - b * a.H;
  const int S = a.S;
  float *Qs = reinterpret_cast<float *>(smem) + (size_t)wave * 2 * 32 * LDH;
  float *Ks = Qs + 32 * LDH;
  const int n = lane & 31, hh = lane >> 5;
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;

  // V operand of step i (keys 2 i + hh), requested first: it is needed last
  float vv[NDB][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = 2 * i + hh;
    const float *vp = a.v + (size_t)(b * S + (key < S ? key : S - 1)) * a.ldv + h * DH + n;
#pragma unroll
    for (int db = 0; db < NDB; ++db) vv[db][i] = vp[32 * db];
  }
  {
    constexpr int LPR = DH / 4, RPI = 64 / LPR;  // lanes per row, rows per load instruction
    float4 q4[32 / RPI], k4[32 / RPI];
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int r = i * RPI + lane / LPR, c = (lane % LPR) * 4;
      const size_t row = (size_t)(b * S + (r < S ? r : S - 1));
      q4[i] = *reinterpret_cast<const float4 *>(a.q + row * a.ldq + h * DH + c);
      k4[i] = *reinterpret_cast<const float4 *>(a.k + row * a.ldk + h * DH + c);
    }
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int r = i * RPI + lane / LPR, c = (lane % LPR) * 4;
      float2 *qd = reinterpret_cast<float2 *>(Qs + r * LDH + c);
      float2 *kd = reinterpret_cast<float2 *>(Ks + r * LDH + c);
      qd[0] = make_float2(q4[i].x, q4[i].y);
      qd[1] = make_float2(q4[i].z, q4[i].w);
      kd[0] = make_float2(k4[i].x, k4[i].y);
      kd[1] = make_float2(k4[i].z, k4[i].w);
    }
  }
  // LDS operations of one wave execute in order; only the compiler must not move
  // the operand reads above the stores
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  v16f st = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  {
    const float *kp = Ks + n * LDH + hh, *qp = Qs + n * LDH + hh;
#pragma unroll
    for (int k0 = 0; k0 < DH; k0 += 2)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[k0], qp[k0], st, 0, 0, 0);
  }
  const int len = a.mask ? 0 : checked_length(a.lengths[b], S);
  float sc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = 8 * (r >> 2) + 4 * hh + (r & 3);  // key of this register
    float v = st[r];
    if (a.alpha != 1.0f) v = a.alpha * v;
    float mk;
    if (a.mask)
      mk = m < S ? a.mask[(size_t)b * S + m] : 0.0f;
    else
      mk = (1.0f - (m < len ? 1.0f : 0.0f)) * minus_inf;
    v = v + mk;
    if (m >= S) v = lowest;
    sc[r] = v;
  }
  // canonical butterfly over 32 keys: masks 1, 2 = register pairs, mask 4 = the
  // other half-wave, masks 8, 16 = register groups
  float t4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    t4[g] = bf_max<32>(fmaxf(fmaxf(sc[4 * g], sc[4 * g + 1]), fmaxf(sc[4 * g + 2], sc[4 * g + 3])));
  const float mx = fmaxf(fmaxf(t4[0], t4[1]), fmaxf(t4[2], t4[3]));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = 8 * (r >> 2) + 4 * hh + (r & 3);
    sc[r] = m < S ? exp_p(sc[r] - mx) : 0.0f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
    t4[g] = bf_add<32>((sc[4 * g] + sc[4 * g + 1]) + (sc[4 * g + 2] + sc[4 * g + 3]));
  const float sum = (t4[0] + t4[1]) + (t4[2] + t4[3]);
#pragma unroll
  for (int r = 0; r < 16; ++r) sc[r] = sc[r] / sum;  // keys >= S: exactly 0
  // P operand of step i: keys 2 i (hh = 0) / 2 i + 1 (hh = 1)
  float pa[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const slimt_u2 s01 = __builtin_amdgcn_permlane32_swap(__float_as_int(sc[4 * g + 0]),
                                                          __float_as_int(sc[4 * g + 1]), false, false);
    const slimt_u2 s23 = __builtin_amdgcn_permlane32_swap(__float_as_int(sc[4 * g + 2]),
                                                          __float_as_int(sc[4 * g + 3]), false, false);
    pa[4 * g + 0] = __int_as_float(s01.x);  // keys 8 g + 0, 8 g + 1
    pa[4 * g + 1] = __int_as_float(s23.x);  // keys 8 g + 2, 8 g + 3
    pa[4 * g + 2] = __int_as_float(s01.y);  // keys 8 g + 4, 8 g + 5
    pa[4 * g + 3] = __int_as_float(s23.y);  // keys 8 g + 6, 8 g + 7
  }
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    v16f o = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16; ++i)  // keys >= S contribute fma(0, v, o) == o
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[i], vv[db][i], o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 8 * (r >> 2) + 4 * hh + (r & 3);  // query of this register
      if (m < S) a.out[(size_t)(b * S + m) * a.ldo + h * DH + 32 * db + n] = o[r];
    }
  }
}

template <int DH>
static hipError_t launch_attention_mfma(const AttnArgs &a, hipStream_t st) {
  const size_t lds = (size_t)4 * 2 * 32 * (DH + 2) * sizeof(float);
  auto k = attention_mfma_kernel<DH>;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.B * a.H + 3) / 4), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs &a, hipStream_t st) {
  if (a.S < 1 || a.S > 128 || a.dh < 1 || a.dh > 64) return hipErrorInvalidValue;
  const auto aligned16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (a.Tq == a.S && a.S <= 32 && (a.dh == 32 || a.dh == 64) && !a.attn && !a.align &&
      a.ldq % 4 == 0 && a.ldk % 4 == 0 && aligned16(a.q) && aligned16(a.k))
    return a.dh == 32 ? launch_attention_mfma<32>(a, st) : launch_attention_mfma<64>(a, st);
  const size_t lds = 2 * (size_t)a.S * (a.dh + 1) * sizeof(float);
  const dim3 grid(a.B * a.H);
  if (a.Tq == 1)
    hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(64), lds, st, a);
  else
    hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stand-alone float ops (op-level C ABI; the engine uses the fused forms)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void layer_norm_kernel(const float *x, const float *scale,
                                                         const float *bias, float eps, int rows,
                                                         int cols, float *y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < rows)
    wave_layer_norm_row(x + (size_t)row * cols, scale, bias, eps, cols, y + (size_t)row * cols,
                        threadIdx.x & 63);
}

// LayerNorm of rows of 64 DPL columns held in registers, one wave per row, with an
// optional int8 copy quantised for the next affine (x and y may alias).
template <int DPL>
__global__ __launch_bounds__(256) void layer_norm_q_kernel(const float *x, const float *scale,
                                                           const float *bias, float eps, int rows,
                                                           float *y, int8_t *y8, float aq8) {
  constexpr int D = 64 * DPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *xr = x + (size_t)row * D;
  float v[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) v[i] = xr[lane + 64 * i];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < DPL; ++i) s += v[i];
  s = wave_sum(s);
  const float mean = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    const float d = v[i] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  const float sigma = __builtin_sqrtf(q / (float)D + eps);
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    const float t = (v[i] - mean) / sigma;
    const float sc = scale[lane + 64 * i] * t;
    const float o = sc + bias[lane + 64 * i];
    y[(size_t)row * D + lane + 64 * i] = o;
    if (y8) y8[(size_t)row * D + lane + 64 * i] = (int8_t)quantize1(o, aq8);
  }
}

hipError_t launch_layer_norm_q(const float *x, const float *scale, const float *bias, float eps,
                               int rows, int cols, float *y, int8_t *y8, float aq8,
                               hipStream_t st) {
  const dim3 grid((rows + 3) / 4);
#define SLIMT_LNQ_CASE(DPL_)                                                                    \
  if (cols == 64 * DPL_) {                                                                      \
    hipLaunchKernelGGL(layer_norm_q_kernel<DPL_>, grid, dim3(256), 0, st, x, scale, bias, eps,  \
                       rows, y, y8, aq8);                                                       \
    return hipGetLastError();                                                                   \
  }
  SLIMT_LNQ_CASE(1) SLIMT_LNQ_CASE(2) SLIMT_LNQ_CASE(4) SLIMT_LNQ_CASE(8)
#undef SLIMT_LNQ_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_layer_norm(const float *x, const float *scale, const float *bias, float eps,
                             int rows, int cols, float *y, hipStream_t st) {
  hipLaunchKernelGGL(layer_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, scale, bias,
                     eps, rows, cols, y);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void softmax_kernel(const float *x, int rows, int cols,
                                                      float *y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *xr = x + (size_t)row * cols;
  float *yr = y + (size_t)row * cols;
  float m = -3.402823466e+38f;
  for (int i = lane; i < cols; i += 64) m = fmaxf(m, xr[i]);
  m = wave_max(m);
  float s = 0.0f;
  for (int i = lane; i < cols; i += 64) s += exp_p(xr[i] - m);
  s = wave_sum(s);
  for (int i = lane; i < cols; i += 64) yr[i] = exp_p(xr[i] - m) / s;
}

hipError_t launch_softmax(const float *x, int rows, int cols, float *y, hipStream_t st) {
  hipLaunchKernelGGL(softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, rows, cols, y);
  return hipGetLastError();
}

__global__ void highway_kernel(const float *x, const float *y, const float *g, size_t n,
                               float *out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sg = sigmoid_p(g[i]);
  const float t1 = sg * x[i];
  const float t2 = (1.0f - sg) * y[i];
  out[i] = t1 + t2;
}

hipError_t launch_highway(const float *x, const float *y, const float *g, size_t n, float *out,
                          hipStream_t st) {
  hipLaunchKernelGGL(highway_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, g,
                     n, out);
  return hipGetLastError();
}

// transpose_3120 (TensorOps.cc:98-121): [B, d2, d1, d0] -> [B, d1, d2, d0]
__global__ void transpose_heads_kernel(const float *in, int d2, int d1, int d0, float *out) {
  const int b = blockIdx.z, i2 = blockIdx.y, i1 = blockIdx.x;
  const float *src = in + (((size_t)b * d2 + i2) * d1 + i1) * d0;
  float *dst = out + (((size_t)b * d1 + i1) * d2 + i2) * d0;
  for (int i = threadIdx.x; i < d0; i += blockDim.x) dst[i] = src[i];
}

hipError_t launch_transpose_heads(const float *in, int B, int d2, int d1, int d0, float *out,
                                  hipStream_t st) {
  hipLaunchKernelGGL(transpose_heads_kernel, dim3(d1, d2, B), dim3(64), 0, st, in, d2, d1, d0, out);
  return hipGetLastError();
}

// ---- centres of the tight K/V cache form (kernels.h, FusedDecodeArgs::kv_centre) ----------------------------------------
// One calibration batch is cached in the f32 form (float(accS), exact): K [sentence][D/4][key][4], V [sentence][key][D]
// per layer and projection. Column sums over all its rows as 64-bit integers (exact, order-free), then
// centre = floor(sum / rows + 1/2) in integer arithmetic: the same numbers on every run.
__global__ __launch_bounds__(256) void kv_centre_sum_kernel(const float *kv, int B, int S, int D, unsigned long long *sums) {
  const int lp = blockIdx.x, d = threadIdx.x + 256 * blockIdx.z;  // lp = 2 * layer + (0: K, 1: V)
  if (d >= D) return;
  const float *base = kv + (size_t)lp * B * S * D;
  long long acc = 0;
  for (int r = blockIdx.y; r < B * S; r += gridDim.y) {
    const int b = r / S, key = r - b * S;
    const float v = (lp & 1) ? base[(size_t)r * D + d] : base[(((size_t)b * (D / 4) + (d >> 2)) * S + key) * 4 + (d & 3)];
    acc += (long long)v;
  }
  atomicAdd(sums + (size_t)lp * D + d, (unsigned long long)acc);
}

__global__ void kv_centre_finish_kernel(const unsigned long long *sums, long long rows, int n, int *centre) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long num = 2 * (long long)sums[i] + rows, den = 2 * rows;
  long long q = num / den;
  if (num % den < 0) q -= 1;  // floor
  centre[i] = (int)q;
}

hipError_t launch_kv_centres(const float *kv, int Ld, int B, int S, int D, unsigned long long *sums, int *centre, hipStream_t st) {
  hipError_t e = hipMemsetAsync(sums, 0, (size_t)Ld * 2 * D * 8, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kv_centre_sum_kernel, dim3(Ld * 2, 64, (D + 255) / 256), dim3(256), 0, st, kv, B, S, D, sums);
  hipLaunchKernelGGL(kv_centre_finish_kernel, dim3((Ld * 2 * D + 255) / 256), dim3(256), 0, st, sums, (long long)B * S, Ld * 2 * D, centre);
  return hipGetLastError();
}

}  // namespace slimt_hip
