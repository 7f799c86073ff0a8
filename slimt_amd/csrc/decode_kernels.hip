// Decode-step kernels (M = batch rows): latency-optimised forms of the
// generic kernels in kernels.hip, same arithmetic bit for bit.
//
// A decode step is ~12 tiny dependent GEMMs (16 x 256 x 256 per MFMA row
// tile); what costs time is the chain of memory round trips, not FLOPs. So:
//  * every GEMM is split over N across many workgroups (>= 64 per launch);
//    LayerNorm therefore moves to the CONSUMER: a producer stores the pre-LN
//    row (y + residual), every consumer normalises its 16 rows in its
//    prologue (cheap: 16 x D floats) -- same canonical reduction order;
//  * each wave issues ALL its weight-fragment loads (pre-tiled, 1 KiB per
//    MFMA operand, fully coalesced) before touching the activations, so a
//    kernel is one memory round trip deep instead of one per k-step;
//  * the Q projection of the cross-attention is fused into the attention
//    kernel (one workgroup = 16 sentences x 1 head, 16 waves), K/V rows are
//    prefetched while LayerNorm + the Q GEMM run, and the attention output is
//    re-quantised in place for the O projection.
//
// Reference semantics: slimt/Modules.cc:190-259,287-319 (DecoderLayer,
// SSRU, Attention), slimt/Transformer.cc:120-183 (Decoder::step).
#include "device_common.h"
#include "kernels.h"

namespace slimt_hip {

constexpr int MAXDPL = 8;  // D <= 512: elements per lane of one f32 row

// ---- 16 f32 rows -> (optional LayerNorm) -> registers ---------------------
// Each wave owns rows [w*RPW, w*RPW + RPW). Lane l holds elements l + 64 i.

template <int RPW>
struct Rows {
  float v[RPW][MAXDPL];
};

template <int RPW>
__device__ __forceinline__ void rows_load(Rows<RPW> &r, const float *x, int B, int D, int row0,
                                          int lane) {
  const int dpl = D >> 6;
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int row = row0 + k;
#pragma unroll
    for (int i = 0; i < MAXDPL; ++i) {
      float t = 0.0f;
      if (i < dpl && row < B) t = x[(size_t)row * D + lane + 64 * i];
      r.v[k][i] = t;
    }
  }
}

// canonical LayerNorm (device_common.h: wave_layer_norm_row) on registers
template <int RPW>
__device__ __forceinline__ void rows_layer_norm(Rows<RPW> &r, const float *scale,
                                                const float *bias, float eps, int D, int lane) {
  const int dpl = D >> 6;
  float sc[MAXDPL], bi[MAXDPL];
#pragma unroll
  for (int i = 0; i < MAXDPL; ++i) {
    sc[i] = i < dpl ? scale[lane + 64 * i] : 0.0f;
    bi[i] = i < dpl ? bias[lane + 64 * i] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXDPL; ++i)
      if (i < dpl) s += r.v[k][i];
    s = wave_sum(s);
    const float mean = s / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXDPL; ++i)
      if (i < dpl) {
        const float d = r.v[k][i] - mean;
        q += d * d;
      }
    q = wave_sum(q);
    const float sigma = __builtin_sqrtf(q / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXDPL; ++i)
      if (i < dpl) {
        const float t = (r.v[k][i] - mean) / sigma;
        const float m = sc[i] * t;
        r.v[k][i] = m + bi[i];
      }
  }
}

template <int RPW>
__device__ __forceinline__ void rows_quantize_to_lds(const Rows<RPW> &r, float aq, char *A_lds,
                                                     int lda, int rl0, int D, int lane) {
  const int dpl = D >> 6;
#pragma unroll
  for (int k = 0; k < RPW; ++k)
#pragma unroll
    for (int i = 0; i < MAXDPL; ++i)
      if (i < dpl) A_lds[(rl0 + k) * lda + lane + 64 * i] = (char)quantize1(r.v[k][i], aq);
}

template <int RPW>
__device__ __forceinline__ void rows_store_lds(const Rows<RPW> &r, float *buf, int ldr, int rl0,
                                               int D, int lane) {
  const int dpl = D >> 6;
#pragma unroll
  for (int k = 0; k < RPW; ++k)
#pragma unroll
    for (int i = 0; i < MAXDPL; ++i)
      if (i < dpl) buf[(rl0 + k) * ldr + lane + 64 * i] = r.v[k][i];
}

// ---- weight fragment prefetch ---------------------------------------------

template <int PF, int NT>
struct BFrag {
  v4i f[PF][NT];
};

template <int PF, int NT>
__device__ __forceinline__ void bfrag_load(BFrag<PF, NT> &b, const v4i *Wp, int KS, int chunk,
                                           int nt0, int n_tiles, int lane) {
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const int ks = chunk * PF + p;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ntile = nt0 + nt;
      v4i t = {0, 0, 0, 0};
      if (ks < KS && ntile < n_tiles) t = Wp[((size_t)ntile * KS + ks) * 64 + lane];
      b.f[p][nt] = t;
    }
  }
}

template <int PF, int NT>
__device__ __forceinline__ void bfrag_mma(const BFrag<PF, NT> &b, const char *A_lds, int lda,
                                          int KS, int chunk, int lr, int lg, v4i (&acc)[NT]) {
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const int ks = chunk * PF + p;
    if (ks < KS) {
      const v4i af = *reinterpret_cast<const v4i *>(A_lds + lr * lda + ks * 64 + lg * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, b.f[p][nt], acc[nt], 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------
// dgemm: one 16-row tile x (4 waves * NT * 16) columns per workgroup
// ---------------------------------------------------------------------------

template <int PF, int NT, int EPI, bool A_I8>
__global__ __launch_bounds__(256) void dgemm_kernel(DGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int nt0 = (blockIdx.y * 4 + wave) * NT;
  const int K = a.w.K, KS = K >> 6, lda = K + 16, D = a.D;
  const int n_tiles = a.w.n_tiles;
  const int n_chunks = (KS + PF - 1) / PF;
  char *A_lds = smem;
  float *rowbuf = reinterpret_cast<float *>(smem + 16 * lda);
  const int ldr = D + 4;
  const v4i *Wp = reinterpret_cast<const v4i *>(a.w.Wp);
  const bool has_res = a.res.x != nullptr;

  // 1. activations first (they are consumed first; vmcnt retires in order)
  Rows<4> xa, xr;
  constexpr int MAXP = 8;  // int8 A: K <= 2048
  v4i a8[MAXP];
  const int units_per_row = K >> 4;       // 16-byte units
  const int passes = (16 * units_per_row + 255) >> 8;
  if constexpr (A_I8) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      v4i t = {0, 0, 0, 0};
      const int u = p * 256 + tid;
      if (p < passes && u < 16 * units_per_row) {
        const int r = u / units_per_row, c = u - r * units_per_row;
        if (m0 + r < a.B) t = *reinterpret_cast<const v4i *>(a.a_i8 + (size_t)(m0 + r) * K + c * 16);
      }
      a8[p] = t;
    }
  } else {
    rows_load<4>(xa, a.a.x, a.B, D, m0 + wave * 4, lane);
  }
  if (has_res) rows_load<4>(xr, a.res.x, a.B, D, m0 + wave * 4, lane);

  // 2. all weight fragments of the first two K chunks in flight
  BFrag<PF, NT> bA, bB;
  bfrag_load<PF, NT>(bA, Wp, KS, 0, nt0, n_tiles, lane);
  if (n_chunks > 1) bfrag_load<PF, NT>(bB, Wp, KS, 1, nt0, n_tiles, lane);

  // 3. LayerNorm / quantise into LDS
  if constexpr (A_I8) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int u = p * 256 + tid;
      if (p < passes && u < 16 * units_per_row) {
        const int r = u / units_per_row, c = u - r * units_per_row;
        *reinterpret_cast<v4i *>(A_lds + r * lda + c * 16) = a8[p];
      }
    }
  } else {
    if (a.a.ln_scale) rows_layer_norm<4>(xa, a.a.ln_scale, a.a.ln_bias, a.eps, D, lane);
    rows_quantize_to_lds<4>(xa, a.w.a_quant, A_lds, lda, wave * 4, D, lane);
  }
  if (has_res) {
    if (a.res.ln_scale) rows_layer_norm<4>(xr, a.res.ln_scale, a.res.ln_bias, a.eps, D, lane);
    rows_store_lds<4>(xr, rowbuf, ldr, wave * 4, D, lane);
  }
  __syncthreads();

  // 4. MFMA over K, next chunks prefetched behind the current one
  v4i acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = v4i{0, 0, 0, 0};
  for (int c = 0; c < n_chunks; c += 2) {
    bfrag_mma<PF, NT>(bA, A_lds, lda, KS, c, lr, lg, acc);
    if (c + 2 < n_chunks) bfrag_load<PF, NT>(bA, Wp, KS, c + 2, nt0, n_tiles, lane);
    if (c + 1 < n_chunks) {
      bfrag_mma<PF, NT>(bB, A_lds, lda, KS, c + 1, lr, lg, acc);
      if (c + 3 < n_chunks) bfrag_load<PF, NT>(bB, Wp, KS, c + 3, nt0, n_tiles, lane);
    }
  }

  // 5. epilogue: y = float(acc + 127 colsum) * u + pb   (Intgemm.inl.cc:146-153)
  const float u = a.w.u;
  if constexpr (EPI == EPI_PLAIN || EPI == EPI_RELU_Q) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ntile = nt0 + nt;
      if (ntile >= n_tiles) continue;
      const int col = ntile * 16 + lr;
      if (col >= a.w.N) continue;
      const int cs = a.w.colsum[col];
      const float pb = a.w.pb[col];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = lg * 4 + r, row = m0 + rl;
        if (row >= a.B) continue;
        float v = (float)(acc[nt][r] + 127 * cs) * u;
        v = v + pb;
        if constexpr (EPI == EPI_PLAIN) {
          if (has_res) v = v + rowbuf[rl * ldr + col];
          a.y[(size_t)row * a.ldy + col] = v;
        } else {
          v = v > 0.0f ? v : 0.0f;
          a.y_i8[(size_t)row * a.ldy8 + col] = (int8_t)quantize1(v, a.a_quant_out);
        }
      }
    }
  } else if constexpr (EPI == EPI_ARGMAX) {
    __syncthreads();  // A_lds is reused as the reduction buffer
    float *red_v = reinterpret_cast<float *>(smem);
    int *red_i = reinterpret_cast<int *>(red_v + 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bv = -3.402823466e+38f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ntile = nt0 + nt;
        const int col = ntile * 16 + lr;
        if (ntile < n_tiles && col < a.w.N) {
          float v = (float)(acc[nt][r] + 127 * a.w.colsum[col]) * u;
          v = v + a.w.pb[col];
          if (v > bv || (v == bv && col < bi)) {
            bv = v;
            bi = col;
          }
        }
      }
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
        red_v[wave * 16 + lg * 4 + r] = bv;
        red_i[wave * 16 + lg * 4 + r] = bi;
      }
    }
    __syncthreads();
    if (tid < 16) {
      float bv = red_v[tid];
      int bi = red_i[tid];
      for (int w = 1; w < 4; ++w) {
        const float ov = red_v[w * 16 + tid];
        const int oi = red_i[w * 16 + tid];
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      const int row = m0 + tid;
      if (row < a.B) {
        a.part_val[(size_t)row * a.n_parts + blockIdx.y] = bv;
        a.part_idx[(size_t)row * a.n_parts + blockIdx.y] = bi;
      }
    }
  }
}

void dgemm_config(int K, int N, int B, int *pf_out, int *nt_out) {
  const int KS = K / 64;
  int pf = KS <= 4 ? 4 : (KS <= 8 ? 8 : 16);
  int nt_max = 16 / pf;  // register budget: 2 * PF * NT * 4 VGPRs of fragments
  const int row_blocks = (B + 15) / 16;
  int nt = 1;
  for (int cand = nt_max; cand >= 1; cand >>= 1) {
    const int col_blocks = (N + 64 * cand - 1) / (64 * cand);
    if (row_blocks * col_blocks >= 128 || cand == 1) {
      nt = cand;
      break;
    }
  }
  *pf_out = pf;
  *nt_out = nt;
}

int dgemm_col_blocks(int K, int N, int B) {
  int pf, nt;
  dgemm_config(K, N, B, &pf, &nt);
  return (N + 64 * nt - 1) / (64 * nt);
}

template <int PF, int NT>
static hipError_t launch_dgemm_t(const DGemmArgs &a, int epi, dim3 grid, size_t lds,
                                 hipStream_t st) {
  const bool i8 = a.a_i8 != nullptr;
  if (epi == EPI_PLAIN && i8)
    hipLaunchKernelGGL((dgemm_kernel<PF, NT, EPI_PLAIN, true>), grid, dim3(256), lds, st, a);
  else if (epi == EPI_PLAIN)
    hipLaunchKernelGGL((dgemm_kernel<PF, NT, EPI_PLAIN, false>), grid, dim3(256), lds, st, a);
  else if (epi == EPI_RELU_Q && !i8)
    hipLaunchKernelGGL((dgemm_kernel<PF, NT, EPI_RELU_Q, false>), grid, dim3(256), lds, st, a);
  else if (epi == EPI_ARGMAX && !i8)
    hipLaunchKernelGGL((dgemm_kernel<PF, NT, EPI_ARGMAX, false>), grid, dim3(256), lds, st, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_dgemm(const DGemmArgs &a, int epilogue, hipStream_t st) {
  const int K = a.w.K, N = a.w.N, D = a.D;
  if (K % 64 || K <= 0 || D % 64 || D <= 0 || D > 64 * MAXDPL || a.B <= 0) return hipErrorInvalidValue;
  if (a.a_i8 ? (K > 2048) : (K != D)) return hipErrorInvalidValue;
  if (a.res.x && N != D) return hipErrorInvalidValue;
  int pf, nt;
  dgemm_config(K, N, a.B, &pf, &nt);
  const int col_blocks = (N + 64 * nt - 1) / (64 * nt);
  if (epilogue == EPI_ARGMAX && a.n_parts != col_blocks) return hipErrorInvalidValue;
  const dim3 grid((a.B + 15) / 16, col_blocks);
  size_t lds = 16 * (size_t)(K + 16) + 16 * (size_t)(D + 4) * sizeof(float);
  if (lds < 1024) lds = 1024;
#define SLIMT_DG_CASE(PF_, NT_) \
  if (pf == PF_ && nt == NT_) return launch_dgemm_t<PF_, NT_>(a, epilogue, grid, lds, st);
  SLIMT_DG_CASE(4, 1) SLIMT_DG_CASE(4, 2) SLIMT_DG_CASE(4, 4)
  SLIMT_DG_CASE(8, 1) SLIMT_DG_CASE(8, 2) SLIMT_DG_CASE(16, 1)
#undef SLIMT_DG_CASE
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// SSRU (Modules.cc:190-235), N-split: a workgroup computes matching column
// slices of f = affine(Wf,bf)(x) and Wx = dot(W)(x), so the highway gate is
// local. Output h_pre = x + relu(c') (pre-LN; the consumer normalises).
// ---------------------------------------------------------------------------

template <int PF>
__global__ __launch_bounds__(256) void dssru_kernel(DSsruArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int ntile = blockIdx.y * 4 + wave;
  const int D = a.D, KS = D >> 6, lda = D + 16, ldr = D + 4;
  char *Af = smem;
  char *Aw = smem + 16 * lda;
  float *rowbuf = reinterpret_cast<float *>(smem + 2 * 16 * lda);
  const int n_tiles = D / 16;

  Rows<4> x;
  rows_load<4>(x, a.x.x, a.B, D, m0 + wave * 4, lane);
  BFrag<PF, 1> bf, bw;
  bfrag_load<PF, 1>(bf, reinterpret_cast<const v4i *>(a.wf.Wp), KS, 0, ntile, n_tiles, lane);
  bfrag_load<PF, 1>(bw, reinterpret_cast<const v4i *>(a.w.Wp), KS, 0, ntile, n_tiles, lane);
  // state column slice for the epilogue, also early
  float cst[4];
  const int col = ntile * 16 + lr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + lg * 4 + r;
    cst[r] = (row < a.B && ntile < n_tiles) ? a.state[(size_t)row * D + col] : 0.0f;
  }
  if (a.x.ln_scale) rows_layer_norm<4>(x, a.x.ln_scale, a.x.ln_bias, a.eps, D, lane);
  rows_quantize_to_lds<4>(x, a.wf.a_quant, Af, lda, wave * 4, D, lane);
  rows_quantize_to_lds<4>(x, a.w.a_quant, Aw, lda, wave * 4, D, lane);
  rows_store_lds<4>(x, rowbuf, ldr, wave * 4, D, lane);
  __syncthreads();
  if (ntile >= n_tiles) return;
  v4i accf[1] = {v4i{0, 0, 0, 0}}, accw[1] = {v4i{0, 0, 0, 0}};
  bfrag_mma<PF, 1>(bf, Af, lda, KS, 0, lr, lg, accf);
  bfrag_mma<PF, 1>(bw, Aw, lda, KS, 0, lr, lg, accw);
  const int csf = a.wf.colsum[col], csw = a.w.colsum[col];
  const float pbf = a.wf.pb[col], pbw = a.w.pb[col];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rl = lg * 4 + r, row = m0 + rl;
    if (row >= a.B) continue;
    float f = (float)(accf[0][r] + 127 * csf) * a.wf.u;
    f = f + pbf;
    float wx = (float)(accw[0][r] + 127 * csw) * a.w.u;
    wx = wx + pbw;
    const float sg = sigmoid_p(f);  // highway(c, Wx, f), TensorOps.cc:674-678
    const float t1 = sg * cst[r];
    const float t2 = (1.0f - sg) * wx;
    const float cn = t1 + t2;
    const size_t o = (size_t)row * D + col;
    a.state[o] = cn;
    const float y = cn > 0.0f ? cn : 0.0f;
    a.h_pre[o] = rowbuf[rl * ldr + col] + y;  // x + relu(c'), Modules.cc:230
  }
}

hipError_t launch_dssru(const DSsruArgs &a, hipStream_t st) {
  const int D = a.D;
  if (D % 64 || D <= 0 || D > 512 || a.B <= 0) return hipErrorInvalidValue;
  const dim3 grid((a.B + 15) / 16, (D / 16 + 3) / 4);
  size_t lds = 2 * 16 * (size_t)(D + 16) + 16 * (size_t)(D + 4) * sizeof(float);
  if (D <= 256)
    hipLaunchKernelGGL(dssru_kernel<4>, grid, dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL(dssru_kernel<8>, grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// cross-attention with fused LayerNorm + Q projection (Modules.cc:287-306):
// one workgroup = 16 sentences x 1 head, one wave per sentence.
// ---------------------------------------------------------------------------

template <int PF, int DH>
__global__ __launch_bounds__(1024) void dqattn_kernel(DQAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QT = DH / 16;  // 16-column tiles of this head's Q slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * 16, h = blockIdx.y;
  const int D = a.D, KS = D >> 6, lda = D + 16, S = a.S;
  char *A_lds = smem;
  float *qbuf = reinterpret_cast<float *>(smem + 16 * lda);  // [16][DH]
  const int b = m0 + wave;
  const bool live = b < a.B;
  const int bc = live ? b : a.B - 1;

  // 1. this sentence's normalised decoder state (A row of the Q GEMM)
  Rows<1> x;
  rows_load<1>(x, a.x.x, a.B, D, b, lane);
  // 2. Q weight fragments (only the waves that run the MFMA)
  BFrag<PF, 1> bq;
  if (wave < QT)
    bfrag_load<PF, 1>(bq, reinterpret_cast<const v4i *>(a.wq.Wp), KS, 0, h * QT + wave, D / 16, lane);
  // 3. K rows of this lane's keys, independent of Q: issue now
  const int j0 = lane < S ? lane : S - 1;
  const int j1 = (lane + 64) < S ? (lane + 64) : S - 1;
  const float *kb = a.k + ((size_t)bc * a.H + h) * (DH / 4) * S * 4;  // [dh/4][S][4]
  constexpr int KPF = (DH < 32 ? DH : 32) / 4;  // float4s prefetched before Q is known
  float4 k0[KPF];
#pragma unroll
  for (int i = 0; i < KPF; ++i)
    k0[i] = *reinterpret_cast<const float4 *>(kb + ((size_t)i * S + j0) * 4);
  // first 16 V rows for this lane's output column
  const int dc = lane < DH ? lane : DH - 1;
  const float *vb = a.v + (size_t)bc * S * a.ldv + h * DH + dc;
  constexpr int VPF = DH > 32 ? 4 : 16;  // V rows requested before Q is known (register budget)
  float v0[VPF];
#pragma unroll
  for (int j = 0; j < VPF; ++j) v0[j] = vb[(size_t)(j < S ? j : S - 1) * a.ldv];

  if (a.x.ln_scale) rows_layer_norm<1>(x, a.x.ln_scale, a.x.ln_bias, a.eps, D, lane);
  rows_quantize_to_lds<1>(x, a.wq.a_quant, A_lds, lda, wave, D, lane);
  __syncthreads();
  if (a.q_given) {  // the caller's projected query
    if (lane < DH) qbuf[wave * DH + lane] = a.q_given[(size_t)bc * D + h * DH + lane];
  } else if (wave < QT) {
    v4i acc[1] = {v4i{0, 0, 0, 0}};
    bfrag_mma<PF, 1>(bq, A_lds, lda, KS, 0, lr, lg, acc);
    const int col = (h * QT + wave) * 16 + lr;
    const int cs = a.wq.colsum[col];
    const float pb = a.wq.pb[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = (float)(acc[0][r] + 127 * cs) * a.wq.u;
      v = v + pb;
      qbuf[(lg * 4 + r) * DH + wave * 16 + lr] = v;
    }
  }
  __syncthreads();
  if (!live) return;

  // 4. scores in the hoisted (PORTABLE) order (decode_fused.hip, unpack24f; oracle cross_attention_portable):
  // the cache holds float(accS); t_j = k-ascending fmaf chain, s_j = alpha * fmaf(t_j, uK, c_h) + mask with
  // c_h = canonical row sum of q_d * pbK[d] over the head's columns (one per lane; lanes past the head hold +0)
  const float *qrow = qbuf + wave * DH;
  const float ch = wave_sum(lane < DH ? qrow[dc] * a.pbk[h * DH + dc] : 0.0f);
  const bool lit = a.literal;  // (uniform) the reference's sequence: every cached value dequantised first, k = float(accS) u + pb
  auto dq4 = [&](float4 v, int i) {
    if (lit) {
      const float4 pb4 = *reinterpret_cast<const float4 *>(a.pbk + h * DH + 4 * i);
      v.x = v.x * a.uk; v.y = v.y * a.uk; v.z = v.z * a.uk; v.w = v.w * a.uk;
      v.x = v.x + pb4.x; v.y = v.y + pb4.y; v.z = v.z + pb4.z; v.w = v.w + pb4.w;
    }
    return v;
  };
  float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const float4 q4 = *reinterpret_cast<const float4 *>(qrow + 4 * i);
    float4 k4;
    if (i < KPF)
      k4 = k0[i];
    else
      k4 = *reinterpret_cast<const float4 *>(kb + ((size_t)i * S + j0) * 4);
    k4 = dq4(k4, i);
    s0 = __builtin_fmaf(q4.x, k4.x, s0);
    s0 = __builtin_fmaf(q4.y, k4.y, s0);
    s0 = __builtin_fmaf(q4.z, k4.z, s0);
    s0 = __builtin_fmaf(q4.w, k4.w, s0);
  }
  if (S > 64) {
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      const float4 q4 = *reinterpret_cast<const float4 *>(qrow + 4 * i);
      const float4 k4 = dq4(*reinterpret_cast<const float4 *>(kb + ((size_t)i * S + j1) * 4), i);
      s1 = __builtin_fmaf(q4.x, k4.x, s1);
      s1 = __builtin_fmaf(q4.y, k4.y, s1);
      s1 = __builtin_fmaf(q4.z, k4.z, s1);
      s1 = __builtin_fmaf(q4.w, k4.w, s1);
    }
  }
  if (!lit) {
    s0 = __builtin_fmaf(s0, a.uk, ch);
    s1 = __builtin_fmaf(s1, a.uk, ch);
  }
  if (a.alpha != 1.0f) {
    s0 = a.alpha * s0;
    s1 = a.alpha * s1;
  }
  const int len = checked_length(a.lengths[b], S);
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  s0 = s0 + (1.0f - (lane < len ? 1.0f : 0.0f)) * minus_inf;
  s1 = s1 + (1.0f - ((lane + 64) < len ? 1.0f : 0.0f)) * minus_inf;
  const float lowest = -3.402823466e+38f;
  if (lane >= S) s0 = lowest;
  if (lane + 64 >= S) s1 = lowest;
  const float m = wave_max(fmaxf(s0, s1));
  const float e0 = lane < S ? exp_p(s0 - m) : 0.0f;
  const float e1 = (lane + 64) < S ? exp_p(s1 - m) : 0.0f;
  const float sum = wave_sum(e0 + e1);
  const float p0 = e0 / sum, p1 = e1 / sum;
  const float P = wave_sum(p0 + p1);  // P_h
  if (a.attn) {
    float *ap = a.attn + ((size_t)b * a.H + h) * S;
    if (lane < S) ap[lane] = p0;
    if (lane + 64 < S) ap[lane + 64] = p1;
  }
  if (a.align && h == 0 && !a.finished[b]) {  // update_alignment, Model.cc:84-108
    const uint32_t t = a.out_len[b];
    if ((int)t < a.Tmax) {
      float *al = a.align + ((size_t)b * a.Tmax + t) * S;
      if (lane < len) al[lane] = p0;
      if (lane + 64 < len) al[lane + 64] = p1;
    }
  }
  // 5. out[d] = fmaf(w_d, uV, pbV[d] * P_h), w_d = key-ascending fmaf chain of p[j] * float(accV[j][d])
  const float pbv_d = a.pbv[h * DH + dc];
  auto dqv = [&](float v) {  // literal: v = float(accS) u + pb per cached value
    if (lit) {
      v = v * a.uv;
      v = v + pbv_d;
    }
    return v;
  };
  float o = 0.0f;
#pragma unroll
  for (int j = 0; j < VPF; ++j) {
    if (j < S) {
      const float pj = __shfl(p0, j, 64);
      o = __builtin_fmaf(pj, dqv(v0[j]), o);
    }
  }
  if constexpr (VPF < 16) {
    float vv[16 - VPF];
#pragma unroll
    for (int j = VPF; j < 16; ++j) vv[j - VPF] = vb[(size_t)(j < S ? j : S - 1) * a.ldv];
#pragma unroll
    for (int j = VPF; j < 16; ++j) {
      if (j < S) {
        const float pj = __shfl(p0, j, 64);
        o = __builtin_fmaf(pj, dqv(vv[j - VPF]), o);
      }
    }
  }
  for (int jb = 16; jb < S; jb += 16) {
    float vv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) vv[j] = vb[(size_t)((jb + j) < S ? (jb + j) : S - 1) * a.ldv];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int jj = jb + j;
      if (jj < S) {
        const float pj = __shfl(jj < 64 ? p0 : p1, jj & 63, 64);
        o = __builtin_fmaf(pj, dqv(vv[j]), o);
      }
    }
  }
  if (!lit) o = __builtin_fmaf(o, a.uv, pbv_d * P);
  if (lane < DH) {
    if (a.out_i8)
      a.out_i8[(size_t)b * D + h * DH + lane] = (int8_t)quantize1(o, a.a_quant_out);
    if (a.out_f32) a.out_f32[(size_t)b * D + h * DH + lane] = o;
  }
}

hipError_t launch_dqattn(const DQAttnArgs &a, hipStream_t st) {
  const int D = a.D;
  if (D % 64 || D > 512 || a.H <= 0 || D % a.H) return hipErrorInvalidValue;
  const int dh = D / a.H;
  if (a.S < 1 || a.S > 128) return hipErrorInvalidValue;
  const dim3 grid((a.B + 15) / 16, a.H);
  const size_t lds = 16 * (size_t)(D + 16) + 16 * (size_t)dh * sizeof(float);
  const int pf = D <= 256 ? 4 : 8;
#define SLIMT_QA_CASE(PF_, DH_)                                                        \
  if (pf == PF_ && dh == DH_) {                                                        \
    hipLaunchKernelGGL((dqattn_kernel<PF_, DH_>), grid, dim3(1024), lds, st, a);       \
    return hipGetLastError();                                                          \
  }
  SLIMT_QA_CASE(4, 16) SLIMT_QA_CASE(4, 32) SLIMT_QA_CASE(8, 32) SLIMT_QA_CASE(8, 64)
  SLIMT_QA_CASE(4, 64)
#undef SLIMT_QA_CASE
  return hipErrorInvalidValue;
}

}  // namespace slimt_hip
