/*
 * slimt_oracle.c -- TEST INFRASTRUCTURE ONLY (see slimt_oracle.h header).
 * PARITY UNPINNED (no reference golden vectors exist; reference unbuildable).
 *
 * Plain-C restatement of slimt's int8 NMT hot path. Citations are
 * `file:line` in the reference checkout (slimt/...).
 * Compile with -ffp-contract=off: every float expression below is meant
 * literally (separate mul/add unless fmaf is written).
 */
#include "slimt_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define SO_HAVE_VNNI 1 /* runtime-dispatched, see gemm_shifted_vnni */
#endif

static int g_mode = SO_FAITHFUL;
void so_set_mode(int mode) { g_mode = mode; }
int so_get_mode(void) { return g_mode; }

/* ------------------------------------------------------------------------ */
/* scalar helpers                                                           */
/* ------------------------------------------------------------------------ */

/* PORTABLE exp: Cody-Waite reduction + degree-5 polynomial, every operation
 * an IEEE f32 op (rintf, fmaf, mul, add) so that a GPU reproduces it bit for
 * bit. Only called with x <= 0 on the hot path (softmax: x - max; sigmoid:
 * -|x|), but is correct up to 88. FAITHFUL: libm expf, which is what
 * std::exp(float) is in TensorOps.cc:33-36,296-314. */
static inline float exp_portable(float x) {
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = fmaf(p, r2, r) + 1.0f;
  int32_t ni = (int32_t)n;
  union {
    uint32_t u;
    float f;
  } s;
  s.u = (uint32_t)(ni + 127) << 23;
  return y * s.f;
}

float so_exp(float x) { return g_mode == SO_PORTABLE ? exp_portable(x) : expf(x); }

/* TensorOps.cc:33-36 */
float so_sigmoid(float x) {
  return x > 0 ? (1.0f / (1.0f + so_exp(-x))) : (so_exp(x) / (1.0f + so_exp(x)));
}

/* Row sum. FAITHFUL: sequential from 0 (TensorOps.cc:305-308,558-561).
 * PORTABLE: element i belongs to lane i%64; each lane adds its elements in
 * ascending order, then a xor-butterfly over the 64 lanes with masks
 * 1,2,4,8,16,32 (missing lanes hold +0). Every lane ends with the same bits
 * (a+b == b+a), so the GPU wave reduction is reproduced exactly. */
float so_row_sum(const float *x, size_t n) {
  if (g_mode != SO_PORTABLE) {
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) sum += x[i];
    return sum;
  }
  float lane[64];
  for (int l = 0; l < 64; l++) {
    float acc = 0.0f;
    for (size_t i = (size_t)l; i < n; i += 64) acc += x[i];
    lane[l] = acc;
  }
  for (int m = 1; m < 64; m <<= 1) {
    float nxt[64];
    for (int l = 0; l < 64; l++) nxt[l] = lane[l] + lane[l ^ m];
    memcpy(lane, nxt, sizeof(lane));
  }
  return lane[0];
}

/* ------------------------------------------------------------------------ */
/* qmm                                                                      */
/* ------------------------------------------------------------------------ */

/* Activation quantisation. intgemm/gemmology PrepareA: cvtps_epi32 under the
 * default MXCSR = round-to-nearest-EVEN, then saturate/clamp to [-127,127]
 * (Intgemm.inl.cc:29-34). Ruy's restatement uses roundf
 * (Ruy.inl.cc:9-24), differing only on exact .5 ties; the parity target is
 * the intgemm path => rintf. */
static inline int8_t quantize_one(float x, float a_quant) {
  float v = rintf(x * a_quant);
  if (v != v) return -127; /* cvtps_epi32(NaN) = INT_MIN -> saturating packs -> max(-127) */
  if (v < -127.0f) v = -127.0f;
  if (v > 127.0f) v = 127.0f;
  return (int8_t)v;
}

void so_quantize(const float *x, float a_quant, size_t n, int8_t *q) {
  for (size_t i = 0; i < n; i++) q[i] = quantize_one(x[i], a_quant);
}

/* Signed form, Ruy.inl.cc:104-107: acc[i,j] = sum_k q[i,k] * W[k,j]. */
void so_gemm_i8_signed(const int8_t *q, const int8_t *W, size_t M, size_t K,
                       size_t N, int32_t *acc) {
#pragma omp parallel for schedule(static) if (M * N * K > (1u << 22))
  for (size_t i = 0; i < M; i++) {
    const int8_t *a = q + i * K;
    for (size_t j = 0; j < N; j++) {
      const int8_t *b = W + j * K;
      int32_t s = 0;
      for (size_t k = 0; k < K; k++) s += (int32_t)a[k] * (int32_t)b[k];
      acc[i * N + j] = s;
    }
  }
}

/* Shift form, Intgemm.inl.cc:149-153 (Int8Shift::Multiply): activations are
 * stored as u8 = q + 127 and accS[i,j] = sum_k (q[i,k]+127) * W[k,j], exact in
 * int32 (|accS| <= 254*127*K < 2^31 for K <= 2048; the AVX512-VNNI kernel the
 * author pinned, scripts/run.sh:34). */
#ifdef SO_HAVE_VNNI
__attribute__((target("avx512f,avx512bw,avx512vnni"))) static void
gemm_shifted_vnni(const int8_t *q, const int8_t *W, size_t M, size_t K,
                  size_t N, int32_t *accS) {
#pragma omp parallel for schedule(static) if (M * N * K > (1u << 22))
  for (size_t i = 0; i < M; i++) {
    uint8_t ua[4096];
    for (size_t k = 0; k < K; k++) ua[k] = (uint8_t)((int)q[i * K + k] + 127);
    size_t j = 0;
    for (; j + 4 <= N; j += 4) {
      __m512i c0 = _mm512_setzero_si512(), c1 = c0, c2 = c0, c3 = c0;
      const int8_t *b0 = W + (j + 0) * K, *b1 = W + (j + 1) * K;
      const int8_t *b2 = W + (j + 2) * K, *b3 = W + (j + 3) * K;
      for (size_t k = 0; k < K; k += 64) {
        __m512i a = _mm512_loadu_si512((const void *)(ua + k));
        c0 = _mm512_dpbusd_epi32(c0, a, _mm512_loadu_si512((const void *)(b0 + k)));
        c1 = _mm512_dpbusd_epi32(c1, a, _mm512_loadu_si512((const void *)(b1 + k)));
        c2 = _mm512_dpbusd_epi32(c2, a, _mm512_loadu_si512((const void *)(b2 + k)));
        c3 = _mm512_dpbusd_epi32(c3, a, _mm512_loadu_si512((const void *)(b3 + k)));
      }
      accS[i * N + j + 0] = _mm512_reduce_add_epi32(c0);
      accS[i * N + j + 1] = _mm512_reduce_add_epi32(c1);
      accS[i * N + j + 2] = _mm512_reduce_add_epi32(c2);
      accS[i * N + j + 3] = _mm512_reduce_add_epi32(c3);
    }
    for (; j < N; j++) {
      __m512i c0 = _mm512_setzero_si512();
      const int8_t *b0 = W + j * K;
      for (size_t k = 0; k < K; k += 64) {
        __m512i a = _mm512_loadu_si512((const void *)(ua + k));
        c0 = _mm512_dpbusd_epi32(c0, a, _mm512_loadu_si512((const void *)(b0 + k)));
      }
      accS[i * N + j] = _mm512_reduce_add_epi32(c0);
    }
  }
}
#endif

void so_gemm_i8_shifted(const int8_t *q, const int8_t *W, size_t M, size_t K,
                        size_t N, int32_t *accS) {
#ifdef SO_HAVE_VNNI
  /* vpdpbusd (u8 x s8 -> s32, exact): the instruction intgemm's
   * AVX512-VNNI Int8Shift kernel is built on. */
  if (K % 64 == 0 && K <= 4096 && __builtin_cpu_supports("avx512vnni")) {
    gemm_shifted_vnni(q, W, M, K, N, accS);
    return;
  }
#endif
#pragma omp parallel for schedule(static) if (M * N * K > (1u << 22))
  for (size_t i = 0; i < M; i++) {
    const int8_t *a = q + i * K;
    for (size_t j = 0; j < N; j++) {
      const int8_t *b = W + j * K;
      int32_t s = 0;
      for (size_t k = 0; k < K; k++) s += ((int32_t)a[k] + 127) * (int32_t)b[k];
      accS[i * N + j] = s;
    }
  }
}

/* Int8Shift::PrepareBias (Intgemm.inl.cc:123-136): a row of ones times W,
 * i.e. colsum[j] = sum_k W[k,j], pushed through the callback
 * out = float(int) * mult + bias  (UnquantizeAndAddBiasAndWrite: cvt, mul,
 * add -- separate roundings). */
static void prepare_bias(const int8_t *W, size_t K, size_t N, const float *bias,
                         float a_quant, float b_quant, float *prepared) {
  float a_alpha = 127.0f / a_quant;
  float b_alpha = 127.0f / b_quant;
  float mult = (-1.0f * (a_alpha * b_alpha)) / 127.0f;
  for (size_t j = 0; j < N; j++) {
    const int8_t *b = W + j * K;
    int32_t s = 0;
    for (size_t k = 0; k < K; k++) s += (int32_t)b[k];
    float v = (float)s * mult;
    prepared[j] = v + (bias ? bias[j] : 0.0f);
  }
}

void so_affine_acc(const float *x, size_t M, size_t K, const int8_t *W,
                   size_t N, float a_quant, int32_t *accS) {
  int8_t *q = (int8_t *)malloc(M * K);
  so_quantize(x, a_quant, M * K, q);
  so_gemm_i8_shifted(q, W, M, K, N, accS);
  free(q);
}

/* Multiply callback (Intgemm.inl.cc:146-153): y = float(accS)*u + pb[j]. */
static void dequant_rows(const int32_t *accS, const float *prepared, float u,
                         size_t M, size_t N, float *y) {
  for (size_t i = 0; i < M; i++)
    for (size_t j = 0; j < N; j++) {
      float v = (float)accS[i * N + j] * u;
      y[i * N + j] = v + prepared[j];
    }
}

/* qmm::affine (Intgemm.inl.cc:92-156) and qmm::dot (:158-226, bias == NULL =>
 * the zero bias tensor of :188-189). */
void so_affine(const float *x, size_t M, size_t K, const int8_t *W, size_t N,
               const float *bias, float a_quant, float b_quant, float *y) {
  int32_t *accS = (int32_t *)malloc(M * N * sizeof(int32_t));
  float *prepared = (float *)malloc(N * sizeof(float));
  so_affine_acc(x, M, K, W, N, a_quant, accS);
  prepare_bias(W, K, N, bias, a_quant, b_quant, prepared);
  float u = 1.0f / (a_quant * b_quant);
  dequant_rows(accS, prepared, u, M, N, y);
  free(prepared);
  free(accS);
}

/* qmm::affine_with_select (Intgemm.inl.cc:7-90): PrepareBias over the FULL N,
 * SelectColumnsB + bias gather, then Multiply on the selected columns. */
void so_affine_select(const float *x, size_t M, size_t K, const int8_t *W,
                      size_t N, const float *bias, float a_quant,
                      float b_quant, const uint32_t *idx, size_t n_idx,
                      float *y) {
  float *prepared = (float *)malloc(N * sizeof(float));
  prepare_bias(W, K, N, bias, a_quant, b_quant, prepared);
  int8_t *Wsel = (int8_t *)malloc(n_idx * K);
  float *psel = (float *)malloc(n_idx * sizeof(float));
  for (size_t c = 0; c < n_idx; c++) {
    memcpy(Wsel + c * K, W + (size_t)idx[c] * K, K);
    psel[c] = prepared[idx[c]];
  }
  int32_t *accS = (int32_t *)malloc(M * n_idx * sizeof(int32_t));
  so_affine_acc(x, M, K, Wsel, n_idx, a_quant, accS);
  float u = 1.0f / (a_quant * b_quant);
  dequant_rows(accS, psel, u, M, n_idx, y);
  free(accS);
  free(psel);
  free(Wsel);
  free(prepared);
}

/* Ruy provider's float order (Ruy.inl.cc:45-54,104-114): signed accumulators,
 * y = float(acc)*u + bias[j]; quantise with roundf (:12). Used only to show
 * the two providers agree to float rounding. */
void so_affine_ruy(const float *x, size_t M, size_t K, const int8_t *W,
                   size_t N, const float *bias, float a_quant, float b_quant,
                   float *y) {
  int8_t *q = (int8_t *)malloc(M * K);
  for (size_t i = 0; i < M * K; i++) {
    float v = roundf(a_quant * x[i]);
    v = v < -127.0f ? -127.0f : v;
    v = v > 127.0f ? 127.0f : v;
    q[i] = (int8_t)v;
  }
  int32_t *acc = (int32_t *)malloc(M * N * sizeof(int32_t));
  so_gemm_i8_signed(q, W, M, K, N, acc);
  float u = 1.0f / (a_quant * b_quant);
  for (size_t i = 0; i < M; i++)
    for (size_t j = 0; j < N; j++) {
      float v = (float)acc[i * N + j] * u;
      y[i * N + j] = v + (bias ? bias[j] : 0.0f);
    }
  free(acc);
  free(q);
}

/* qmm::prepare_weight_transposed (Intgemm.inl.cc:228-235 ->
 * Int8::PrepareBTransposed; Ruy.inl.cc:259-265): weights is B^T, f32
 * [rows=N][cols=K]; quantise elementwise into the canonical [N][K]. */
void so_prepare_weight_transposed(const float *weights, int8_t *prepared,
                                  float quant_mult, size_t cols, size_t rows) {
  so_quantize(weights, quant_mult, rows * cols, prepared);
}

/* qmm::prepare_weight_quantized_transposed (Intgemm.inl.cc:237-243;
 * Ruy.inl.cc:267-274 = memcpy): input is already [N=cols][K=rows]. */
void so_prepare_weight_quantized_transposed(const int8_t *input,
                                            int8_t *output, size_t rows,
                                            size_t cols) {
  memcpy(output, input, rows * cols);
}

/* Io.cc:275-283 */
void so_unquantize_embedding(const int8_t *q, float quant_mult, size_t n,
                             float *out) {
  for (size_t i = 0; i < n; i++) out[i] = (float)q[i] * (1 / quant_mult);
}

/* ------------------------------------------------------------------------ */
/* TensorOps                                                                */
/* ------------------------------------------------------------------------ */

/* TensorOps.cc:542-580 (eps 1e-6 default, TensorOps.hh:67-68). */
void so_layer_norm(const float *in, const float *scale, const float *bias,
                   float eps, size_t rows, size_t cols, float *out) {
  float *tmp = (float *)malloc(cols * sizeof(float));
  for (size_t j = 0; j < rows; j++) {
    const float *x = in + j * cols;
    float *y = out + j * cols;
    float sum = so_row_sum(x, cols);
    float mean = sum / cols;
    for (size_t i = 0; i < cols; i++) {
      float v = x[i] - mean;
      tmp[i] = v * v;
    }
    float sq = so_row_sum(tmp, cols);
    float sigma = sqrtf(sq / cols + eps);
    for (size_t i = 0; i < cols; i++) {
      float t = (x[i] - mean) / sigma;
      float s = scale[i] * t;
      y[i] = s + bias[i];
    }
  }
  free(tmp);
}

/* TensorOps.cc:296-314 (scalar path; Simd.hh is dead code, SURVEY #9). */
void so_softmax(const float *logits, size_t rows, size_t cols, float *out) {
  float *tmp = (float *)malloc(cols * sizeof(float));
  for (size_t i = 0; i < rows; i++) {
    const float *xs = logits + i * cols;
    float max_value = -3.402823466e+38f;
    for (size_t j = 0; j < cols; j++) max_value = xs[j] > max_value ? xs[j] : max_value;
    for (size_t j = 0; j < cols; j++) tmp[j] = so_exp(xs[j] - max_value);
    float sumexp = so_row_sum(tmp, cols);
    for (size_t j = 0; j < cols; j++) out[i * cols + j] = tmp[j] / sumexp;
  }
  free(tmp);
}

/* TensorOps.cc:662-682 */
void so_highway(const float *x, const float *y, const float *g, size_t n,
                float *out) {
  for (size_t i = 0; i < n; i++) {
    float sg = so_sigmoid(g[i]);
    float a = sg * x[i];
    float b = (1.0f - sg) * y[i];
    out[i] = a + b;
  }
}

/* TensorOps.cc:163-181 */
void so_relu(const float *a, size_t n, float *out) {
  for (size_t i = 0; i < n; i++) out[i] = a[i] > 0.0f ? a[i] : 0.0f;
}

/* TensorOps.cc:123-141 */
void so_add(const float *a, const float *b, size_t n, float *out) {
  for (size_t i = 0; i < n; i++) out[i] = a[i] + b[i];
}

/* TensorOps.cc:245-265 */
void so_sinusoidal_signal(int start, size_t seq, size_t dim, float *out) {
  float num_timescales = (float)dim / 2;
  /* std::log(10000.0F) as its correctly rounded f32 value (what a constant-
   * folding compiler emits); avoids libm/compiler disagreement in the last bit */
  const float log_10000 = 9.210340371976184f;
  float log_timescale_increment = log_10000 / (num_timescales - 1.0f);
  for (size_t p = (size_t)start; p < seq + (size_t)start; ++p) {
    for (int i = 0; i < num_timescales; ++i) {
      float v = p * expf(i * -log_timescale_increment);
      size_t offset = (p - (size_t)start) * dim + (size_t)i;
      out[offset] = sinf(v);
      out[offset + (size_t)(int)num_timescales] = cosf(v);
    }
  }
}

/* Transformer.cc:24-49: x *= sqrt(D); x += sinusoid(start..start+S-1). */
void so_transform_embedding(float *emb, size_t batch, size_t seq, size_t dim,
                            size_t start) {
  float s = sqrtf((float)dim);
  for (size_t i = 0; i < batch * seq * dim; i++) emb[i] = emb[i] * s;
  float *pos = (float *)malloc(seq * dim * sizeof(float));
  so_sinusoidal_signal((int)start, seq, dim, pos);
  for (size_t b = 0; b < batch; b++)
    for (size_t i = 0; i < seq * dim; i++) emb[b * seq * dim + i] += pos[i];
  free(pos);
}

/* TensorOps.cc:227-243 */
void so_index_select(const float *table, const uint32_t *ids, size_t n,
                     size_t dim, float *out) {
  for (size_t i = 0; i < n; i++)
    memcpy(out + i * dim, table + (size_t)ids[i] * dim, dim * sizeof(float));
}

/* TensorOps.cc:98-121 */
void so_transpose_3120(const float *in, size_t d3, size_t d2, size_t d1,
                       size_t d0, float *out) {
  size_t cols = d0;
  size_t rows = d3 * d2 * d1;
  size_t rest = rows / (d2 * d1);
  for (size_t k = 0; k < rest; ++k) {
    size_t shift = k * d1 * d2;
    for (size_t j = 0; j < d1 * d2; ++j) {
      size_t src = j + shift;
      size_t dst = j / d1 + (j % d1) * d2 + shift;
      memcpy(out + dst * cols, in + src * cols, cols * sizeof(float));
    }
  }
}

/* batch_matrix_multiply (TensorOps.cc:479-529) over cblas_sgemm / ruy::Mul
 * (:336-473). The vendor GEMM's summation order is unspecified; the
 * restatement fixes it: C = alpha * (k-ascending fmaf chain from 0), the
 * chain being what an FMA sgemm micro-kernel (and gfx950's f32 MFMA) does;
 * alpha applied once after the sum (Ruy path :437-446; BLAS semantics). */
void so_bmm(const float *A, const float *B, size_t batch, size_t rows_a,
            size_t cols_a, size_t rows_b, size_t cols_b, int trans_b,
            float alpha, float *C) {
  size_t m = rows_a, k = cols_a;
  size_t n = trans_b ? rows_b : cols_b;
  for (size_t bi = 0; bi < batch; bi++) {
    const float *a = A + bi * rows_a * cols_a;
    const float *b = B + bi * rows_b * cols_b;
    float *c = C + bi * m * n;
    for (size_t i = 0; i < m; i++)
      for (size_t j = 0; j < n; j++) {
        float s = 0.0f;
        for (size_t t = 0; t < k; t++) {
          float bv = trans_b ? b[j * cols_b + t] : b[t * cols_b + j];
          s = fmaf(a[i * k + t], bv, s);
        }
        c[i * n + j] = (alpha != 1.0f) ? alpha * s : s;
      }
  }
}

/* scaled_dot_product_attention (Modules.cc:24-86). q [B,H,Tq,dh], k,v
 * [B,H,S,dh], mask [B,S] additive; out [B,H,Tq,dh], attn [B,H,Tq,S]. */
void so_sdpa(const float *q, const float *k, const float *v,
             const float *mask, size_t B, size_t H, size_t Tq, size_t S,
             size_t dh, float *out, float *attn) {
  float d_k = 1.0f / sqrtf((float)dh);
  float *qkt = (float *)malloc(B * H * Tq * S * sizeof(float));
  so_bmm(q, k, B * H, Tq, dh, S, dh, 1, d_k, qkt);
  for (size_t b = 0; b < B; b++)
    for (size_t r = 0; r < H * Tq; r++)
      for (size_t s = 0; s < S; s++) {
        float *p = qkt + (b * H * Tq + r) * S + s;
        *p = *p + mask[b * S + s];
      }
  so_softmax(qkt, B * H * Tq, S, attn);
  so_bmm(attn, v, B * H, Tq, S, S, dh, 0, 1.0f, out);
  free(qkt);
}

/* greedy_sample / greedy_sample_from_words (Transformer.cc:279-339):
 * first maximum wins (strict >), scan from class 0. */
void so_greedy_sample(const float *logits, size_t batch, size_t stride,
                      const uint32_t *words, uint32_t *out) {
  for (size_t i = 0; i < batch; i++) {
    size_t max_index = 0;
    float max_value = logits[i * stride];
    for (size_t c = 1; c < stride; c++) {
      float value = logits[i * stride + c];
      if (value > max_value) {
        max_index = c;
        max_value = value;
      }
    }
    out[i] = words ? words[max_index] : (uint32_t)max_index;
  }
}

/* Input.cc:20-63 */
void so_make_mask(const uint32_t *lengths, size_t B, size_t S, float *mask) {
  float lowest_half = -3.402823466e+38f / 2.0f;
  float minus_inf = lowest_half > -99999999.0f ? lowest_half : -99999999.0f;
  for (size_t b = 0; b < B; b++)
    for (size_t s = 0; s < S; s++) {
      float one = s < lengths[b] ? 1.0f : 0.0f;
      mask[b * S + s] = (1.0f - one) * minus_inf;
    }
}

/* ------------------------------------------------------------------------ */
/* model                                                                    */
/* ------------------------------------------------------------------------ */

typedef struct {
  const int8_t *W; /* [N][K] */
  const float *b;  /* [N] or NULL */
  float a_quant, b_quant;
  int K, N;
  /* cache (reference_cost == 0) */
  float *prepared;
} so_affine_p;

typedef struct {
  const float *scale, *bias;
} so_ln_p;

typedef struct {
  so_affine_p q, k, v, o;
  so_ln_p ln;
} so_attn_p;

typedef struct {
  so_attn_p attn;
  so_affine_p ffn1, ffn2;
  so_ln_p ffn_ln;
} so_enc_layer;

typedef struct {
  so_affine_p rnn_f, rnn_w; /* Wf (affine) / W (dot) */
  so_ln_p rnn_ln;
  so_attn_p attn;
  so_affine_p ffn1, ffn2;
  so_ln_p ffn_ln;
} so_dec_layer;

struct so_model {
  int D, F, H, V, Le, Ld;
  float *embedding;   /* f32 [V][D], Io.cc:191-200 */
  int8_t *out_W;      /* Wemb_intgemm8 [V][D] + mult, Io.cc:206-224 */
  so_affine_p output; /* Transformer.cc:111-113 */
  so_enc_layer *enc;
  so_dec_layer *dec;
  int reference_cost;
  int threads;
  /* per-batch caches for reference_cost == 0 (not thread safe; test infra) */
  float *kv_cache;
  int32_t *kv_acc; /* PORTABLE: the K / V projections' shifted accumulators, [Ld][K,V][B*S][D] */
  const float *kv_acc_src;
  size_t kv_acc_B, kv_acc_S;
  float kv_acc_probe[4];
  size_t kv_B, kv_S;
  const float *kv_src;
  float kv_probe[4]; /* first/last values of the cached encoder_out: a recycled
                        pointer with different contents must not hit the cache */
};

static const so_param *find_param(const so_param *p, size_t n, const char *name) {
  for (size_t i = 0; i < n; i++)
    if (strcmp(p[i].name, name) == 0) return &p[i];
  fprintf(stderr, "[so] missing parameter %s\n", name);
  return NULL;
}

static int bind_affine(so_affine_p *a, const so_param *p, size_t n,
                       const char *prefix, const char *wname, const char *bname) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s%s", prefix, wname);
  const so_param *W = find_param(p, n, buf);
  snprintf(buf, sizeof buf, "%s%s_QuantMultA", prefix, wname);
  const so_param *qa = find_param(p, n, buf);
  if (!W || !qa) return -1;
  a->W = (const int8_t *)W->data;
  a->K = W->rows;
  a->N = W->cols;
  /* Modules.cc:18-22: b_quant is the float right after the int8 payload */
  memcpy(&a->b_quant, a->W + (size_t)W->rows * W->cols, sizeof(float));
  a->a_quant = *(const float *)qa->data; /* Modules.cc:150 */
  a->b = NULL;
  a->prepared = NULL;
  if (bname) {
    snprintf(buf, sizeof buf, "%s%s", prefix, bname);
    const so_param *b = find_param(p, n, buf);
    if (!b) return -1;
    a->b = (const float *)b->data;
  }
  return 0;
}

static int bind_ln(so_ln_p *ln, const so_param *p, size_t n, const char *prefix) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s_ln_scale", prefix);
  const so_param *s = find_param(p, n, buf);
  snprintf(buf, sizeof buf, "%s_ln_bias", prefix);
  const so_param *b = find_param(p, n, buf);
  if (!s || !b) return -1;
  ln->scale = (const float *)s->data;
  ln->bias = (const float *)b->data;
  return 0;
}

/* Attention::register_parameters (Modules.cc:361-378) */
static int bind_attn(so_attn_p *a, const so_param *p, size_t n, const char *layer,
                     const char *kind) {
  char prefix[256];
  snprintf(prefix, sizeof prefix, "%s_%s_", layer, kind);
  int rc = 0;
  rc |= bind_affine(&a->q, p, n, prefix, "Wq", "bq");
  rc |= bind_affine(&a->k, p, n, prefix, "Wk", "bk");
  rc |= bind_affine(&a->v, p, n, prefix, "Wv", "bv");
  rc |= bind_affine(&a->o, p, n, prefix, "Wo", "bo");
  char lnp[300];
  snprintf(lnp, sizeof lnp, "%sWo", prefix);
  rc |= bind_ln(&a->ln, p, n, lnp);
  return rc;
}

so_model *so_model_create(const so_param *params, size_t n, int enc_layers,
                          int dec_layers, int heads) {
  so_model *m = (so_model *)calloc(1, sizeof(so_model));
  m->Le = enc_layers;
  m->Ld = dec_layers;
  m->H = heads;
  m->threads = 1;
  const so_param *wemb = find_param(params, n, "Wemb");
  if (!wemb) return NULL;
  m->V = wemb->rows;
  m->D = wemb->cols;
  size_t VD = (size_t)m->V * m->D;
  /* Io.cc:182-224: dequantise Wemb, then re-quantise it transposed for the
   * output layer with the same multiplier. */
  float mult;
  memcpy(&mult, (const int8_t *)wemb->data + VD, sizeof(float));
  m->embedding = (float *)malloc(VD * sizeof(float));
  so_unquantize_embedding((const int8_t *)wemb->data, mult, VD, m->embedding);
  m->out_W = (int8_t *)malloc(VD + sizeof(float));
  so_prepare_weight_transposed(m->embedding, m->out_W, mult, (size_t)m->D, (size_t)m->V);
  memcpy(m->out_W + VD, &mult, sizeof(float));
  const so_param *none_q = find_param(params, n, "none_QuantMultA");
  const so_param *out_b = find_param(params, n, "decoder_ff_logit_out_b");
  if (!none_q || !out_b) return NULL;
  m->output.W = m->out_W;
  m->output.K = m->D;
  m->output.N = m->V;
  m->output.b = (const float *)out_b->data;
  m->output.a_quant = *(const float *)none_q->data;
  m->output.b_quant = mult;

  m->enc = (so_enc_layer *)calloc((size_t)enc_layers, sizeof(so_enc_layer));
  m->dec = (so_dec_layer *)calloc((size_t)dec_layers, sizeof(so_dec_layer));
  int rc = 0;
  char layer[64], buf[128];
  for (int i = 0; i < enc_layers; i++) {
    snprintf(layer, sizeof layer, "encoder_l%d", i + 1);
    rc |= bind_attn(&m->enc[i].attn, params, n, layer, "self");
    snprintf(buf, sizeof buf, "%s_ffn_", layer);
    rc |= bind_affine(&m->enc[i].ffn1, params, n, buf, "W1", "b1");
    rc |= bind_affine(&m->enc[i].ffn2, params, n, buf, "W2", "b2");
    snprintf(buf, sizeof buf, "%s_ffn_ffn", layer);
    rc |= bind_ln(&m->enc[i].ffn_ln, params, n, buf);
  }
  for (int i = 0; i < dec_layers; i++) {
    snprintf(layer, sizeof layer, "decoder_l%d", i + 1);
    rc |= bind_attn(&m->dec[i].attn, params, n, layer, "context");
    snprintf(buf, sizeof buf, "%s_ffn_", layer);
    rc |= bind_affine(&m->dec[i].ffn1, params, n, buf, "W1", "b1");
    rc |= bind_affine(&m->dec[i].ffn2, params, n, buf, "W2", "b2");
    snprintf(buf, sizeof buf, "%s_ffn_ffn", layer);
    rc |= bind_ln(&m->dec[i].ffn_ln, params, n, buf);
    /* SSRU::register_parameters (Modules.cc:389-400) */
    snprintf(buf, sizeof buf, "%s_rnn_", layer);
    rc |= bind_affine(&m->dec[i].rnn_w, params, n, buf, "W", NULL);
    /* Wf's bias is named "bf" (not "bWf") */
    {
      so_affine_p *a = &m->dec[i].rnn_f;
      rc |= bind_affine(a, params, n, buf, "Wf", NULL);
      char bn[200];
      snprintf(bn, sizeof bn, "%sbf", buf);
      const so_param *b = find_param(params, n, bn);
      if (!b) rc = -1; else a->b = (const float *)b->data;
    }
    snprintf(buf, sizeof buf, "%s_rnn_ffn", layer);
    rc |= bind_ln(&m->dec[i].rnn_ln, params, n, buf);
  }
  if (rc) {
    so_model_destroy(m);
    return NULL;
  }
  m->F = m->enc[0].ffn1.N;
  return m;
}

static void free_affine_cache(so_affine_p *a) {
  free(a->prepared);
  a->prepared = NULL;
}

void so_model_destroy(so_model *m) {
  if (!m) return;
  for (int i = 0; m->enc && i < m->Le; i++) {
    free_affine_cache(&m->enc[i].attn.q); free_affine_cache(&m->enc[i].attn.k);
    free_affine_cache(&m->enc[i].attn.v); free_affine_cache(&m->enc[i].attn.o);
    free_affine_cache(&m->enc[i].ffn1); free_affine_cache(&m->enc[i].ffn2);
  }
  for (int i = 0; m->dec && i < m->Ld; i++) {
    free_affine_cache(&m->dec[i].attn.q); free_affine_cache(&m->dec[i].attn.k);
    free_affine_cache(&m->dec[i].attn.v); free_affine_cache(&m->dec[i].attn.o);
    free_affine_cache(&m->dec[i].ffn1); free_affine_cache(&m->dec[i].ffn2);
    free_affine_cache(&m->dec[i].rnn_f); free_affine_cache(&m->dec[i].rnn_w);
  }
  free_affine_cache(&m->output);
  free(m->kv_cache);
  free(m->kv_acc);
  free(m->enc);
  free(m->dec);
  free(m->embedding);
  free(m->out_W);
  free(m);
}

int so_model_dim(const so_model *m) { return m->D; }
int so_model_ffn(const so_model *m) { return m->F; }
int so_model_vocab(const so_model *m) { return m->V; }
void so_model_set_reference_cost(so_model *m, int on) { m->reference_cost = on; }
void so_model_set_threads(so_model *m, int n) {
  m->threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(m->threads);
#endif
}

/* OpenMP's thread count is a per-thread setting: a model used from another host thread (the
 * bench's one-worker-per-core CPU baseline, pytest workers) would otherwise run its GEMMs on
 * a team as large as the machine, whatever set_threads said. Every entry point calls this. */
static void use_model_threads(const so_model *m) {
#ifdef _OPENMP
  omp_set_num_threads(m->threads);
#else
  (void)m;
#endif
}

/* affine()/linear() wrappers of Modules.cc:145-180 */
static void apply_affine(const so_model *m, const so_affine_p *a_, const float *x,
                         size_t M, float *y) {
  so_affine_p *a = (so_affine_p *)a_;
  if (m->reference_cost) {
    so_affine(x, M, (size_t)a->K, a->W, (size_t)a->N, a->b, a->a_quant, a->b_quant, y);
    return;
  }
  if (!a->prepared) {
    a->prepared = (float *)malloc((size_t)a->N * sizeof(float));
    prepare_bias(a->W, (size_t)a->K, (size_t)a->N, a->b, a->a_quant, a->b_quant, a->prepared);
  }
  int32_t *accS = (int32_t *)malloc(M * (size_t)a->N * sizeof(int32_t));
  so_affine_acc(x, M, (size_t)a->K, a->W, (size_t)a->N, a->a_quant, accS);
  float u = 1.0f / (a->a_quant * a->b_quant);
  dequant_rows(accS, a->prepared, u, M, (size_t)a->N, y);
  free(accS);
}

static void apply_ln(const so_ln_p *ln, const float *x, size_t rows, size_t cols,
                     float *y) {
  so_layer_norm(x, ln->scale, ln->bias, 1e-6f, rows, cols, y);
}

/* Attention::forward (Modules.cc:287-319). q [B,Tq,D]; kv [B,S,D]; if
 * yk_pre/yv_pre are given they are the already split K/V projections
 * ([B,H,S,dh]) of kv (cached variant). out y [B,Tq,D], attn [B,H,Tq,S]. */
static void attention_forward(const so_model *m, const so_attn_p *a, const float *q,
                              const float *kv, const float *mask, size_t B,
                              size_t Tq, size_t S, const float *yk_pre,
                              const float *yv_pre, float *y, float *attn) {
  size_t D = (size_t)m->D, H = (size_t)m->H, dh = D / H;
  float *yq = (float *)malloc(B * Tq * D * sizeof(float));
  float *sq = (float *)malloc(B * Tq * D * sizeof(float));
  apply_affine(m, &a->q, q, B * Tq, yq);
  so_transpose_3120(yq, B, Tq, H, dh, sq); /* split_heads, Modules.cc:88-126 */
  float *sk = NULL, *sv = NULL;
  const float *pk = yk_pre, *pv = yv_pre;
  if (!pk) {
    float *yk = (float *)malloc(B * S * D * sizeof(float));
    float *yv = (float *)malloc(B * S * D * sizeof(float));
    sk = (float *)malloc(B * S * D * sizeof(float));
    sv = (float *)malloc(B * S * D * sizeof(float));
    apply_affine(m, &a->k, kv, B * S, yk);
    apply_affine(m, &a->v, kv, B * S, yv);
    so_transpose_3120(yk, B, S, H, dh, sk);
    so_transpose_3120(yv, B, S, H, dh, sv);
    free(yk);
    free(yv);
    pk = sk;
    pv = sv;
  }
  float *ao = (float *)malloc(B * Tq * D * sizeof(float));
  float *attn_local = attn ? attn : (float *)malloc(B * H * Tq * S * sizeof(float));
  so_sdpa(sq, pk, pv, mask, B, H, Tq, S, dh, ao, attn_local);
  float *joined = (float *)malloc(B * Tq * D * sizeof(float));
  so_transpose_3120(ao, B, H, Tq, dh, joined); /* join_heads, Modules.cc:128-143 */
  float *yo = (float *)malloc(B * Tq * D * sizeof(float));
  apply_affine(m, &a->o, joined, B * Tq, yo);
  float *xpy = (float *)malloc(B * Tq * D * sizeof(float));
  so_add(q, yo, B * Tq * D, xpy); /* Modules.cc:314 */
  apply_ln(&a->ln, xpy, B * Tq, D, y);
  free(xpy); free(yo); free(joined);
  if (!attn) free(attn_local);
  free(ao); free(sk); free(sv); free(sq); free(yq);
}

/* FFN block pattern of Modules.cc:251-257,326-331 */
static void ffn_block(const so_model *m, const so_affine_p *f1, const so_affine_p *f2,
                      const so_ln_p *ln, const float *x, size_t M, float *y) {
  size_t D = (size_t)m->D, F = (size_t)f1->N;
  float *h1 = (float *)malloc(M * F * sizeof(float));
  apply_affine(m, f1, x, M, h1);
  so_relu(h1, M * F, h1);
  float *h2 = (float *)malloc(M * D * sizeof(float));
  apply_affine(m, f2, h1, M, h2);
  so_add(h2, x, M * D, h2);
  apply_ln(ln, h2, M, D, y);
  free(h2);
  free(h1);
}

/* EncoderLayer::forward (Modules.cc:321-334) */
void so_encoder_layer(const so_model *m, int layer, const float *x,
                      const float *mask, size_t B, size_t S, float *out,
                      float *attn) {
  use_model_threads(m);
  const so_enc_layer *L = &m->enc[layer - 1];
  size_t D = (size_t)m->D;
  float *a = (float *)malloc(B * S * D * sizeof(float));
  attention_forward(m, &L->attn, x, x, mask, B, S, S, NULL, NULL, a, attn);
  ffn_block(m, &L->ffn1, &L->ffn2, &L->ffn_ln, a, B * S, out);
  free(a);
}

/* Model.cc:195-197 */
void so_embed(const so_model *m, const uint32_t *ids, size_t B, size_t S,
              float *out) {
  use_model_threads(m);
  so_index_select(m->embedding, ids, B * S, (size_t)m->D, out);
  so_transform_embedding(out, B, S, (size_t)m->D, 0);
}

/* Encoder::forward (Transformer.cc:57-69) */
void so_encode(const so_model *m, const float *x, const float *mask, size_t B,
               size_t S, float *out) {
  use_model_threads(m);
  size_t n = B * S * (size_t)m->D;
  float *cur = (float *)malloc(n * sizeof(float));
  float *nxt = (float *)malloc(n * sizeof(float));
  memcpy(cur, x, n * sizeof(float));
  for (int i = 1; i <= m->Le; i++) {
    so_encoder_layer(m, i, cur, mask, B, S, nxt, NULL);
    float *t = cur;
    cur = nxt;
    nxt = t;
  }
  memcpy(out, cur, n * sizeof(float));
  free(cur);
  free(nxt);
}

/* SSRU::forward (Modules.cc:190-235) */
static void ssru_forward(const so_model *m, const so_dec_layer *L, float *state,
                         const float *x, size_t B, float *h) {
  size_t D = (size_t)m->D, n = B * D;
  float *f = (float *)malloc(n * sizeof(float));
  float *wx = (float *)malloc(n * sizeof(float));
  apply_affine(m, &L->rnn_f, x, B, f);  /* :217 */
  apply_affine(m, &L->rnn_w, x, B, wx); /* :218 dot */
  float *c_t = (float *)malloc(n * sizeof(float));
  so_highway(state, wx, f, n, c_t); /* :223 highway(c, Wxt, f) */
  float *yv = (float *)malloc(n * sizeof(float));
  so_relu(c_t, n, yv);
  so_add(x, yv, n, yv); /* :230 */
  apply_ln(&L->rnn_ln, yv, B, D, h);
  memcpy(state, c_t, n * sizeof(float)); /* :232 */
  free(yv); free(c_t); free(wx); free(f);
}

/* cross-attention K/V of the encoder output, per decoder layer, cached once
 * per (encoder_out pointer, B, S) when reference_cost == 0. */
static const float *kv_for(const so_model *m_, int layer, int which,
                           const float *encoder_out, size_t B, size_t S) {
  so_model *m = (so_model *)m_;
  size_t D = (size_t)m->D, H = (size_t)m->H, dh = D / H;
  size_t per = B * S * D;
  const float probe[4] = {encoder_out[0], encoder_out[per / 2], encoder_out[per - 1],
                          encoder_out[per / 3]};
  if (m->kv_src != encoder_out || m->kv_B != B || m->kv_S != S || !m->kv_cache ||
      memcmp(probe, m->kv_probe, sizeof probe) != 0) {
    free(m->kv_cache);
    m->kv_cache = (float *)malloc(per * 2 * (size_t)m->Ld * sizeof(float));
    float *tmp = (float *)malloc(per * sizeof(float));
    for (int l = 0; l < m->Ld; l++) {
      apply_affine(m, &m->dec[l].attn.k, encoder_out, B * S, tmp);
      so_transpose_3120(tmp, B, S, H, dh, m->kv_cache + (size_t)(2 * l) * per);
      apply_affine(m, &m->dec[l].attn.v, encoder_out, B * S, tmp);
      so_transpose_3120(tmp, B, S, H, dh, m->kv_cache + (size_t)(2 * l + 1) * per);
    }
    free(tmp);
    m->kv_src = encoder_out;
    m->kv_B = B;
    m->kv_S = S;
    memcpy(m->kv_probe, probe, sizeof probe);
  }
  return m->kv_cache + (size_t)(2 * layer + which) * per;
}

/* PORTABLE order of the decoder's cross-attention (Attention::forward, Modules.cc:287-319, over the
 * K / V projections of the encoder output, which do not change between steps). The reference
 * dequantises every K / V element (y = float(accS) * u + pb, Intgemm.inl.cc:146-153) and then runs the
 * float attention (Modules.cc:24-86). The GPU keeps the projections' int32 accumulators between steps;
 * rebuilding u * acc + pb per cached element and step is two thirds of its attention arithmetic, so the
 * PORTABLE order applies the (per column constant) u and pb AFTER the sums -- the same real numbers,
 * other roundings; FAITHFUL stays the literal sequence:
 *   scores   t_j = fmaf chain over the head's columns d (ascending, from 0) of q_d * float(accK[j][d])
 *            c_h = row sum (PORTABLE row order) of the products q_d * pbK[d] over the head's columns
 *            s_j = alpha * fmaf(t_j, uK, c_h) + mask_j       (alpha, mask, softmax as Modules.cc:45-75)
 *   context  w_d = fmaf chain over the keys j (ascending, from 0) of p_j * float(accV[j][d])
 *            P_h = row sum (PORTABLE row order) of the head's probabilities
 *            o_d = fmaf(w_d, uV, pbV[d] * P_h)
 * float(acc) is exact (|acc| <= 254 * 127 * K < 2^24 for K <= 512). */
static const int32_t *kv_acc_for(const so_model *m_, int layer, int which,
                                 const float *encoder_out, size_t B, size_t S) {
  so_model *m = (so_model *)m_;
  size_t D = (size_t)m->D, per = B * S * D;
  const float probe[4] = {encoder_out[0], encoder_out[per / 2], encoder_out[per - 1],
                          encoder_out[per / 3]};
  if (m->kv_acc_src != encoder_out || m->kv_acc_B != B || m->kv_acc_S != S || !m->kv_acc ||
      memcmp(probe, m->kv_acc_probe, sizeof probe) != 0) {
    free(m->kv_acc);
    m->kv_acc = (int32_t *)malloc(per * 2 * (size_t)m->Ld * sizeof(int32_t));
    for (int l = 0; l < m->Ld; l++) {
      const so_affine_p *k = &m->dec[l].attn.k, *v = &m->dec[l].attn.v;
      so_affine_acc(encoder_out, B * S, D, k->W, D, k->a_quant, m->kv_acc + (size_t)(2 * l) * per);
      so_affine_acc(encoder_out, B * S, D, v->W, D, v->a_quant, m->kv_acc + (size_t)(2 * l + 1) * per);
    }
    m->kv_acc_src = encoder_out;
    m->kv_acc_B = B;
    m->kv_acc_S = S;
    memcpy(m->kv_acc_probe, probe, sizeof probe);
  }
  return m->kv_acc + (size_t)(2 * layer + which) * per;
}

/* out [B][D] (heads joined), attn [B][H][S] */
static void cross_attention_portable(const so_model *m, const so_attn_p *a, int layer, const float *yq,
                                     const float *encoder_out, const float *mask, size_t B, size_t S,
                                     float *out, float *attn) {
  size_t D = (size_t)m->D, H = (size_t)m->H, dh = D / H;
  const int32_t *accK = kv_acc_for(m, layer, 0, encoder_out, B, S);
  const int32_t *accV = kv_acc_for(m, layer, 1, encoder_out, B, S);
  float *pbK = (float *)malloc(D * sizeof(float)), *pbV = (float *)malloc(D * sizeof(float));
  prepare_bias(a->k.W, D, D, a->k.b, a->k.a_quant, a->k.b_quant, pbK);
  prepare_bias(a->v.W, D, D, a->v.b, a->v.a_quant, a->v.b_quant, pbV);
  const float uK = 1.0f / (a->k.a_quant * a->k.b_quant), uV = 1.0f / (a->v.a_quant * a->v.b_quant);
  const float alpha = 1.0f / sqrtf((float)dh);
  float *sc = (float *)malloc(S * sizeof(float)), *prod = (float *)malloc(dh * sizeof(float));
  for (size_t b = 0; b < B; b++)
    for (size_t h = 0; h < H; h++) {
      const float *q = yq + b * D + h * dh;
      for (size_t d = 0; d < dh; d++) prod[d] = q[d] * pbK[h * dh + d];
      const float c = so_row_sum(prod, dh);
      for (size_t j = 0; j < S; j++) {
        const int32_t *k = accK + (b * S + j) * D + h * dh;
        float t = 0.0f;
        for (size_t d = 0; d < dh; d++) t = fmaf(q[d], (float)k[d], t);
        float s = fmaf(t, uK, c);
        if (alpha != 1.0f) s = alpha * s;
        sc[j] = s + mask[b * S + j];
      }
      float *p = attn + (b * H + h) * S;
      so_softmax(sc, 1, S, p);
      const float P = so_row_sum(p, S);
      for (size_t d = 0; d < dh; d++) {
        float w = 0.0f;
        for (size_t j = 0; j < S; j++) w = fmaf(p[j], (float)accV[(b * S + j) * D + h * dh + d], w);
        const float pbP = pbV[h * dh + d] * P;
        out[b * D + h * dh + d] = fmaf(w, uV, pbP);
      }
    }
  free(prod); free(sc); free(pbV); free(pbK);
}

/* Attention::forward (Modules.cc:287-319) for the decoder's cross-attention in the PORTABLE order:
 * q projection, the attention above, then the output projection, residual and LayerNorm as
 * attention_forward. q [B,D] (Tq = 1), y [B,D], attn [B,H,S]. */
static void cross_attention_forward_portable(const so_model *m, const so_attn_p *a, int layer,
                                             const float *q, const float *encoder_out, const float *mask,
                                             size_t B, size_t S, float *y, float *attn) {
  size_t D = (size_t)m->D;
  float *yq = (float *)malloc(B * D * sizeof(float));
  float *joined = (float *)malloc(B * D * sizeof(float));
  float *yo = (float *)malloc(B * D * sizeof(float));
  float *xpy = (float *)malloc(B * D * sizeof(float));
  apply_affine(m, &a->q, q, B, yq);
  cross_attention_portable(m, a, layer, yq, encoder_out, mask, B, S, joined, attn);
  apply_affine(m, &a->o, joined, B, yo);
  so_add(q, yo, B * D, xpy); /* Modules.cc:314 */
  apply_ln(&a->ln, xpy, B, D, y);
  free(xpy); free(yo); free(joined); free(yq);
}

/* The attention proper of a decoder layer's cross-attention, as an op of its own (tests: the PORTABLE
 * order against the FAITHFUL one on the same projected query): yq [B,D] = the Q projection's output,
 * out [B,D] = the joined heads BEFORE the output projection, attn [B,H,S]. FAITHFUL: K / V dequantised
 * element by element, then scaled_dot_product_attention (Modules.cc:24-86, 287-306); PORTABLE: the
 * hoisted order above. */
void so_cross_attention(const so_model *m, int layer, const float *yq, const float *encoder_out,
                        const float *mask, size_t B, size_t S, float *out, float *attn) {
  use_model_threads(m);
  const so_attn_p *a = &m->dec[layer].attn;
  if (g_mode == SO_PORTABLE) {
    ((so_model *)m)->kv_acc_src = NULL;
    cross_attention_portable(m, a, layer, yq, encoder_out, mask, B, S, out, attn);
    return;
  }
  size_t D = (size_t)m->D, H = (size_t)m->H, dh = D / H;
  float *yk = (float *)malloc(B * S * D * sizeof(float)), *yv = (float *)malloc(B * S * D * sizeof(float));
  float *sk = (float *)malloc(B * S * D * sizeof(float)), *sv = (float *)malloc(B * S * D * sizeof(float));
  float *sq = (float *)malloc(B * D * sizeof(float)), *ao = (float *)malloc(B * D * sizeof(float));
  apply_affine(m, &a->k, encoder_out, B * S, yk);
  apply_affine(m, &a->v, encoder_out, B * S, yv);
  so_transpose_3120(yq, B, 1, H, dh, sq);
  so_transpose_3120(yk, B, S, H, dh, sk);
  so_transpose_3120(yv, B, S, H, dh, sv);
  so_sdpa(sq, sk, sv, mask, B, H, 1, S, dh, ao, attn);
  so_transpose_3120(ao, B, H, 1, dh, out);
  free(ao); free(sq); free(sv); free(sk); free(yv); free(yk);
}

/* Decoder::step (Transformer.cc:120-183) with DecoderLayer::forward
 * (Modules.cc:237-259). */
void so_decode_step(const so_model *m, const float *encoder_out,
                    const float *mask, size_t B, size_t S, float *states,
                    const uint32_t *prev, const uint32_t *shortlist,
                    size_t n_sl, float *logits, float *attn) {
  use_model_threads(m);
  size_t D = (size_t)m->D, H = (size_t)m->H, n = B * D;
  float *x = (float *)malloc(n * sizeof(float));
  if (!prev) {
    memset(x, 0, n * sizeof(float)); /* :138-144 */
  } else {
    so_index_select(m->embedding, prev, B, D, x); /* :146-156 */
  }
  so_transform_embedding(x, B, 1, D, 0); /* :160, position always 0 */
  float *h = (float *)malloc(n * sizeof(float));
  float *a = (float *)malloc(n * sizeof(float));
  float *attn_l = (float *)malloc(B * H * S * sizeof(float));
  for (int l = 0; l < m->Ld; l++) {
    const so_dec_layer *L = &m->dec[l];
    ssru_forward(m, L, states + (size_t)l * n, x, B, h);
    if (g_mode == SO_PORTABLE) {
      cross_attention_forward_portable(m, &L->attn, l, h, encoder_out, mask, B, S, a, attn_l);
    } else {
      const float *pk = NULL, *pv = NULL;
      if (!m->reference_cost) {
        pk = kv_for(m, l, 0, encoder_out, B, S);
        pv = kv_for(m, l, 1, encoder_out, B, S);
      }
      attention_forward(m, &L->attn, h, encoder_out, mask, B, 1, S, pk, pv, a, attn_l);
    }
    ffn_block(m, &L->ffn1, &L->ffn2, &L->ffn_ln, a, B, x);
    /* :165-174: alignment = attention of the LAST layer */
    if (attn && l + 1 == m->Ld) memcpy(attn, attn_l, B * H * S * sizeof(float));
  }
  const so_affine_p *o = &m->output;
  if (shortlist) { /* :176-179 */
    if (m->reference_cost) {
      so_affine_select(x, B, D, o->W, (size_t)o->N, o->b, o->a_quant, o->b_quant,
                       shortlist, n_sl, logits);
    } else {
      so_affine_p *oc = (so_affine_p *)o;
      if (!oc->prepared) {
        oc->prepared = (float *)malloc((size_t)o->N * sizeof(float));
        prepare_bias(o->W, D, (size_t)o->N, o->b, o->a_quant, o->b_quant, oc->prepared);
      }
      int8_t *Wsel = (int8_t *)malloc(n_sl * D);
      float *psel = (float *)malloc(n_sl * sizeof(float));
      for (size_t c = 0; c < n_sl; c++) {
        memcpy(Wsel + c * D, o->W + (size_t)shortlist[c] * D, D);
        psel[c] = oc->prepared[shortlist[c]];
      }
      int32_t *accS = (int32_t *)malloc(B * n_sl * sizeof(int32_t));
      so_affine_acc(x, B, D, Wsel, n_sl, o->a_quant, accS);
      dequant_rows(accS, psel, 1.0f / (o->a_quant * o->b_quant), B, n_sl, logits);
      free(accS); free(psel); free(Wsel);
    }
  } else {
    apply_affine(m, o, x, B, logits); /* :181 */
  }
  free(attn_l); free(a); free(h); free(x);
}

/* Model::forward + Model::decode (Model.cc:111-204). */
size_t so_translate(const so_model *m, const uint32_t *src_ids,
                    const uint32_t *lengths, size_t B, size_t S,
                    const uint32_t *shortlist, size_t n_sl,
                    float limit_factor, uint32_t eos_id, uint32_t *out_ids,
                    uint32_t *out_len, float *align) {
  use_model_threads(m);
  size_t D = (size_t)m->D, H = (size_t)m->H;
  size_t N = shortlist ? n_sl : (size_t)m->V;
  ((so_model *)m)->kv_src = NULL; /* new batch: the cross-attention K/V cache is stale */
  ((so_model *)m)->kv_acc_src = NULL;
  float *mask = (float *)malloc(B * S * sizeof(float));
  so_make_mask(lengths, B, S, mask);
  float *emb = (float *)malloc(B * S * D * sizeof(float));
  float *enc = (float *)malloc(B * S * D * sizeof(float));
  so_embed(m, src_ids, B, S, emb);
  so_encode(m, emb, mask, B, S, enc);
  free(emb);

  size_t max_seq_length = (size_t)(limit_factor * (float)S); /* :160 */
  /* the first step (:144-157) is unconditional: one token is always recorded, so the
   * output arrays have max(max_seq_length, 1) columns */
  if (max_seq_length == 0) max_seq_length = 1;
  float *states = (float *)calloc((size_t)m->Ld * B * D, sizeof(float)); /* :145 */
  float *logits = (float *)malloc(B * N * sizeof(float));
  float *attn = (float *)malloc(B * H * S * sizeof(float));
  uint32_t *prev = (uint32_t *)malloc(B * sizeof(uint32_t));
  uint8_t *complete = (uint8_t *)calloc(B, 1);
  memset(out_len, 0, B * sizeof(uint32_t));
  if (align) memset(align, 0, B * max_seq_length * S * sizeof(float));
  size_t remaining = B, steps = 0;
  for (size_t i = 0; (i == 0) || (i < max_seq_length && remaining > 0); i++) {
    so_decode_step(m, enc, mask, B, S, states, i == 0 ? NULL : prev, shortlist,
                   n_sl, logits, attn);
    so_greedy_sample(logits, B, N, shortlist, prev);
    steps++;
    size_t finished = 0;
    for (size_t b = 0; b < B; b++) {
      if (!complete[b]) {
        /* update_alignment (:84-108) runs before record with the old flags */
        if (align && out_len[b] < max_seq_length)
          memcpy(align + (b * max_seq_length + out_len[b]) * S, attn + b * H * S,
                 lengths[b] * sizeof(float));
        complete[b] = (prev[b] == eos_id); /* record, :127-137 */
        if (out_len[b] < max_seq_length) out_ids[b * max_seq_length + out_len[b]] = prev[b];
        out_len[b]++;
      }
      finished += complete[b];
    }
    remaining = B - finished;
  }
  free(complete); free(prev); free(attn); free(logits); free(states);
  free(enc); free(mask);
  return steps;
}

/* ---- lexical shortlist (slimt/Shortlist.{hh,cc}) -------------------------- */

/* hash_combine over uint64 words starting at header.frequent (Utils.hh:47-67;
 * std::hash<uint64_t> is the identity in libstdc++) */
uint64_t so_shortlist_checksum(const void *blob, size_t size) {
  if (size < 16) return 0;
  const unsigned char *p = (const unsigned char *)blob + 16;
  size_t n = (size - 16) / 8; /* Shortlist.cc:67-73 */
  uint64_t seed = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t v;
    memcpy(&v, p + 8 * i, 8);
    seed ^= v + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
  }
  return seed;
}

int so_shortlist_parse(const void *blob, size_t size, int check, size_t target_vocab,
                       so_shortlist *out) {
  uint64_t h[6];
  if (size < sizeof(h)) return 1; /* Shortlist.cc:49-51 */
  memcpy(h, blob, sizeof(h));
  if (h[0] != SO_SHORTLIST_MAGIC) return 2; /* :56 */
  uint64_t expected = sizeof(h) + h[4] * 8 + h[5] * 4; /* :58-60 */
  if (expected != size) return 3;
  if (check && so_shortlist_checksum(blob, size) != h[1]) return 4; /* :66-76 */
  out->frequent = h[2];
  out->best = h[3];
  out->word_to_offset_size = h[4];
  out->shortlist_size = h[5];
  out->word_to_offset = (const uint64_t *)((const char *)blob + sizeof(h));
  out->shortlist = (const uint32_t *)((const char *)blob + sizeof(h) + h[4] * 8);
  if (check) { /* content_check(), Shortlist.cc:16-38 */
    if (h[4] == 0) return 6;
    for (uint64_t i = 0; i + 1 < h[4]; ++i)
      if (out->word_to_offset[i] >= h[5]) return 5;
    if (out->word_to_offset[h[4] - 1] != h[5]) return 6;
    for (uint64_t j = 0; j < h[5]; ++j)
      if (out->shortlist[j] >= target_vocab) return 7;
  }
  return 0;
}

size_t so_shortlist_generate(const so_shortlist *sl, int shared, size_t source_vocab,
                             size_t target_vocab, const uint32_t *words, size_t n_words,
                             uint32_t *out) {
  unsigned char *source_table = (unsigned char *)calloc(source_vocab ? source_vocab : 1, 1);
  unsigned char *target_table = (unsigned char *)calloc(target_vocab ? target_vocab : 1, 1);
  /* most frequent words, Shortlist.cc:125-127 */
  for (uint64_t i = 0; i < sl->frequent && i < target_vocab; ++i) target_table[i] = 1;
  /* unique source words -> their aligned target words, :131-145 */
  for (size_t k = 0; k < n_words; ++k) {
    uint32_t word = words[k];
    if (shared) target_table[word] = 1;
    if (!source_table[word]) {
      uint64_t begin = sl->word_to_offset[word], end = sl->word_to_offset[word + 1];
      for (uint64_t j = begin; j < end; ++j) target_table[sl->shortlist[j]] = 1;
      source_table[word] = 1;
    }
  }
  /* multiple-of-eight patch, :148-165 */
  size_t ones = 0;
  for (size_t i = 0; i < target_vocab; ++i) ones += target_table[i];
  for (size_t i = sl->frequent; i < target_vocab && ones % 8 != 0; ++i)
    if (!target_table[i]) {
      target_table[i] = 1;
      ones++;
    }
  /* bucket sort, :168-173 */
  size_t n = 0;
  for (size_t i = 0; i < target_vocab; ++i)
    if (target_table[i]) out[n++] = (uint32_t)i;
  free(source_table);
  free(target_table);
  return n;
}
