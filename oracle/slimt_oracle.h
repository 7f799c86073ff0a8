/*
 * slimt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the slimt int8 transformer-NMT hot path:
 * slimt::qmm (intgemm "Int8Shift" arithmetic) + TensorOps + Modules +
 * Transformer + the greedy loop of Model::decode. Every function cites the
 * reference file:line it follows (paths relative to the slimt checkout).
 *
 * PARITY UNPINNED: the reference ships no golden vectors / known-answer tests
 * for this path (tests/generate-units.py:149-155 maps every int8 op to NoOp,
 * CI asserts no output) and its hot path cannot be built in this image (it
 * needs intgemm | ruy | gemmology+xsimd, a BLAS with cblas.h and
 * sentencepiece, all absent; stand-ins are not allowed). The restatement is
 * therefore pinned only by (a) independent numpy/torch re-derivations in
 * tests/, (b) the signed-vs-shifted accumulator identity, (c) the committed
 * fixtures under tests/golden/ that this oracle generated itself.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library. The product (libslimt_hip.so) never links or calls it.
 *
 * Two float modes (so_set_mode):
 *   SO_FAITHFUL (0): the reference's scalar float path -- libm expf, strictly
 *       sequential f32 sums (TensorOps.cc:296-314,553-578).
 *   SO_PORTABLE (1): same formulas, but (i) exp is a fixed fmaf polynomial and
 *       (ii) row sums use a fixed 64-lane tree order, so the GPU can reproduce
 *       every float BIT-FOR-BIT. FAITHFUL vs PORTABLE differ by float rounding
 *       only (<= ~1e-6 relative; checked in tests/test_oracle.py).
 */
#ifndef SLIMT_ORACLE_H
#define SLIMT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SO_FAITHFUL = 0, SO_PORTABLE = 1 };
void so_set_mode(int mode);
int so_get_mode(void);

/* ---- scalar helpers (exposed for tests) -------------------------------- */
float so_exp(float x);
float so_sigmoid(float x);
float so_row_sum(const float *x, size_t n);

/* ---- qmm (QMM.hh:48-63; Intgemm.inl.cc; Ruy.inl.cc) -------------------- */
/* W is the canonical prepared layout: int8 [N][K], K contiguous (== the
 * Marian intgemm8 payload, Io.cc:225-239). */
void so_quantize(const float *x, float a_quant, size_t n, int8_t *q);
void so_gemm_i8_signed(const int8_t *q, const int8_t *W, size_t M, size_t K,
                       size_t N, int32_t *acc);
void so_gemm_i8_shifted(const int8_t *q, const int8_t *W, size_t M, size_t K,
                        size_t N, int32_t *accS);
void so_affine(const float *x, size_t M, size_t K, const int8_t *W, size_t N,
               const float *bias /* NULL => dot */, float a_quant,
               float b_quant, float *y);
void so_affine_acc(const float *x, size_t M, size_t K, const int8_t *W,
                   size_t N, float a_quant, int32_t *accS);
void so_affine_select(const float *x, size_t M, size_t K, const int8_t *W,
                      size_t N, const float *bias, float a_quant,
                      float b_quant, const uint32_t *idx, size_t n_idx,
                      float *y);
void so_affine_ruy(const float *x, size_t M, size_t K, const int8_t *W,
                   size_t N, const float *bias, float a_quant, float b_quant,
                   float *y);
void so_prepare_weight_transposed(const float *weights, int8_t *prepared,
                                  float quant_mult, size_t cols, size_t rows);
void so_prepare_weight_quantized_transposed(const int8_t *input,
                                            int8_t *output, size_t rows,
                                            size_t cols);
void so_unquantize_embedding(const int8_t *q, float quant_mult, size_t n,
                             float *out);

/* ---- TensorOps ---------------------------------------------------------- */
void so_layer_norm(const float *in, const float *scale, const float *bias,
                   float eps, size_t rows, size_t cols, float *out);
void so_softmax(const float *logits, size_t rows, size_t cols, float *out);
void so_highway(const float *x, const float *y, const float *g, size_t n,
                float *out);
void so_relu(const float *a, size_t n, float *out);
void so_add(const float *a, const float *b, size_t n, float *out);
void so_sinusoidal_signal(int start, size_t seq, size_t dim, float *out);
void so_transform_embedding(float *emb, size_t batch, size_t seq, size_t dim,
                            size_t start);
void so_index_select(const float *table, const uint32_t *ids, size_t n,
                     size_t dim, float *out);
void so_transpose_3120(const float *in, size_t d3, size_t d2, size_t d1,
                       size_t d0, float *out);
void so_bmm(const float *A, const float *B, size_t batch, size_t rows_a,
            size_t cols_a, size_t rows_b, size_t cols_b, int trans_b,
            float alpha, float *C);
void so_sdpa(const float *q, const float *k, const float *v,
             const float *mask, size_t B, size_t H, size_t Tq, size_t S,
             size_t dh, float *out, float *attn);
void so_greedy_sample(const float *logits, size_t batch, size_t stride,
                      const uint32_t *words /* NULL => identity */,
                      uint32_t *out);

/* ---- model -------------------------------------------------------------- */
typedef struct so_param {
  const char *name;  /* Marian parameter name */
  int32_t type;      /* 0 = f32, 1 = intgemm8 (int8 [cols][rows] + f32 mult) */
  int32_t rows;      /* logical rows (K for weights; V for Wemb)            */
  int32_t cols;      /* logical cols (N for weights; D for Wemb)            */
  const void *data;
} so_param;

typedef struct so_model so_model;

so_model *so_model_create(const so_param *params, size_t n, int enc_layers,
                          int dec_layers, int heads);
void so_model_destroy(so_model *m);
int so_model_dim(const so_model *m);
int so_model_ffn(const so_model *m);
int so_model_vocab(const so_model *m);
/* 1 = recompute cross-attention K/V + PrepareBias every call exactly like the
 * reference's op sequence (cost-faithful, used by bench cpu_baseline);
 * 0 = cache them per batch (same numbers, faster tests). */
void so_model_set_reference_cost(so_model *m, int on);
void so_model_set_threads(so_model *m, int n);

/* Model.cc:195-197: index_select + transform_embedding. out [B,S,D] */
void so_embed(const so_model *m, const uint32_t *ids, size_t B, size_t S,
              float *out);
/* Transformer.cc:57-69. x [B,S,D] in, mask [B,S] additive, out [B,S,D] */
void so_encode(const so_model *m, const float *x, const float *mask, size_t B,
               size_t S, float *out);
/* debug taps: output of encoder layer `layer` only (1-based) */
void so_encoder_layer(const so_model *m, int layer, const float *x,
                      const float *mask, size_t B, size_t S, float *out,
                      float *attn /* nullable [B,H,S,S] */);
/* Transformer.cc:120-183. states [Ld][B,D] in/out; prev NULL => step 0;
 * shortlist NULL => full vocab. logits [B,N]; attn [B,H,1,S] (last layer) */
/* the attention proper of decoder layer `layer`'s cross-attention (0-based) on an already projected query:
 * FAITHFUL = dequantised K / V + scaled_dot_product_attention, PORTABLE = the hoisted order (slimt_oracle.c) */
void so_cross_attention(const so_model *m, int layer, const float *yq, const float *encoder_out,
                        const float *mask, size_t B, size_t S, float *out, float *attn);
void so_decode_step(const so_model *m, const float *encoder_out,
                    const float *mask, size_t B, size_t S, float *states,
                    const uint32_t *prev, const uint32_t *shortlist,
                    size_t n_sl, float *logits, float *attn);
/* Model.cc:111-204 greedy loop. out_ids [B,Tmax] (Tmax = max(1, (size_t)(limit*S))),
 * out_len [B], align nullable [B,Tmax,S] (rows: head 0 of last layer,
 * first lengths[b] keys, zero elsewhere). returns steps executed. */
size_t so_translate(const so_model *m, const uint32_t *src_ids,
                    const uint32_t *lengths, size_t B, size_t S,
                    const uint32_t *shortlist, size_t n_sl,
                    float limit_factor, uint32_t eos_id, uint32_t *out_ids,
                    uint32_t *out_len, float *align);
/* Input.cc:20-63: mask from lengths: 0 on tokens, -99999999 on pads */
void so_make_mask(const uint32_t *lengths, size_t B, size_t S, float *mask);

/* ---- lexical shortlist (slimt/Shortlist.{hh,cc}) -------------------------- */
/* Binary blob layout, Shortlist.hh:77-84 + Shortlist.cc:41-104: header of six
 * uint64 {magic, checksum, frequent, best, word_to_offset_size, shortlist_size},
 * then word_to_offset[uint64], then shortlist[uint32]. */
#define SO_SHORTLIST_MAGIC 0xF11A48D5013417F5ull /* Shortlist.hh:40 */
typedef struct so_shortlist {
  uint64_t frequent, best, word_to_offset_size, shortlist_size;
  const uint64_t *word_to_offset; /* into the blob */
  const uint32_t *shortlist;      /* into the blob */
} so_shortlist;
/* checksum of the blob body as ShortlistGenerator::load computes it
 * (Shortlist.cc:66-76, Utils.hh:47-67 with libstdc++'s identity std::hash) */
uint64_t so_shortlist_checksum(const void *blob, size_t size);
/* load(): 0 on success; 1 too short, 2 bad magic, 3 size mismatch, 4 checksum,
 * 5 offsets out of range, 6 last offset != size, 7 id >= target vocab
 * (the reference aborts in each case, Shortlist.cc:16-38,49-76). */
int so_shortlist_parse(const void *blob, size_t size, int check, size_t target_vocab,
                       so_shortlist *out);
/* ShortlistGenerator::generate (Shortlist.cc:115-175): words = the batch's
 * source tokens without padding (Input::words(), Input.cc:24). out: sorted
 * unique target ids, capacity target_vocab; returns their count. */
size_t so_shortlist_generate(const so_shortlist *sl, int shared, size_t source_vocab,
                             size_t target_vocab, const uint32_t *words, size_t n_words,
                             uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif
