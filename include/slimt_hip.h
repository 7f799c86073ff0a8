/*
 * slimt_hip.h -- C ABI of the MI355X (gfx950) backend for slimt's int8
 * transformer-NMT hot path. Plain pointers and sizes only; every function
 * returns 0 on success or a non-zero status (slimt_hip_last_error() gives the
 * message): a POSITIVE status is the hipError_t of a failed runtime call -- work
 * may already be queued on the context's stream, so synchronise or destroy the
 * context before freeing the buffers the call was given --, a NEGATIVE one is a
 * check of the library's own (arguments, sizes, state). No exceptions cross this boundary; nothing here falls back to a
 * CPU implementation -- without a HIP device the compute entry points fail.
 *
 * Citations are file:line in the reference checkout (jerinphilip/slimt).
 * The op-level group is what a `Provider::Hip` in slimt/QMM.cc would forward
 * to (INTEGRATION.md shows the .inl.cc); the engine-level group is what
 * `Encoder::forward` / `Decoder::step` / `Model::forward` dispatch to under
 * SLIMT_HAS_HIP.
 *
 * Weight layout at this boundary is slimt's canonical prepared layout: int8
 * [N][K], K contiguous (== the Marian intgemm8 payload, slimt/Io.cc:225-239),
 * i.e. what prepare_weight_quantized_transposed below emits.
 */
#ifndef SLIMT_HIP_H
#define SLIMT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIMT_HIP_ABI_VERSION 3

/* ---- status -------------------------------------------------------------- */
int slimt_hip_abi_version(void);
const char *slimt_hip_last_error(void); /* thread-local, never NULL */
int slimt_hip_device_count(int *count);

/* ---- process-wide setup (optional) ---------------------------------------
 * The library itself never reads or writes the environment when it is loaded.
 * One HIP stream per translate worker: the HIP runtime multiplexes streams onto
 * FOUR hardware queues unless GPU_MAX_HW_QUEUES says otherwise, which caps the
 * batches really in flight (20 blocking workers: 10.4 M tok/s with 4 queues,
 * 26.5 M with 32). The runtime reads that variable once, at ITS first call from
 * anywhere in the process. _request_hw_queues(n) sets it (only if the process
 * has not chosen a value) and returns 0, or returns 1 -- changing nothing -- when
 * this library has already called into HIP (too late). It calls setenv: call it
 * from main() before threads start and before any model is created, or set the
 * variable in the launcher instead (bench.py, slimt_amd.frontend and host_test
 * do one of the two; host/Service warns once when it is started with more than
 * two workers per device and fewer than 8 queues). _hw_queues() returns the
 * value the environment holds now (0 = unset). */
int slimt_hip_request_hw_queues(int n);
int slimt_hip_hw_queues(void);

/* ---- op level: slimt::qmm::* (slimt/QMM.hh:48-63) ------------------------ */
/* Host pointers in, host pointers out (H2D, kernels, D2H on `device` 0
 * unless slimt_hip_set_device was called on this thread). Stateless and
 * re-entrant like the reference providers (slimt/QMM.cc:36-75). */
int slimt_hip_set_device(int device);

/* qmm::affine (QMM.hh:48; Intgemm.inl.cc:92-156). bias == NULL gives
 * qmm::dot (QMM.hh:56; Intgemm.inl.cc:158-226). x f32 [M,K] row-major,
 * W int8 [N][K], y f32 [M,N]. Requires K % 64 == 0 (intgemm's own tile
 * constraint), K <= 4096. */
int slimt_hip_affine(const float *x, size_t M, size_t K, const int8_t *W_nk,
                     size_t N, const float *bias, float a_quant,
                     float b_quant, float *y);
/* qmm::affine_with_select (QMM.hh:51; Intgemm.inl.cc:7-90): y [M,n_idx],
 * column c is vocabulary id idx[c]. */
int slimt_hip_affine_select(const float *x, size_t M, size_t K,
                            const int8_t *W_nk, size_t N, const float *bias,
                            float a_quant, float b_quant, const uint32_t *idx,
                            size_t n_idx, float *y);
/* Debug/parity: the raw int32 accumulators of Int8Shift::Multiply,
 * accS[i,j] = sum_k (q[i,k]+127) * W[k,j]  (Intgemm.inl.cc:149-153). */
int slimt_hip_affine_acc_i32(const float *x, size_t M, size_t K,
                             const int8_t *W_nk, size_t N, float a_quant,
                             int32_t *accS);
/* qmm::prepare_weight_transposed (QMM.hh:59; Intgemm.inl.cc:228-235):
 * weights f32 B^T [rows][cols] -> int8 canonical [rows][cols]. Runs on the
 * host (load-time only, slimt/Io.cc:215). */
int slimt_hip_prepare_weight_transposed(const float *weights, int8_t *prepared,
                                        float quantization_multiplier,
                                        size_t cols, size_t rows);
/* qmm::prepare_weight_quantized_transposed (QMM.hh:62;
 * Intgemm.inl.cc:237-243): the canonical layout IS the file layout, so this
 * is a copy (like the Ruy provider, Ruy.inl.cc:267-274). Host side. */
int slimt_hip_prepare_weight_quantized_transposed(const int8_t *input,
                                                  int8_t *output, size_t rows,
                                                  size_t cols);

/* ---- op level: slimt/TensorOps.hh float ops ------------------------------ */
/* layer_norm (TensorOps.cc:542-580) */
int slimt_hip_layer_norm(const float *x, const float *scale, const float *bias,
                         float eps, size_t rows, size_t cols, float *y);
/* softmax (TensorOps.cc:282-315) */
int slimt_hip_softmax(const float *x, size_t rows, size_t cols, float *y);
/* highway (TensorOps.cc:662-682): out = sigmoid(g)*x + (1-sigmoid(g))*y */
int slimt_hip_highway(const float *x, const float *y, const float *g, size_t n,
                      float *out);
/* scaled_dot_product_attention (Modules.cc:24-86) on already split heads:
 * q [B,H,Tq,dh], k,v [B,H,S,dh], mask [B,S] additive -> out [B,H,Tq,dh],
 * attn [B,H,Tq,S] (nullable). */
int slimt_hip_sdpa(const float *q, const float *k, const float *v,
                   const float *mask, size_t B, size_t H, size_t Tq, size_t S,
                   size_t dh, float *out, float *attn);

/* ---- engine level -------------------------------------------------------- */
typedef struct slimt_hip_model slimt_hip_model; /* weights on one device */
typedef struct slimt_hip_ctx slimt_hip_ctx;     /* stream + workspace, one per
                                                   worker thread (Frontend.cc:212-226) */

/* One named tensor as held by slimt::Transformer::items_ after
 * io::load_items, or straight from the Marian .bin (Io.cc:114-161). */
typedef struct slimt_hip_param {
  const char *name; /* Marian parameter name (Modules.cc:336-406) */
  int32_t type;     /* 0 = f32 [rows,cols]; 1 = intgemm8: int8 payload in file
                       order ([cols][rows]; Wemb: [rows][cols]) followed by one
                       f32 quantisation multiplier (Io.cc:225-239) */
  int32_t rows;     /* shape[-2] */
  int32_t cols;     /* shape[-1] */
  const void *data; /* host memory, borrowed for the duration of the call */
  uint64_t bytes;   /* size of `data`; checked against rows * cols (+ the trailing
                       multiplier) before anything is read. 0 = not stated */
} slimt_hip_param;

typedef struct slimt_hip_dims { /* Model::Config (Model.hh:33-51) */
  int32_t encoder_layers;
  int32_t decoder_layers;
  int32_t num_heads;
} slimt_hip_dims;

/* Transformer::Transformer (Transformer.cc:87-94) + load_parameters
 * (:185-225) + the Wemb handling of Io.cc:182-224: uploads and re-tiles the
 * int8 weights for the MFMA B operand, precomputes column sums and prepared
 * biases. Unknown names are ignored, missing ones are an error. */
int slimt_hip_model_create(const slimt_hip_param *params, size_t n_params,
                           const slimt_hip_dims *dims, int device,
                           slimt_hip_model **out);
/* The same from the Marian .bin container itself (the `View model` that
 * Transformer::Transformer receives, Transformer.cc:87-94; format Io.cc:114-161):
 * the items are located inside `bin` (bounds-checked, nothing is copied on the
 * host) and handed to slimt_hip_model_create. */
int slimt_hip_model_create_from_bin(const void *bin, size_t size,
                                    const slimt_hip_dims *dims, int device,
                                    slimt_hip_model **out);
int slimt_hip_model_destroy(slimt_hip_model *model);
/* Admission of the persistent decoders of all contexts of `model`: at most
 * about `workgroups` decoder workgroups (one CU each, 16 sentences) run at a
 * time, later launches wait on their own stream; the remaining CUs stay with
 * the encoders of the batches behind them. Default: 7/8 of the device's CUs;
 * 0 = no limit. Results do not depend on it. */
int slimt_hip_model_set_decoder_budget(slimt_hip_model *model, int workgroups);
/* Cache policy of the persistent decoder's K/V cache loads: 0 (default) = chosen
 * per launch (non-temporal once the K/V of the contexts with a pending decoder
 * exceeds what the Infinity Cache can serve), 1 = always
 * temporal, 2 = always non-temporal. Results do not depend on it. Needs the
 * decoder admission (budget > 0) for 0 and 2. */
int slimt_hip_model_set_kv_cache_policy(slimt_hip_model *model, int policy);
/* XCD-affine placement of the persistent decoder (needs the decoder admission, budget > 0):
 * 0 (default) = a batch's workgroups run wherever the dispatcher puts them; 1 / 2 / 4 = a batch
 * of up to 16 / 32 / 64 workgroups is over-launched on every XCD and its tiles are claimed only by
 * workgroups that find themselves (HW_REG_XCC_ID) on the batch's one / two / four home XCDs, so that the
 * batch's own shortlisted output layer streams through those L2s only (each MI355X XCD has a private
 * 4 MB L2). Placement changes speed only; results do not depend on it. */
int slimt_hip_model_set_xcd_affinity(slimt_hip_model *model, int xcds);
/* Storage format of the cross-attention K/V cache that slimt_hip_translate* keeps between
 * its encoder and decoder launches (the reference recomputes K and V every step,
 * slimt/Modules.cc:248-249). Every format caches the int8 GEMM's ACCUMULATORS, and the
 * attention applies the projections' unquantisation multiplier and prepared bias after its
 * sums (the same real numbers as the reference's dequantise-then-attend, other
 * roundings, <= 2.5e-5 apart; every format gives the same floats as every other):
 *   0 (default) = packed integers where the kernels support it (emb 256 / head dim 32 with
 *       sources of up to 128 tokens, emb 512 / head dim 64 up to 32), f32 elsewhere. Packed
 *       means, per sentence and decoder layer and decided by the encoder: 16 bits per value
 *       where every accumulator less its column's centre lies in [-2^15, 2^15)
 *       (slimt_hip_model_set_kv_centres; the first batch of >= 1024 rows is cached as f32
 *       to calibrate the centres when none were set), else 20 bits where every accumulator
 *       lies in [-2^19, 2^19), else 24 bits (holds any accumulator);
 *   1 = always f32 (float(acc), exact);
 *   2 = packed, always 24 bits;
 *   3 = f32, and the attention in the reference's LITERAL sequence: every cached value
 *       dequantised first (k = float(acc) * unquant + bias, Intgemm.inl.cc:146-153), then
 *       Modules.cc:24-86 on those floats. Runs the decoder one launch per stage (the
 *       persistent decoder has the hoisted order only): for checking, not for speed.
 * Formats 0 / 1 / 2 give the same floats as each other; format 3 differs from them by
 * roundings only (<= 2.5e-5 of the context vectors' scale), which can move an arg-max on a
 * near-tie: see DESIGN 2 for what that means for translate output. */
int slimt_hip_model_set_kv_cache_format(slimt_hip_model *model, int format);
/* Sentences per workgroup of the persistent decoder in decode mode 0 (needs the decoder admission,
 * budget > 0): on (default) = a launch uses 8 or 4 sentences per workgroup instead of 16 while the
 * decoders of all contexts with one pending still fit the budget at that size -- a batch of 256 alone
 * then runs on 64 CUs instead of 16 (the reference's default is ONE worker, slimt/Frontend.hh:25) --,
 * off = always 16. Results do not depend on it. */
int slimt_hip_model_set_adaptive_decoder_rows(slimt_hip_model *model, int on);
int slimt_hip_model_info(const slimt_hip_model *model, int32_t *dim_emb,
                         int32_t *dim_ffn, int32_t *vocab, int32_t *heads);
/* the device the model's weights live on (-1 for NULL) */
int slimt_hip_model_device(const slimt_hip_model *model);

/* stream: a hipStream_t to run on (borrowed), or NULL to create one. */
int slimt_hip_ctx_create(slimt_hip_model *model, size_t max_batch,
                         size_t max_source_length, void *stream,
                         slimt_hip_ctx **out);
/* Same, for token-budget batching (slimt/Batcher.cc:95-120: many short
 * sentences or few long ones, (B + 1) * S <= max_words): the workspace holds
 * batches with B <= max_batch, S <= max_source_length and B * S <= max_tokens. */
int slimt_hip_ctx_create_budget(slimt_hip_model *model, size_t max_batch,
                                size_t max_source_length, size_t max_tokens, void *stream,
                                slimt_hip_ctx **out);
int slimt_hip_ctx_destroy(slimt_hip_ctx *ctx);
/* Contexts alive on `device` in this process. A context is a stream; past 22 of
 * them on one device the hardware queues are time-sliced whatever
 * GPU_MAX_HW_QUEUES says (20 contexts 33 M tok/s, 24: 23.8 M, 32: 20.4 M on the
 * headline workload), so the library says so once on stderr when the 23rd is
 * created (SLIMT_HIP_QUIET=1 silences it). slimt runs `workers` Async threads
 * (Frontend.hh:25), one context each: keep workers <= 22 per device and process. */
int slimt_hip_contexts_on_device(int device, int *count);
int slimt_hip_ctx_stream(slimt_hip_ctx *ctx, void **stream);
int slimt_hip_ctx_synchronize(slimt_hip_ctx *ctx);
/* Execution strategy of slimt_hip_translate* / slimt_hip_encode: 0 = automatic
 * (persistent fused encoder / decoder kernels when the model shape supports
 * them), 1 = one launch per stage (and per decode step; the kernels behind
 * slimt_hip_decode_step), 2 / 3 / 4 / 5 = automatic, but the persistent decoder
 * is forced to 16 / 32 / 8 / 4 sentences per workgroup where it has that
 * variant, 6 = automatic, but output layers are shared by clusters of four
 * 16-sentence workgroups where the kernel has that (emb 256, sources of up to 32
 * tokens, decoder admission on) (tuning and tests; 0 picks 32 for output layers
 * of more than 16k columns -- clusters measured 6 % below it on the full
 * vocabulary --, and 8 or 4 while CUs would idle:
 * slimt_hip_model_set_adaptive_decoder_rows). Same results in every mode. */
int slimt_hip_ctx_set_decode_mode(slimt_hip_ctx *ctx, int mode);
/* Rows (source tokens) per workgroup of the persistent encoder for emb 256 models: 0
 * (default) = chosen per call (64-row tiles from 32 of them on),
 * 32 or 64 = forced (tuning and tests). Same results either way. */
int slimt_hip_ctx_set_encode_rows(slimt_hip_ctx *ctx, int rows);
/* Which kernels a translate call with source length S would use in the current
 * mode: *encoder_fused / *decoder_fused = 1 for the persistent kernels, 0 for
 * the per-stage ones. */
int slimt_hip_ctx_plan(const slimt_hip_ctx *ctx, size_t S, int *encoder_fused,
                       int *decoder_fused);

/* Model::forward (Model.cc:187-204) = embed + Encoder::forward + the greedy
 * loop of Model::decode (Model.cc:111-185). Host buffers.
 *  src_ids  [B,S] padded token ids, lengths [B]
 *  shortlist sorted unique target ids (Shortlist.cc:115-175), n_shortlist == 0
 *            => full vocabulary (Transformer.cc:181)
 *  out_ids  [B,Tmax], Tmax = max(1, (size_t)(limit_factor * S)): the first step is
 *           unconditional, the loop then runs while i < limit_factor * S (Model.cc:144-161)
 *  out_len  [B] tokens recorded per sentence, EOS included (Model.cc:127-137)
 *  align    nullable [B,Tmax,S]: row t = attention of head 0 of the LAST
 *           decoder layer over the first lengths[b] keys (Model.cc:84-108) */
int slimt_hip_translate(slimt_hip_ctx *ctx, const uint32_t *src_ids,
                        const uint32_t *lengths, size_t B, size_t S,
                        const uint32_t *shortlist, size_t n_shortlist,
                        float limit_factor, uint32_t eos_id, uint32_t *out_ids,
                        uint32_t *out_len, float *align);
/* Same without the final wait: the work is queued on the ctx stream and the call
 * returns; slimt_hip_ctx_synchronize(ctx) waits for it. Every buffer must stay
 * valid (and unchanged) until then, and should be pinned (slimt_hip_host_alloc):
 * with src_ids, lengths, out_ids, out_len (and align) all pinned the persistent
 * kernels read and write them in host memory themselves and no copy is queued at
 * all; otherwise H2D / D2H copies bracket the kernels (and asynchronous copies of
 * many contexts queue behind each other's kernels). The shortlist is uploaded
 * only when it differs from the previous call's. One call in flight per ctx: a
 * worker that wants to assemble its next batch while this one runs uses two
 * contexts (host/Service.cc). */
int slimt_hip_translate_async(slimt_hip_ctx *ctx, const uint32_t *src_ids,
                              const uint32_t *lengths, size_t B, size_t S,
                              const uint32_t *shortlist, size_t n_shortlist,
                              float limit_factor, uint32_t eos_id,
                              uint32_t *out_ids, uint32_t *out_len, float *align);
/* Pinned (page-locked) host memory for the staging buffers of the call above. */
int slimt_hip_host_alloc(size_t bytes, void **out);
int slimt_hip_host_free(void *p);
/* Same with every buffer already resident in device memory; asynchronous on
 * the ctx stream when `steps_hint` > 0 (runs exactly that many decode steps,
 * no early-exit read-back), otherwise syncs every few steps to stop as soon
 * as every sentence has emitted EOS.
 * Wait for an asynchronous call with slimt_hip_ctx_synchronize(ctx), not on the
 * stream yourself: a bounded wait INSIDE the launches (the in-launch shortlist
 * hand-over, the cluster logits' hand-overs) that ran out is reported through a
 * word only that function reads and clears -- it then fails with the reason, and
 * the batch's outputs must be discarded. A caller that waits on the stream alone
 * sees rc 0 for such a batch and the error on the NEXT call that does synchronise
 * through the library. */
/* Device arrays cannot be checked by the host: a token or shortlist id >= the vocabulary reads the table's last
 * row instead of faulting -- that sentence's result is undefined (as in the reference, which does not check
 * either), the other sentences' results are not affected. */
int slimt_hip_translate_device(slimt_hip_ctx *ctx, const uint32_t *d_src_ids,
                               const uint32_t *d_lengths, size_t B, size_t S,
                               const uint32_t *d_shortlist, size_t n_shortlist,
                               float limit_factor, uint32_t eos_id,
                               uint32_t *d_out_ids, uint32_t *d_out_len,
                               float *d_align, int steps_hint);

/* ---- several batches in one launch pair -----------------------------------
 * What `workers` concurrent Model::forward calls are in the reference (Async,
 * slimt/Frontend.cc:207-227, each worker translating the batch the Batcher hands
 * it, Batcher.cc:95-120): n batches, each with its own padded source length (at
 * most S, the launch's), its own arrays, its own shortlist and its own outputs,
 * translated by ONE encoder and ONE decoder launch on ctx's stream. A batch padded
 * to fewer tokens than the launch keeps ITS length's step limit and alignment
 * width (Model.cc:159-161, 84-108); the extra positions are padding like any
 * other (masked keys of weight exactly 0, Input.cc:49-63). The reference's default batch is 1024
 * padded tokens (Frontend.hh:21-39: 32 sentences of 32 tokens) -- one such batch
 * per launch pair occupies 2 of 256 CUs in the decoder; merged, k of them fill
 * what one batch of k x B would. Every sentence's arithmetic is independent of
 * its neighbours, so each batch's outputs equal those of its own
 * slimt_hip_translate* call bit for bit (tests/test_gpu_translate_many.py).
 *
 * The launch works on sum_j roundup(B_j, 32) "global" sentences (_rows below),
 * every one padded to the call's S: ctx must hold that many (max_batch) and that
 * many times S padded tokens.
 * Limits: n <= 8; the persistent kernels for sources of up to 64 tokens. Whatever
 * cannot be merged (more batches, longer sources, decode modes 1 / 6, K/V cache
 * format 3) is translated batch by batch, in order, on the same stream: same
 * results, same asynchrony. Batches that give the SAME shortlist pointer and
 * size share one packed output layer. */
typedef struct slimt_hip_batch {
  const uint32_t *src_ids;   /* [B][S] */
  const uint32_t *lengths;   /* [B] */
  size_t B;
  size_t S;                  /* this batch's padded length, <= the call's S; 0 = the call's S */
  const uint32_t *shortlist; /* sorted unique target ids, or NULL with n_shortlist == 0: full vocabulary */
  size_t n_shortlist;
  uint32_t *out_ids;         /* [B][Tmax], Tmax = max(1, (size_t)(limit_factor * S)), S = this batch's */
  uint32_t *out_len;         /* [B] */
  float *align;              /* nullable [B][Tmax][S] */
} slimt_hip_batch;

/* global sentences a merged launch of these batch sizes occupies in its context */
size_t slimt_hip_translate_many_rows(const size_t *B, size_t n_batches);
/* every array (shortlists included) resident in device memory; steps_hint as for _translate_device */
int slimt_hip_translate_many_device(slimt_hip_ctx *ctx, const slimt_hip_batch *batches, size_t n_batches,
                                    size_t S, float limit_factor, uint32_t eos_id, int steps_hint);
/* host arrays, asynchronous like _translate_async (slimt_hip_ctx_synchronize waits): ids, lengths and
 * outputs should be pinned (slimt_hip_host_alloc) -- the kernels then read and write them in place;
 * batches whose arrays are not are translated one by one through _translate_async. The shortlists
 * are host arrays; the merged path needs them to be ONE array (or none) for all batches. */
int slimt_hip_translate_many_async(slimt_hip_ctx *ctx, const slimt_hip_batch *batches, size_t n_batches,
                                   size_t S, float limit_factor, uint32_t eos_id);

/* Step-wise mirrors for parity tests ------------------------------------- */
/* Model.cc:195-201: embed + Encoder::forward (Transformer.cc:57-69). Keeps
 * the encoder output in the ctx for slimt_hip_decode_*. enc_out nullable
 * host [B,S,D]; layer_out nullable host [Le,B,S,D] (every layer's output);
 * embed_out nullable host [B,S,D]. */
int slimt_hip_encode(slimt_hip_ctx *ctx, const uint32_t *src_ids,
                     const uint32_t *lengths, size_t B, size_t S,
                     float *embed_out, float *layer_out, float *enc_out);
/* Decoder::start_states (Transformer.cc:78-85) + per-batch setup (cross
 * attention K/V, shortlist gather). */
int slimt_hip_decode_begin(slimt_hip_ctx *ctx, const uint32_t *shortlist,
                           size_t n_shortlist);
/* Decoder::step (Transformer.cc:120-183). prev == NULL => first step.
 * logits host [B,N] (N = n_shortlist or vocab), attn nullable host [B,H,S]
 * (last decoder layer), states nullable host [Ld,B,D] (SSRU cells after the
 * step). */
int slimt_hip_decode_step(slimt_hip_ctx *ctx, const uint32_t *prev,
                          float *logits, float *attn, float *states);

/* The same three steps with the reference's own ARGUMENTS, for the class-level
 * mirror (host/Transformer.hh): Encoder::forward takes the transformed embedding
 * and the mask (Transformer.cc:57-69), Decoder::step takes encoder_out, mask and
 * the caller's states on every call (Transformer.cc:120-183).
 * _encode_embedded: embedding host [B,S,D] (after transform_embedding), lengths
 *   [B] (= the mask: the first lengths[b] keys are tokens) -> enc_out host [B,S,D].
 *   Always the per-stage kernels (the persistent encoder starts from token ids).
 * _decode_begin_from: like _decode_begin for an encoder output handed in from
 *   the host (uploads it, computes the cross-attention K/V, gathers the shortlist).
 * _decode_step_states: like _decode_step, but the SSRU cells are the caller's:
 *   states_in [Ld,B,D] is uploaded first, states_out receives them after the step. */
int slimt_hip_encode_embedded(slimt_hip_ctx *ctx, const float *embedding,
                              const uint32_t *lengths, size_t B, size_t S,
                              float *enc_out);
int slimt_hip_decode_begin_from(slimt_hip_ctx *ctx, const float *encoder_out,
                                const uint32_t *lengths, size_t B, size_t S,
                                const uint32_t *shortlist, size_t n_shortlist);
int slimt_hip_decode_step_states(slimt_hip_ctx *ctx, const uint32_t *prev,
                                 const float *states_in, float *logits,
                                 float *attn, float *states_out);

/* ---- lexical shortlist (next row f3: slimt/Shortlist.{hh,cc}) --------------
 * Replaces ShortlistGenerator (Shortlist.hh:38-90): _create = the constructor's
 * load() over the binary shortlist blob (Shortlist.cc:41-104; layout
 * Shortlist.hh:77-84), _generate = generate() (Shortlist.cc:115-175) as Model::
 * forward calls it per batch on Input::words() (Model.cc:117-120, Input.cc:24).
 * The blob is copied to the device; the caller keeps ownership of `blob`.
 * check != 0 also verifies the header checksum (Shortlist.cc:66-76). Offsets and
 * ids are always range-checked at load (out-of-range data is undefined
 * behaviour in the reference; here it is rejected). */
typedef struct slimt_hip_shortlist slimt_hip_shortlist;
int slimt_hip_shortlist_create(const void *blob, size_t blob_size, size_t source_vocab,
                               size_t target_vocab, int shared, int check, int device,
                               slimt_hip_shortlist **out);
int slimt_hip_shortlist_destroy(slimt_hip_shortlist *sl);
/* header fields (Shortlist.cc:78-81) */
int slimt_hip_shortlist_info(const slimt_hip_shortlist *sl, uint64_t *frequent, uint64_t *best);
/* Threading: a handle may be shared by any number of threads, like the reference's
 * const generate() on one shared generator (Model.cc:117-120). _generate stages
 * through buffers of the handle and serialises its callers internally;
 * _generate_device / slimt_hip_translate_device_generated only read the handle and
 * use the calling context's scratch and stream, so they run concurrently (one
 * context per thread, as everywhere).
 * Host arrays: src_ids [B][S] padded rows, lengths [B] (only the first lengths[b]
 * tokens of a row are words). out_ids: capacity >= target_vocab; *n_out = number
 * of ids written (sorted, unique, a multiple of 8 when enough ids are free). */
int slimt_hip_shortlist_generate(slimt_hip_shortlist *sl, const uint32_t *src_ids,
                                 const uint32_t *lengths, size_t B, size_t S, uint32_t *out_ids,
                                 size_t *n_out);
/* Device arrays, asynchronous on ctx's stream (d_out_ids: capacity >=
 * target_vocab uint32, d_n_out: one uint32), e.g. ahead of
 * slimt_hip_translate_device on the same context. */
int slimt_hip_shortlist_generate_device(slimt_hip_shortlist *sl, slimt_hip_ctx *ctx,
                                        const uint32_t *d_src_ids, const uint32_t *d_lengths,
                                        size_t B, size_t S, uint32_t *d_out_ids,
                                        uint32_t *d_n_out);

/* Model::forward with its shortlist step on the device (Model.cc:117-120,195-203):
 * generates the batch's lexical shortlist on ctx's stream, then translates with
 * it. With the persistent kernels nothing returns to the host in between (the
 * shortlist's size stays on the device: three launches in all, asynchronous like
 * slimt_hip_translate_device); with the stage kernels one 4-byte read-back sizes
 * the launches. The shortlist's target vocabulary must be the model's. */
int slimt_hip_translate_device_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *shortlist,
                                         const uint32_t *d_src_ids, const uint32_t *d_lengths,
                                         size_t B, size_t S, float limit_factor, uint32_t eos_id,
                                         uint32_t *d_out_ids, uint32_t *d_out_len, float *d_align,
                                         int steps_hint);

/* The same on HOST buffers -- Model::forward as the reference's workers call it (Model.cc:111-204),
 * shortlist step included: _translate_generated waits for the result, _translate_async_generated
 * queues the work on ctx's stream like slimt_hip_translate_async (same buffer rules: with every
 * array pinned the kernels read ids / lengths from, and write tokens, lengths and alignment rows to,
 * host memory themselves; nothing is copied, nothing synchronises, and no host shortlist exists at
 * all). A handle may be shared by every context of its device. */
int slimt_hip_translate_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *shortlist,
                                  const uint32_t *src_ids, const uint32_t *lengths, size_t B, size_t S,
                                  float limit_factor, uint32_t eos_id, uint32_t *out_ids,
                                  uint32_t *out_len, float *align);
int slimt_hip_translate_async_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *shortlist,
                                        const uint32_t *src_ids, const uint32_t *lengths, size_t B,
                                        size_t S, float limit_factor, uint32_t eos_id,
                                        uint32_t *out_ids, uint32_t *out_len, float *align);

/* Several batches in one launch pair (slimt_hip_translate_many_* above) with every batch's OWN lexical shortlist, generated from its source words inside the encoder launch
 * (Model.cc:117-120 per batch; slimt_hip_translate_*_generated below): the first n workgroups to start each
 * generate one batch's list and publish it, every workgroup packs its share of the n output layers at the end of
 * its encoder work. batches[].shortlist / n_shortlist are ignored. Falls back batch by batch like the others
 * (also when the generator's bitmaps do not fit the encoder's LDS). */
int slimt_hip_translate_many_device_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const slimt_hip_batch *batches,
                                              size_t n_batches, size_t S, float limit_factor, uint32_t eos_id, int steps_hint);
int slimt_hip_translate_many_async_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const slimt_hip_batch *batches,
                                             size_t n_batches, size_t S, float limit_factor, uint32_t eos_id);


/* ---- measurement --------------------------------------------------------- */
/* When enabled, HIP events bracket every launch of kernel family `kernel_id`
 * on the ctx stream; slimt_hip_profile_read returns the number of launches
 * and their summed duration since the last reset. */
enum {
  SLIMT_HIP_K_NONE = 0,
  SLIMT_HIP_K_GEMM_ENC = 1,    /* encoder projections (M = B*S rows)       */
  SLIMT_HIP_K_GEMM_DEC = 2,    /* decoder projections (M = B rows)         */
  SLIMT_HIP_K_LOGITS = 3,      /* output projection + fused argmax         */
  SLIMT_HIP_K_ATTN_ENC = 4,
  SLIMT_HIP_K_ATTN_DEC = 5,
  SLIMT_HIP_K_SSRU = 6,
  SLIMT_HIP_K_DECODE_FUSED = 7, /* persistent whole-loop decoder           */
  SLIMT_HIP_K_ENCODE_FUSED = 8, /* persistent whole-stack encoder          */
  SLIMT_HIP_K_COUNT = 9
};
int slimt_hip_profile_enable(slimt_hip_ctx *ctx, int kernel_id);
int slimt_hip_profile_read(slimt_hip_ctx *ctx, uint64_t *launches,
                           double *total_ms, double *int8_macs,
                           double *weight_bytes);
int slimt_hip_profile_reset(slimt_hip_ctx *ctx);
/* Diagnostic: returns (into out[0..n), n <= 64) the 100 MHz wall-clock stamps
 * the persistent decoder's workgroup 0 wrote at its phase boundaries during
 * the previously selected step, then selects `step` for the next translate
 * call (step < 0 disables stamping). */
int slimt_hip_debug_decode_stamps(slimt_hip_ctx *ctx, int step, uint64_t *out,
                                  size_t n);
/* Diagnostic: the decoder's cross-attention proper (Modules.cc:24-86 as Attention::forward calls it,
 * Modules.cc:287-306) of decoder layer `layer` on the caller's projected queries yq [B][D], over
 * the f32 K/V cache of ctx's current batch (slimt_hip_decode_begin[_from] first): joined [B][D] =
 * the joined heads before the output projection, attn (nullable) [B][H][S] = the probabilities.
 * literal = 0: the hoisted order every decoder here runs; 1: the reference's literal sequence
 * (K/V cache format 3). Tests bound one against the other and both against the checker. */
int slimt_hip_debug_cross_attention(slimt_hip_ctx *ctx, int layer, int literal, const float *yq,
                                    float *joined, float *attn);
/* Diagnostic: break (broken != 0) or restore the hand-over of a shortlist generated inside the
 * encoder launch (slimt_hip_translate*_generated): the waiting workgroups then look for a
 * publication that never comes and give up after `poll_limit` polls (1..2^24; the default 2^24
 * is about two seconds). A waiter that gives up packs nothing and the context's next wait --
 * slimt_hip_ctx_synchronize, or the blocking translate calls -- FAILS: the batch's results are
 * not to be used. Tests use it to reach that path. */
int slimt_hip_debug_break_shortlist_handoff(slimt_hip_ctx *ctx, int broken, unsigned poll_limit);
/* Diagnostic: which form each sentence-layer of ctx's last batch was cached in -- out[l * B + b],
 * 0 = 20-bit, 1 = 24-bit, 2 = 16-bit (the tight form); *batch = B, or 0 when the batch's caches
 * are all in one form (f32 or 24-bit: formats 1 / 2, a shape without the narrow form, or the
 * centres' calibration batch). Waits for ctx's stream. */
int slimt_hip_debug_kv_formats(slimt_hip_ctx *ctx, uint8_t *out, size_t n, size_t *batch);
/* Diagnostic: format 0's watch. The library counts the sentence-layers it cached (submitted) and those
 * that needed the 24-bit form (wide; updated by the device, a few batches behind); once more than one
 * in 32 did (after 1024 were submitted) the model caches every batch in the 24-bit form from then on
 * (*switched_to_24_bit = 1): a model whose accumulators do not fit 20 bits then runs at the 24-bit
 * form's own speed instead of through its per-sentence fallback. slimt_hip_model_set_kv_cache_format
 * and slimt_hip_debug_kv_narrow_limit start the watch afresh. Any pointer may be NULL. */
int slimt_hip_debug_kv_watch(slimt_hip_model *model, int *switched_to_24_bit, uint64_t *wide, uint64_t *submitted);
/* Diagnostic: accumulators must lie in [-limit, limit) for the 20-bit form (default and
 * maximum 2^19, what 20 bits hold). Tests lower it so that some sentences of a batch take the
 * 24-bit form next to narrow ones. Results do not depend on it. */
int slimt_hip_debug_kv_narrow_limit(slimt_hip_model *model, int limit);
/* Diagnostic: the tight (16-bit) form below the 20-bit one. Where the decoder has a reader for it (D = 256 /
 * F = 1536 with sentences of up to 128 tokens, D = 512 / F = 2048 up to 32) format 0 caches a
 * sentence-layer whose K and V accumulators, less their columns' centres (slimt_hip_model_set_kv_centres),
 * all lie in [-limit, limit) as plain int16 (default and maximum 2^15; 0 = never tried; tests lower it so
 * that a batch mixes all three forms). Results do not depend on it. Starts the watches afresh. */
int slimt_hip_debug_kv_tight_limit(slimt_hip_model *model, int limit);
/* The tight form's per-column centres, [Ld][K, V][D] int32 (n = Ld * 2 * D, each within (-2^24, 2^24): a float holds it exactly):
 * the 16-bit form caches accumulator - centre, the decoder adds the centre back (exact), so the centres
 * decide which sentences fit the form and nothing else -- every result is the same for any centres.
 * Without this call the library calibrates them itself: the first batch of at least 1024 rows that could
 * take the form is cached as f32, its column means (floor(sum / rows + 1/2), integer arithmetic) become
 * the centres, and the form is tried from the first batch submitted after that reduction has finished.
 * Call it before the first translate, or with no batch of this model in flight; it fails while a
 * calibration batch is in flight. A deployment that wants the same form for the same sentence on
 * every run sets centres it has stored (slimt_hip_debug_kv_centres reads the calibrated ones). */
int slimt_hip_model_set_kv_centres(slimt_hip_model *model, const int32_t *centres, size_t n);
/* Diagnostic: *ready = 1 and out[0 .. Ld * 2 * D) = the centres once they exist (set, or calibrated and
 * the reduction finished), else *ready = 0. out may be NULL (state only). */
int slimt_hip_debug_kv_centres(slimt_hip_model *model, int32_t *out, size_t n, int *ready);
/* Diagnostic: the tight form's watch, per decoder layer l < 4: submitted[l] sentences were allowed to
 * try it, missed[l] of them did not fit (updated by the device, a few batches behind); once more than
 * one in 32 of a layer's sentences missed (after 1024 were submitted) that layer stops trying (bit l of
 * *layers_off): the 20-bit form is an out-of-line fallback in the kernels with the tight reader. Any pointer may be NULL; the arrays hold 4 entries. */
int slimt_hip_debug_kv_tight_watch(slimt_hip_model *model, unsigned *layers_off, uint64_t *missed, uint64_t *submitted);
/* Re-calibration of the centres (round 6): the first calibration batch need not look like the traffic behind
 * it, so the first `max_recalibrations` times a layer's watch trips (above) the engine starts a new GENERATION
 * of centres instead of switching the layer off: the next batch of >= 1024 rows is cached as f32 and calibrates
 * them (into a buffer of its own: batches in flight keep the generation they were encoded with), and the form
 * is tried again. Only a trip past that limit switches a layer off for good. Default 2, at most 3;
 * max_recalibrations < 0 leaves the limit alone. *generations_started (nullable): 0 until the first trip.
 * The counters of _kv_tight_watch are those of the current generation. Results never depend on any of this. */
int slimt_hip_debug_kv_recalibrations(slimt_hip_model *model, int *generations_started, int max_recalibrations);
/* Diagnostic (process-wide): while device_buf != NULL, thread 0 of every
 * workgroup of the persistent encoder / decoder appends a begin and an end
 * event to it: device_buf[0] = event counter (zero it first), then 3 uint64
 * per event {kernel (1 = decoder, 2 = encoder) | end << 8 | blockIdx << 16,
 * HW_ID | XCC_ID << 32, 100 MHz wall clock}; events past `capacity` are
 * dropped. NULL switches it off. tools/occupancy_trace.py reads it. */
int slimt_hip_debug_occupancy_trace(void *device_buf, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* SLIMT_HIP_H */
