// Host-callable launchers for the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace slimt_hip {

// A prepared (device-resident) int8 weight: MFMA-B-fragment tiling + the
// constants of the intgemm epilogue.
//   Wp     [(n_tile, k_step, lane)] x 16 B: lane l of k-step s of tile t holds
//          W[n = 16 t + (l & 15)][k = 64 s + 16 (l >> 4) .. +15]
//   colsum [n]  sum_k W[k, n]
//   pb     [n]  float(colsum[n]) * (-127/(aq*bq)) + bias[n]   (PrepareBias)
//   u           1 / (aq * bq)
struct PreparedWeight {
  const void *Wp = nullptr;
  const int *colsum = nullptr;
  const float *pb = nullptr;
  // The same epilogue constants once more, laid out for a wave that streams tiles t, t + 16, t + 32, ... (the
  // persistent decoder): per PAIR of such tiles and column one 16-byte quad {colsum(t), pb(t), colsum(t + 16),
  // pb(t + 16)}, pair (t, t + 16) at index 16 (t / 32) + t % 16 -- one load instruction per two tiles where colsum
  // and pb cost two per tile (a 64-byte load occupies the CU's address path like a 1 KiB one: the output layer's
  // stream ran 15 % faster without them). Lives behind the colsum array (epi_pair_offset_ints).
  const int *cp4 = nullptr;
  float u = 0.f;
  float a_quant = 0.f;
  float b_quant = 0.f;
  int K = 0;
  int N = 0;        // logical columns
  int n_tiles = 0;  // ceil(N / 16)
};

size_t packed_weight_bytes(int K, int N);
// the colsum allocation: [n_tiles * 16] column sums, then (256-byte aligned) the pair constants
__host__ __device__ inline size_t epi_pair_offset_ints(int N) { return ((size_t)((N + 15) / 16) * 16 + 63) / 64 * 64; }
__host__ __device__ inline size_t epi_pair_count(int N) { return (size_t)16 * (((N + 15) / 16 + 31) / 32); }
inline size_t colsum_alloc_bytes(int N) { return (epi_pair_offset_ints(N) + epi_pair_count(N) * 64) * sizeof(int); }

// Diagnostic occupancy trace (tools/occupancy_trace.py): when buf != nullptr,
// thread 0 of every workgroup of the persistent kernels appends a begin and an
// end event {kernel | end << 8 | blockIdx << 16, HW_ID | XCC_ID << 32, 100 MHz
// wall clock} (3 x u64) after the header buf[0] = event counter.
struct OccTrace {
  unsigned long long *buf = nullptr;
  unsigned capacity = 0;  // events
};

// One weight-packing job (load time; once per batch for the shortlisted output
// layer): rows idx[n] (or n) of W [N_src][K] -> MFMA fragment order + colsum + pb.
struct PackArgs {
  const int8_t *W = nullptr;
  int K = 0, N = 0;
  const uint32_t *idx = nullptr;  // nullable
  const float *bias = nullptr;    // nullable, indexed by SOURCE row
  float mult = 0.f;  // (-1 * (127/aq * 127/bq)) / 127, Intgemm.inl.cc:123-126
  void *Wp = nullptr;
  int *colsum = nullptr;
  float *pb = nullptr;
  const uint32_t *n_dev = nullptr;  // nullable: the actual N lives on the device (N is then an upper bound)
  int n_src = 0;  // rows of W (0 = not stated): a gathered index past it reads the last row instead of faulting
};
inline float pack_mult(float a_quant, float b_quant) {
  const float a_alpha = 127.0f / a_quant;
  const float b_alpha = 127.0f / b_quant;
  return (-1.0f * (a_alpha * b_alpha)) / 127.0f;
}

// W: device int8 [N_src][K]; idx (device, nullable) selects/gathers rows;
// bias (device, nullable) is indexed by SOURCE row. Writes Wp/colsum/pb.
hipError_t launch_pack_weight(const int8_t *W, int K, int N, const uint32_t *idx,
                              const float *bias, float a_quant, float b_quant, void *Wp,
                              int *colsum, float *pb, hipStream_t st);

enum GemmEpilogue {
  EPI_PLAIN = 0,   // y = deq (+bias)                        -> f32 [M][ldy]
  EPI_RELU_Q = 1,  // quant(relu(deq), a_quant_out)          -> int8 [M][ldy8]
  EPI_RES_LN = 2,  // LN(deq + res) over the full row        -> f32 [M][ldy]
  EPI_ARGMAX = 3,  // per-row first-max over this block's columns -> partials
  EPI_ACC = 4,     // raw accS                               -> int32 [M][N]
};

struct GemmArgs {
  // A operand: exactly one of x_f32 / x_i8
  const float *x_f32 = nullptr;
  const int8_t *x_i8 = nullptr;
  int lda = 0;  // row stride in elements
  int M = 0;
  // B operand
  PreparedWeight w;
  // outputs
  float *y = nullptr;
  int ldy = 0;
  int8_t *y_i8 = nullptr;
  int ldy8 = 0;
  float a_quant_out = 0.f;
  int32_t *acc_out = nullptr;
  // EPI_PLAIN only: kc_S > 0 stores y in the decoder's K-cache layout
  //   [sentence][head][d/4][key][4]  (row = sentence*kc_S + key, col = head*kc_dh + d)
  // so that a wave reading one head's keys issues fully coalesced 16-byte loads.
  int kc_S = 0, kc_dh = 0;
  // EPI_PLAIN only: store float(accS) (exact) instead of the dequantised value -- the decoder's cross-attention
  // K / V cache, whose consumer applies u and pb after its sums (FusedDecodeArgs::kv24)
  bool raw_acc = false;
  // EPI_RES_LN; EPI_PLAIN: nullable residual added to the output
  const float *res = nullptr;
  int ldres = 0;
  const float *ln_scale = nullptr;
  const float *ln_bias = nullptr;
  float eps = 1e-6f;
  // EPI_ARGMAX: partials [M][n_parts]
  float *part_val = nullptr;
  int *part_idx = nullptr;
  int n_parts = 0;
};

// rows_per_block: 16, 32 or 64 (RM = 1, 2, 4 MFMA row tiles sharing each B
// fragment). Returns the number of column blocks used (EPI_ARGMAX: n_parts).
int gemm_col_blocks(int N, int epilogue, int *nt_out);
hipError_t launch_gemm(const GemmArgs &a, int epilogue, int rows_per_block, hipStream_t st);
// Large-M tiling of the same GEMM (gemm_tile.hip): 128-row blocks, bit-identical results.
// launch_gemm routes EPI_PLAIN / EPI_RELU_Q calls there when gemm_tile_supported().
bool gemm_tile_supported(const GemmArgs &a, int epilogue);
hipError_t launch_gemm_tile(const GemmArgs &a, int epilogue, hipStream_t st);

struct SsruArgs {
  const float *x = nullptr;  // [B][D]
  int B = 0, D = 0;
  PreparedWeight wf, w;      // Wf (affine, bias in pb), W (dot)
  float *state = nullptr;    // [B][D] in/out
  const float *ln_scale = nullptr, *ln_bias = nullptr;
  float eps = 1e-6f;
  float *h = nullptr;        // [B][D]
};
hipError_t launch_ssru(const SsruArgs &a, hipStream_t st);

struct EmbedArgs {
  const int8_t *wemb = nullptr;  // int8 [V][D]
  float inv_mult = 0.f, sqrt_d = 0.f;
  const float *pos = nullptr;    // [max_S][D] sinusoid table
  int D = 0;
  // rows of wemb: an id past the table (only device-resident inputs can carry one: host entry points check) reads the
  // last row instead of faulting -- the result of that sentence is undefined, as in the reference; the others' are not
  int V = 0;
};
__host__ __device__ inline uint32_t embed_row(const EmbedArgs &e, uint32_t tok) {
  return tok < (uint32_t)e.V ? tok : (uint32_t)(e.V - 1);
}
hipError_t launch_embed_encoder(const EmbedArgs &e, const uint32_t *ids, int B, int S, float *x,
                                hipStream_t st);

struct DecodeState {
  // all device pointers
  uint32_t *prev = nullptr;       // [B] last sampled token
  uint32_t *out_ids = nullptr;    // [B][Tmax]
  uint32_t *out_len = nullptr;    // [B]
  uint8_t *finished = nullptr;    // [B]
  int *n_finished = nullptr;      // [1]
  const uint32_t *shortlist = nullptr;  // nullable
  int Tmax = 0;
  uint32_t eos = 0;
  // the output layer's prepared bias of column 0 and its multiplier: a NaN there makes logit[0] NaN for every
  // row, and the reference's scan, which starts from logits[0] and only moves on `value > max`, then stays at
  // class 0 (Transformer.cc:287-298) -- the arg-max kernels skip NaNs, so the rule is applied where a token is taken
  const float *pb0 = nullptr;
  float u_out = 1.0f;
};
// Reduce the argmax partials of the previous step (if first == 0), record the
// tokens (Model.cc:127-137) and build the next decoder input embedding
// (Transformer.cc:133-160). with_embed == 0: sample/record only (last step).
hipError_t launch_decode_begin_step(const EmbedArgs &e, const DecodeState &s, int B, int first,
                                    int with_embed, const float *part_val, const int *part_idx,
                                    int n_parts, float *x, hipStream_t st);
// set prev tokens explicitly (step-wise parity API) and embed
hipError_t launch_embed_decoder(const EmbedArgs &e, const uint32_t *prev, int B, int first,
                                float *x, hipStream_t st);

struct AttnArgs {
  const float *q = nullptr;  // [B*Tq][ldq]
  const float *k = nullptr;  // [B*S][ldk]
  const float *v = nullptr;  // [B*S][ldv]
  int ldq = 0, ldk = 0, ldv = 0;
  const uint32_t *lengths = nullptr;  // [B] -> additive mask 0 / -99999999 (Input.cc:49-63)
  const float *mask = nullptr;        // optional explicit [B][S] mask (op-level API)
  int B = 0, H = 0, Tq = 0, S = 0, dh = 0;
  float alpha = 0.f;  // 1/sqrt(dh)
  float *out = nullptr;  // [B*Tq][ldo] joined heads
  int ldo = 0;
  float *attn = nullptr;  // nullable [B][H][Tq][S]
  // alignment export (decoder, head 0): align[b][out_len[b]][s<len] unless finished
  float *align = nullptr;
  const uint32_t *out_len = nullptr;
  const uint8_t *finished = nullptr;
  int Tmax = 0;
};
hipError_t launch_attention(const AttnArgs &a, hipStream_t st);

// ---- decode-step kernels (decode_kernels.hip) --------------------------------
// 16 f32 rows [B][D]; ln_scale != nullptr => LayerNorm them on load (the
// producer stored the pre-LN sum, Modules.cc:230,254-257,314-316).
struct RowSrc {
  const float *x = nullptr;
  const float *ln_scale = nullptr;
  const float *ln_bias = nullptr;
};

struct DGemmArgs {
  int B = 0, D = 0;
  RowSrc a;                       // A operand from f32 rows (K == D) ...
  const int8_t *a_i8 = nullptr;   // ... or already quantised int8 [B][K]
  PreparedWeight w;
  RowSrc res;                     // residual rows added in the epilogue (N == D)
  float *y = nullptr;
  int ldy = 0;
  int8_t *y_i8 = nullptr;
  int ldy8 = 0;
  float a_quant_out = 0.f;
  float *part_val = nullptr;
  int *part_idx = nullptr;
  int n_parts = 0;
  float eps = 1e-6f;
};
int dgemm_col_blocks(int K, int N, int B);
hipError_t launch_dgemm(const DGemmArgs &a, int epilogue, hipStream_t st);

struct DSsruArgs {
  int B = 0, D = 0;
  RowSrc x;
  PreparedWeight wf, w;
  float *state = nullptr;  // [B][D] in/out
  float *h_pre = nullptr;  // [B][D]: x + relu(c'), pre-LN
  float eps = 1e-6f;
};
hipError_t launch_dssru(const DSsruArgs &a, hipStream_t st);

struct DQAttnArgs {
  int B = 0, D = 0, H = 0, S = 0;
  RowSrc x;            // decoder state rows (pre-LN of the SSRU block)
  PreparedWeight wq;
  const float *k = nullptr;  // cached K, layout [B][H][dh/4][S][4]; values float(accS) (FusedDecodeArgs::kv24)
  const float *v = nullptr;  // cached V [B*S][ldv], likewise
  int ldv = 0;
  // the K / V projections' unquantisation multipliers and prepared biases [D]: applied after the sums
  float uk = 0.f, uv = 0.f;
  const float *pbk = nullptr, *pbv = nullptr;
  const uint32_t *lengths = nullptr;
  float alpha = 0.f, eps = 1e-6f;
  // q_given (nullable [B][D]): the projected queries themselves -- the LayerNorm and the Q projection are skipped
  // (slimt_hip_debug_cross_attention: the attention proper on a caller's query, against the checker's)
  const float *q_given = nullptr;
  // literal: the reference's own float sequence (Modules.cc:24-86 on K, V dequantised element by element,
  // Intgemm.inl.cc:146-153: k = float(accS) u + pb) instead of the hoisted order -- same cache, u and pb applied per value
  bool literal = false;
  int8_t *out_i8 = nullptr;  // joined heads, quantised for the O projection
  float a_quant_out = 0.f;
  float *out_f32 = nullptr;  // optional f32 copy
  float *attn = nullptr;     // nullable [B][H][S]
  float *align = nullptr;
  const uint32_t *out_len = nullptr;
  const uint8_t *finished = nullptr;
  int Tmax = 0;
};
hipError_t launch_dqattn(const DQAttnArgs &a, hipStream_t st);

// ---- merged launches (slimt_hip_translate_many*) -------------------------------
// Several batches of ONE padded source length share an encoder and a decoder launch: what k concurrent Model::forward
// calls of k workers are in the reference (Frontend.cc:207-227; its default batches are 1024 padded tokens,
// Frontend.hh:21-39 -- 32 sentences at S = 32 --, far too few to fill 256 CUs one batch per launch pair). The launch sees
// ONE batch of `B` global sentences; sub-batch j owns the global sentences first[j] .. first[j] + n[j] - 1. When the
// sub-batches have output layers of their own (different shortlists), first[j] is a multiple of the decoder's tile (16,
// or 32 where the 32-sentence tiling may run) -- a tile has ONE output layer --, and the global sentences in between are
// HOLES: the encoder runs them as empty sentences of pad tokens (their cache slots exist; nothing of the caller's is
// read), the decoder never owns them. When all share one layer they follow each other densely (FusedDecodeArgs::sub_dense).
// The K/V cache, its form bytes and every other per-sentence workspace are indexed by the global sentence; only the
// caller's arrays go through these tables, which travel in the kernel arguments (no copy, no helper launch).
// A sentence's arithmetic never depends on its neighbours, so each sub-batch's results are those of its own call.
// A sub-batch may be padded to FEWER tokens than the launch (S_j <= S): its arrays have rows of S_j, its decode loop the
// limit and its alignment rows the width of ITS length (Model.cc:159-161,84-108: both follow the batch's padded length),
// and the positions S_j .. S - 1 of its sentences are padding like any other -- masked keys whose weight is exactly 0
// (Input.cc:49-63), query rows nobody reads.
constexpr int kMaxMerge = 8;
struct MergeIn {  // encoder side, one per sub-batch
  const uint32_t *ids = nullptr;      // [n][S]: rows of S (this sub-batch's own padded length) ids
  const uint32_t *lengths = nullptr;  // [n]
  int first = 0, n = 0;
  int S = 0, pad = 0;
};
struct MergeOut {  // decoder side, one per sub-batch
  const uint32_t *lengths = nullptr;    // [n]
  uint32_t *out_ids = nullptr;          // [n][Tmax]
  uint32_t *out_len = nullptr;          // [n]
  float *align = nullptr;               // nullable [n][Tmax][S] (a staging block when align_out is set)
  float *align_out = nullptr;           // nullable (FusedDecodeArgs::align_out)
  const uint32_t *shortlist = nullptr;  // nullable: this sub-batch's column -> vocabulary id
  int first = 0, n = 0;
  int job = 0;    // which of the launch's packed output layers (sub-batches with one shortlist share one)
  int N = 0;      // its columns
  int S = 0;          // this sub-batch's own padded length: the width of its alignment rows
  int Tmax = 0;       // ... its row length of out_ids / align: max(1, (size_t)(limit_factor * S))
  int max_steps = 0;  // ... and the decode steps its tiles run at most
  int pad = 0;
};
struct MergePack {  // one packing job per DISTINCT shortlist of the launch; job j's buffers lie j strides behind job 0's
  const uint32_t *idx = nullptr;
  int N = 0, pad = 0;
};
// which sub-batch owns global sentence g (first[] ascends; holes belong to the sub-batch in front of them)
template <class T>
__host__ __device__ inline int merge_find(const T *sub, int n_sub, int g) {
  int j = n_sub - 1;
  while (j > 0 && g < sub[j].first) --j;
  return j;
}

// ---- persistent fused decoder (decode_fused.hip) ------------------------------
struct FusedLayerW {
  PreparedWeight rnn_f, rnn_w, q, o, ffn1, ffn2;
  const float *rnn_ln_s = nullptr, *rnn_ln_b = nullptr;
  const float *attn_ln_s = nullptr, *attn_ln_b = nullptr;
  const float *ffn_ln_s = nullptr, *ffn_ln_b = nullptr;
};

struct FusedDecodeArgs {
  int B = 0, S = 0, Ld = 0;
  int max_steps = 0;  // decode steps to run at most (Model.cc:160-161)
  int Tmax = 0;       // row length of out_ids / align
  FusedLayerW L[4];
  PreparedWeight out;                   // (shortlisted) output layer
  const uint32_t *out_n_dev = nullptr;  // nullable: number of output columns, on the device
  const uint32_t *shortlist = nullptr;  // nullable: column -> vocabulary id
  EmbedArgs emb;
  const float *kv = nullptr;            // [Ld][2][B*S*D]: K as [B][H][dh/4][S][4], V as [B*S][D]; values float(accS)
  float *cells = nullptr;               // [Ld][B][D] SSRU cells, used (and zeroed) when D > 256
  const uint32_t *lengths = nullptr;
  float alpha = 0.f, eps = 1e-6f;
  uint32_t eos = 0;
  uint32_t *out_ids = nullptr;  // [B][Tmax]
  uint32_t *out_len = nullptr;  // [B]
  float *align = nullptr;       // nullable [B][Tmax][S]
  // nullable: `align` is then a staging buffer in device memory (rows written as the steps go, nothing
  // else initialised) and each sentence's [Tmax][S] block is written HERE once, when its loop has
  // ended -- zeros outside the recorded rows / beyond the sentence's length -- as 16-byte stores:
  // the form for a destination in pinned host memory (one burst across PCIe per sentence instead of
  // a 4-byte-per-lane store per step).
  float *align_out = nullptr;
  float *attn = nullptr;        // nullable debug [B][H][S]
  unsigned long long *stamps = nullptr;  // nullable diagnostic [64] phase stamps
  int stamp_step = 0;
  int rows_per_wg = 0;  // sentences per workgroup: 0 / 16 = a full row tile, 32 / 8 / 4 where supported (fused_decode_rows)
  // nullable: ticket counter of the over-subscribed launch (16-row kernel). Every workgroup
  // of every launch takes one ticket; ticket - ticket_base is the tile it runs (or none).
  unsigned *ticket = nullptr;
  unsigned ticket_base = 0;
  // XCD-affine placement (home_mask != 0; speed only, never correctness): a launch over-subscribes
  // every XCD with candidates (workgroups are dealt round-robin over the 8 XCDs, whatever their CUs
  // are doing) and a candidate claims a tile only on a HOME XCD of this batch (bit XCC_ID of
  // home_mask) -- so that the batch's own shortlisted output layer (1 MB) is streamed through ONE
  // (or two) 4 MB L2 instead of all eight -- or when fewer candidates are still to arrive than tiles
  // are unclaimed (then anyone takes one: every tile is claimed whatever the dispatcher does).
  // *xstate = {arrivals : 32 | claimed : 32} of all launches so far (never reset); xarr_base /
  // xclaim_base = its halves before this launch; xgrid = candidates of this launch.
  unsigned long long *xstate = nullptr;
  unsigned home_mask = 0;
  unsigned xarr_base = 0, xclaim_base = 0, xgrid = 0;
  // Packed K/V cache (D = 256 / d_head 32: S <= 32 written by encode_fused / encode_tall, 33..64 by
  // encode_tall; D = 512 / d_head 64, S <= 32, by encode_wide): every cached value is the int8 GEMM's accumulator as a 24-bit
  // integer -- the shifted one, accS = acc + 127 colsum, at K = 256 (|accS| < 2^23), the signed
  // one at K = 512 (accS needs 25 bits there; the decoder adds the column's term); 16 values = 48 bytes = one 16-byte quad in each of
  // three planes, so that loads stay 16 bytes per lane and contiguous across lanes:
  //   K [B][D/16][plane][S][16 B]             (16 consecutive columns of one key)
  //   V [B][ceil(S/4)][plane][D/4][16 B]      (4 keys x 4 consecutive columns, key-major)
  // in the same per-layer planes as the f32 form. BOTH forms cache the accumulator, not the dequantised
  // value (the f32 form as float(accS), exact): the decoder's attention applies the projections' u and pb
  // after its sums (decode_fused.hip, unpack24f: the hoisted PORTABLE order of the oracle), the same
  // arithmetic from either form, the packed one from 25 % fewer bytes, load instructions and registers in flight.
  bool kv24 = false;
  // The NARROW form of the packed cache (D = 256, S <= 32): a sentence-layer whose K and V accumulators all lie in
  // [-2^19, 2^19) is cached in 20 bits per value -- accS >> 4 as a signed 16-bit plane, accS & 15 as a nibble plane
  // (decode_attention_packed.inl.h, attention_packed32 with Form20: layouts and the unpack) -- in the slot its 24-bit form would take; any other
  // keeps the 24-bit form, which holds every accumulator K = 256 can produce. kv_fmt[l * B + b] (written by the
  // encoder that filled the cache, FusedEncodeArgs::kv_fmt) says which: 0 = narrow, 1 = 24-bit; nullptr = all 24-bit.
  // Both forms give back the same integers, so results do not depend on it: 17 % fewer K/V bytes per step.
  const unsigned char *kv_fmt = nullptr;
  // kv_fmt may hold 2 = the TIGHT form (D = 256, S <= 32): the accumulator less its column's centre as plain int16
  // (decode_attention_packed.inl.h, attention_packed32 with Form16), for sentence-layers whose K and V all lie in [-2^15, 2^15) then. Set by the engine
  // when this batch's encoder was allowed to write it: the launch then uses the kernels with that form inlined.
  bool kv_tight = false;
  // ... held relative to a per-column CENTRE: the cache has r = accS - kv_centre[l][K, V][d], the reader adds float(centre)
  // back (exact: integers below 2^24). The accumulators of a column scatter around a mean that depends on the column (the
  // encoder output's per-feature mean times the weights: on the synthetic tiny11 the column means' spread is 7.5 k, the
  // scatter around them 3.2 k), so the centres -- the column means of one calibration batch (engine.cpp, kv_calibrate) --
  // are what makes the form fit practically every sentence. Any integers would give the same results.
  const int *kv_centre[4][2] = {};
  float kv_u4096[4][2] = {};      // [layer][K, V]: u / 4096 (narrow form: the integers come back as accS * 4096)
  const float *kv_pb[4][2] = {};  // [layer][K, V]: the projections' prepared biases [D] (both forms)
  const int *kv_cs[4][2] = {};    // [layer][K, V]: their column sums [D] (D = 512: the cache holds the signed accumulator)
  float kv_u[4][2] = {};          // [layer][K, V]: unquantisation multiplier u (f32 form)
  float kv_u256[4][2] = {};       // [layer][K, V]: u / 256 (packed form: the integers come back as accS * 256)
  // Cluster logits (decode_fused.hip, CL; output layers of 16k columns and more): `cluster` consecutive 16-sentence tiles
  // share the output layer, each member computing 1 / cluster of its columns for all of them. cl_act [tiles][16][D] int8:
  // every tile's quantised input rows of the current step; cl_part [tiles][16 cluster + 1][2]: per member and cluster
  // sentence its best (logit bits, column), then {all my sentences ended, 0}; cl_sync [clusters]: arrivals, ZERO at launch.
  // dev_error (nullable, pinned host memory): set non-zero when a bounded wait inside a launch ran out.
  int cluster = 0;
  unsigned char *cl_act = nullptr;
  int *cl_part = nullptr;
  unsigned *cl_sync = nullptr;
  unsigned *dev_error = nullptr;
  // merged launch (MergeOut above): n_sub > 0 -- lengths / out_ids / out_len / align / align_out / shortlist / out above are
  // then NOT used (each tile takes its sub-batch's); the packed output layer of job j: Wp, colsum (+ pair constants), pb of
  // `out` moved by j strides (bytes)
  int n_sub = 0;
  // sub_dense: the sub-batches share ONE output layer (one shortlist for all, or the full vocabulary) and follow each other
  // without holes -- a tile may then hold sentences of several (each wave finds its sentence's own); else every sub-batch
  // starts at a multiple of the tile (a tile has one output layer) and the sentences in between are holes
  int sub_dense = 0;
  MergeOut sub[kMaxMerge];
  size_t out_stride_wp = 0, out_stride_cs = 0, out_stride_pb = 0;
  bool ln_in_lds = false;  // set by the launcher: the LayerNorm constants of all layers fit LDS beside the rest
  bool kv_nt = false;  // non-temporal K/V cache loads (d_head 32 / 64 shapes; see decode_fused.hip)
  // with kv_nt: which caches are still read temporally, in eighths of a layer: sentence b's cache of
  // layer l is kept iff 8 l + (b mod 8) < kv_temporal_eighths (8 = all of layer 0, 12 = layer 0 and
  // half the sentences' layer 1, ...)
  int kv_temporal_eighths = 0;
  OccTrace trace;
};
// hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), but only when
// `bytes` exceeds what was already set for this kernel on the current device: the call takes
// runtime-wide locks, and a host pipeline has a dozen threads launching at once.
hipError_t set_dynamic_lds_once(const void *kernel, int bytes);
int fused_decode_grid(int B, bool tickets, int rows);
int fused_encode_grid(int B, int S, bool tickets);
bool fused_decode_supported(int D, int F, int H, int Ld);
bool fused_decode_mid_supported(int D, int F, int H, int Ld);
bool fused_decode_tight_supported(int D, int F, int H, int Ld);
bool fused_decode_tight_rows32_supported(int D, int F, int H, int Ld);
bool fused_decode_tight_mid_supported(int D, int F, int H, int Ld, int mid);
bool fused_decode_long24_supported(int D, int F, int H, int Ld);
int fused_decode_rows(int D, int F, int H, int Ld, int S, int B, int forced, bool kv24);
hipError_t launch_decode_fused(const FusedDecodeArgs &a, int D, int F, int H, hipStream_t st);

// ---- lexical shortlist generation (shortlist.hip) -------------------------------
struct ShortlistArgs {
  const unsigned long long *w2o = nullptr;  // [src_vocab + 1] offsets into lists
  const uint32_t *lists = nullptr;          // sorted target ids per source word
  unsigned long long frequent = 0;
  int shared = 0;
  int src_vocab = 0, tgt_vocab = 0;
  const uint32_t *ids = nullptr;      // [B][S] padded source tokens
  const uint32_t *lengths = nullptr;  // [B]
  int B = 0, S = 0;
  uint32_t *out = nullptr;    // [tgt_vocab] capacity
  uint32_t *n_out = nullptr;  // [1]
  uint32_t *n_out_host = nullptr;  // nullable: a second copy of the count (pinned host memory: the host's hint for its next launch)
  uint32_t *scratch = nullptr;  // target + source bitmaps, zero on entry and on exit
};
size_t shortlist_lds_bytes(int src_vocab, int tgt_vocab);
// LDS of the one-workgroup form (both bitmaps + the scan array), and whether an encoder launch can host it
inline size_t shortlist_in_launch_lds_bytes(int src_vocab, int tgt_vocab) {
  return ((size_t)(tgt_vocab + 31) / 32 + (size_t)(src_vocab + 31) / 32 + 1024) * sizeof(uint32_t);
}
size_t shortlist_scratch_bytes(int src_vocab, int tgt_vocab);
hipError_t launch_shortlist_generate(const ShortlistArgs &a, hipStream_t st);

// ---- persistent fused encoder (encode_fused.hip) ------------------------------
struct FusedEncLayerW {
  PreparedWeight q, k, v, o, ffn1, ffn2;
  const float *attn_ln_s = nullptr, *attn_ln_b = nullptr;
  const float *ffn_ln_s = nullptr, *ffn_ln_b = nullptr;
};

struct FusedEncodeArgs {
  int B = 0, S = 0, Le = 0, Ld = 0;
  FusedEncLayerW L[6];
  PreparedWeight dec_k[4], dec_v[4];  // decoder cross-attention K / V projections
  EmbedArgs emb;
  const uint32_t *ids = nullptr;      // [B][S]
  const uint32_t *lengths = nullptr;  // [B]
  float alpha = 0.f, eps = 1e-6f;
  float *kv = nullptr;         // [Ld][2][B*S*D]: K as [B][H][dh/4][S][4], V as [B*S][D]
  bool kv24 = false;           // write the packed 24-bit form instead (FusedDecodeArgs::kv24)
  bool kv_store_nt = false;    // 64-row encoder, packed form: non-temporal cache stores (the decoder of this batch will stream it)
  // packed form, S <= 32, D = 256: [Ld][B] bytes; the encoder writes a sentence-layer's cache in the narrow 20-bit form
  // when every K and V accumulator of its workgroup's rows fits (FusedDecodeArgs::kv_fmt) and records 0, else the
  // 24-bit form and 1. nullptr: always the 24-bit form. kv_narrow_limit: accumulators must lie in
  // [-limit, limit) for the narrow form -- 2^19, what 20 bits hold; tests lower it to force the 24-bit form on
  // some sentences of a batch (never above 2^19)
  unsigned char *kv_fmt = nullptr;
  int kv_narrow_limit = 1 << 19;
  // the tight 16-bit form (FusedDecodeArgs::kv_tight) is tried first in the decoder layers of kv_tight_layers (bit l):
  // centred accumulators must lie in [-kv_tight_limit, kv_tight_limit) (2^15, what int16 holds; tests lower it); recorded as 2
  int kv_tight_limit = 0;
  unsigned kv_tight_layers = 0;
  const int *kv_centre[4][2] = {};  // [layer][K, V][D]: the tight form holds accS - centre (FusedDecodeArgs::kv_centre)
  unsigned long long *kv_not16_count = nullptr;  // [Ld], like kv_wide_count: + 1 per sentence of layer l that tried and did not take the tight form
  // nullable, pinned host memory: + 1 per sentence-layer that took the 24-bit form (the engine watches the share: a model
  // whose accumulators mostly do not fit 20 bits is switched to the 24-bit form altogether, engine.cpp)
  unsigned long long *kv_wide_count = nullptr;
  PackArgs pack;               // the batch's shortlisted output layer, packed by the
  int pack_tiles = 0;          // encoder's workgroups on the side (0 = nothing to pack)
  // merged launch (MergeIn above): n_sub > 0 -- ids / lengths above are not used. n_pack > 1: `pack` describes job 0 and
  // job j differs by its idx / N and by j strides (bytes) on Wp / colsum / pb; pack_tiles = tiles of the LARGEST job
  int n_sub = 0;
  MergeIn sub[kMaxMerge];
  int n_pack = 0;
  MergePack pjob[kMaxMerge];
  size_t pack_stride_wp = 0, pack_stride_cs = 0, pack_stride_pb = 0;
  // ShortlistGenerator::generate inside this launch (the S <= 64 encoders; gen.w2o != nullptr): the workgroup
  // that claims tile 0 generates the batch's shortlist (gen.out / gen.n_out = pack.idx / pack.n_dev) before its
  // encoder work and publishes *gen_flag = gen_epoch; every workgroup packs its share of the output layer at
  // the END of its encoder work, behind that flag (the publisher is running before any waiter can start:
  // tiles are claimed in start order, and it waits for nobody).
  ShortlistArgs gen;
  unsigned *gen_flag = nullptr;
  unsigned gen_epoch = 0;
  unsigned gen_spin_limit = 1u << 24;  // polls a waiter makes before it gives up (tests shorten it)
  unsigned gen_wait_xor = 0;           // debug: the waiters look for gen_epoch ^ this -- non-zero = a publisher that never comes
  unsigned *dev_error = nullptr;       // nullable, pinned host memory: set to 1 when a waiter gave up (FusedDecodeArgs::dev_error)
  unsigned *ticket = nullptr;  // nullable: over-subscribed launch (see FusedDecodeArgs)
  unsigned ticket_base = 0;
  OccTrace trace;
  float *enc_out = nullptr;    // nullable [B*S][D]
  float *layer_out = nullptr;  // nullable [Le][B*S][D]
  float *embed_out = nullptr;  // nullable [B*S][D]
  unsigned long long *stamps = nullptr;  // nullable diagnostic phase stamps
  int stamp_layer = 0;
};
bool fused_encode_supported(int D, int F, int H, int Le, int Ld, int S);
hipError_t launch_encode_fused(const FusedEncodeArgs &a, int D, int F, int H, hipStream_t st);
// D = 512 / F = 2048 / 8 heads (encode_wide.hip): same arguments, reached through launch_encode_fused
// 64-row tiles for D = 256 (encode_tall.hip): half the weight bytes per source token
bool tall_encode_supported(int D, int F, int H, int Le, int Ld, int S);
int tall_encode_grid(int B, int S, bool tickets);
hipError_t launch_encode_tall(const FusedEncodeArgs &a, int F, hipStream_t st);
bool wide_encode_supported(int D, int F, int H, int Le, int Ld, int S);
hipError_t launch_encode_wide(const FusedEncodeArgs &a, hipStream_t st);

// rows of 64 / 128 / 256 / 512 columns; y8 (nullable): int8 copy quantised with aq8
hipError_t launch_layer_norm_q(const float *x, const float *scale, const float *bias, float eps,
                               int rows, int cols, float *y, int8_t *y8, float aq8,
                               hipStream_t st);
hipError_t launch_layer_norm(const float *x, const float *scale, const float *bias, float eps,
                             int rows, int cols, float *y, hipStream_t st);
hipError_t launch_softmax(const float *x, int rows, int cols, float *y, hipStream_t st);
hipError_t launch_highway(const float *x, const float *y, const float *g, size_t n, float *out,
                          hipStream_t st);
// [B,H,T,dh] <-> [B,T,H*dh] helpers for the op-level sdpa API
hipError_t launch_transpose_heads(const float *in, int B, int d2, int d1, int d0, float *out,
                                  hipStream_t st);

// ---- persistent per-sentence encoder for 32 < S <= 128 (kernels.hip) -----------
// One workgroup owns one sentence for the whole encoder: the stages of the
// layer-by-layer path (same tile code, same numbers) run back to back inside
// one launch with workgroup barriers in between; the f32 / int8 intermediates
// go through the context's global scratch (L2-resident per sentence).
struct LongEncodeArgs {
  FusedEncodeArgs f;  // sizes, layer weights, decoder K/V projections, embedding, inputs, outputs, pack job
  int H = 0, D = 0, F = 0;
  float *x = nullptr, *y = nullptr;                    // [B*S][D] residual stream (ping-pong)
  float *q = nullptr, *k = nullptr, *v = nullptr;      // [B*S][D]
  float *att = nullptr;                                // [B*S][D]
  int8_t *h8 = nullptr;                                // [B*S][F]
};
bool long_encode_supported(int D, int F, int H, int Le, int Ld, int S);
hipError_t launch_encode_long(const LongEncodeArgs &a, hipStream_t st);
// the column means (rounded) of an f32 K/V cache [Ld][K, V][B * S][D]: centre[(2 l + p) * D + d] (FusedDecodeArgs::kv_centre)
hipError_t launch_kv_centres(const float *kv, int Ld, int B, int S, int D, unsigned long long *sums, int *centre, hipStream_t st);

}  // namespace slimt_hip
