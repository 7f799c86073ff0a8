// Persistent fused encoder for D = 512 ("base": F = 2048, 8 heads of 64): embedding + all
// encoder layers + the decoder's cross-attention K/V cache in ONE launch, 32 rows
// (= floor(32 / S) whole sentences, S <= 32) per workgroup -- what encode_fused.hip does for
// D = 256, re-planned because at D = 512 nothing fits the same way in 160 KiB of LDS:
//
//   * the residual stream (32 x 512 f32 = 64 KiB) lives in REGISTERS: wave w owns rows 2 w and
//     2 w + 1, lane L holds columns L + 64 i (the element-to-lane map of the canonical row sum),
//     16 registers. LayerNorm, the residual adds, the embedding and every quantisation of x run
//     on the owner from registers;
//   * three int8 A-operand buffers (32 x 512 B each: x quantised once per layer with the
//     multipliers of Q, K and V), unpadded with XOR-swizzled 16-byte chunks -- padded they would
//     not fit --, so a round's three projections run back to back with no barrier in between;
//   * heads are staged in two rounds of four: q / k / v of four heads in f32 (97 KiB), their
//     attention on the f32 matrix cores (v_mfma_f32_32x32x2_f32 chains over ascending k, bit
//     identical to the ascending fmaf chain; 32 + 2 x 16 MFMAs per (sentence, head)), output
//     quantised into the O projection's A operand (round 0: a narrow buffer of its own, round 1: Q's);
//   * GEMM outputs that meet the residual (O projection, FFN2) cross from the column-tile
//     owner to the row owner through an f32 exchange tile that time-shares the q / k / v
//     region, as does the FFN's hidden layer (32 x 2048 int8): FFN2's accumulators stay in
//     registers until every wave has read the hidden layer, then become the exchange tile.
//
// Weights stream from L2 in MFMA-fragment order with the operands swapped (weights as A):
// an accumulator lane holds 4 consecutive columns of one row, so outputs leave as 16-byte /
// 4-byte pieces and epilogue constants arrive as 16-byte loads. Arithmetic is bit-identical
// to the layer-by-layer kernels (kernels.hip) and to the oracle's portable order.
// Reference: Model.cc:195-201, Transformer.cc:57-69, Modules.cc:287-334, TensorOps.cc:542-580.
#include "device_common.h"
#include "shortlist_device.h"
#include "kernels.h"

namespace slimt_hip {

namespace {

constexpr int WNW = 16;  // waves per workgroup
constexpr int WR = 32;   // rows per workgroup

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t wrsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ v4i wload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Four columns at once, two per packed instruction (v_pk_mul_f32 / v_pk_add_f32: the same IEEE
// operations as the scalar forms, multiply and add stay separate roundings).
typedef float wf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 wdequant4(const v4i &c, const int (&cs)[4], float u, const float (&pb)[4]) {
  wf2 lo = {(float)(c[0] + __mul24(127, cs[0])), (float)(c[1] + __mul24(127, cs[1]))};
  wf2 hi = {(float)(c[2] + __mul24(127, cs[2])), (float)(c[3] + __mul24(127, cs[3]))};
  const wf2 uu = {u, u};
  lo = lo * uu;
  hi = hi * uu;
  const wf2 pl = {pb[0], pb[1]}, ph = {pb[2], pb[3]};
  lo = lo + pl;
  hi = hi + ph;
  float4 v;
  v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
  return v;
}
// relu, then PrepareA with `aq`, four int8 in one register (see encode_tall.hip, trelu_quant4: for a
// positive multiplier the relu rides in the clamp; v_cvt_pk_u8_f32 rounds to nearest even, saturates at 0 and
// packs; the upper clamp 127 is taken on the four packed bytes)
__device__ __forceinline__ int wrelu_quant4(const float4 &v, float aq) {
  wf2 lo = {v.x, v.y}, hi = {v.z, v.w};
  const wf2 qq = {aq, aq};
  lo = lo * qq;
  hi = hi * qq;
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_u8_f32(lo.x, 0, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(lo.y, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(hi.x, 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(hi.y, 3, w);
  const unsigned m = w & 0x80808080u;
  return (int)((w | (m - (m >> 7))) & 0x7f7f7f7fu);
}

// epilogue constants of column tile `tile` for this lane's 4 columns (4 lg .. 4 lg + 3)
struct Epi4 {
  int cs[4];
  float pb[4];
};
__device__ __forceinline__ Epi4 load_epi4(const PreparedWeight &w, int tile, int lg) {
  const rsrc_t rc = wrsrc(w.colsum, (unsigned)w.n_tiles * 64u), rp = wrsrc(w.pb, (unsigned)w.n_tiles * 64u);
  const v4i c = wload(rc, lg * 16, tile * 64);
  const float4 p = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp, lg * 16, tile * 64, 0));
  Epi4 e;
  e.cs[0] = c[0]; e.cs[1] = c[1]; e.cs[2] = c[2]; e.cs[3] = c[3];
  e.pb[0] = p.x; e.pb[1] = p.y; e.pb[2] = p.z; e.pb[3] = p.w;
  return e;
}

// four 24-bit two's-complement integers, little endian, in 12 bytes (the packed K/V cache)
typedef int wv3i __attribute__((ext_vector_type(3)));
__device__ __forceinline__ wv3i wpack24(v4i x) {
  wv3i o;
  o.x = (x.x & 0xffffff) | (x.y << 24);
  o.y = ((x.y >> 8) & 0xffff) | (x.z << 16);
  o.z = ((x.z >> 16) & 0xff) | (x.w << 8);
  return o;
}

// canonical LayerNorm of one row held in registers (v[i] = column lane + 64 i), in place
template <int DPL>
__device__ __forceinline__ void ln_regs(float (&v)[DPL], const float (&scale)[DPL], const float (&bias)[DPL],
                                        float eps) {
  constexpr int D = 64 * DPL;
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
  float tq[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) tq[i] = v[i] - mean;
  SharedDiv(sigma, SLIMT_DIV_LN_D).quot<DPL, false>(tq, SLIMT_DIV_LN_N);  // (v - mean) / sigma, correctly rounded
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    const float t = tq[i];
    const float m = scale[i] * t;
    v[i] = m + bias[i];
  }
}

template <int DPL>
__device__ __forceinline__ void load_ln_regs(const float *scale, const float *bias, int lane, float (&sc)[DPL],
                                             float (&bi)[DPL]) {
  const rsrc_t rs = wrsrc(scale, 64u * DPL * 4u), rb = wrsrc(bias, 64u * DPL * 4u);
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    sc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, i * 256, 0));
    bi[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, i * 256, 0));
  }
}

}  // namespace

// see decode_fused.hip: keeps lane-derived offsets from being hoisted and spilled
// Diagnostic phase stamps (100 MHz wall clock) of workgroup 0 in one layer (slots 0..10).
#define SLIMT_WSTAMP(id)                                                              \
  do {                                                                                \
    if (a.stamps && s0 == 0 && tid == 0 && l == a.stamp_layer)                        \
      a.stamps[(id)] = wall_clock64();                                                \
  } while (0)

// Loop conditions of the phase that runs once per batch (the decoder's K/V cache, behind the layers). The register allocator weighs a
// value by the STATIC frequency of the blocks that use it -- loop depth, 32 iterations assumed per level --, so the phase's four-deep
// loops outweighed the encoder layers (depth 1-2) and the layers' values were the ones spilled: 58 spilled registers, 10 scratch
// stores + 21 loads inside a layer. With its loops marked unlikely the layers keep the 12 they had without the phase (the residual
// stream parked across the FFN) and the phase takes the spills (profiles/r06_encode_wide_cold_phase.txt).
#define SLIMT_ONCE_PER_BATCH(c) __builtin_expect(!!(c), 0)

#define SLIMT_WPHASE_LANE                               \
  int lane = lane0;                                     \
  asm volatile("" : "+v"(lane));                        \
  const int lr = lane & 15, lg = lane >> 4;             \
  (void)lr;                                             \
  (void)lg

template <int KSD, int KSF, int DH>
__global__ __launch_bounds__(1024) void encode_wide_kernel(FusedEncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 64 * KSD, F = 64 * KSF, H = D / DH;
  constexpr int HR = 4;              // heads per round
  constexpr int RC = HR * DH;        // q / k / v columns per round
  constexpr int NR = H / HR;         // rounds
  // int8 A rows: unpadded, the 16-byte chunks of a row XOR-swizzled with the row (a fragment read
  // of 16 rows x one chunk then covers all 64 banks without pad bytes): THREE A buffers -- x
  // quantised once per layer with the multipliers of Q, K and V -- fit beside the rest
  constexpr int LDA = D;
  constexpr int LDO = RC + 16;       // int8 attention output rows of one round
  // f32 q / k / v rows: the attention's 16x16x4 operands are read by lanes (row n = lane % 16,
  // k index g = lane / 16) -- q, k at [row n][d + g]: stride = 4 mod 64 words is conflict-free;
  // v at [key g][d + n]: stride = 16 mod 64 is
  constexpr int LDQ = RC + 4;
  constexpr int LDV = RC + 16;
  constexpr int LDY = D + 4;         // f32 exchange rows
  constexpr int LDH = F + 32;        // int8 hidden rows (+ 32: an FFN2 fragment read then covers all 64 banks, encode_tall.hip)
  static_assert(RC / 16 == WNW, "one 16-column tile of a round's projection per wave");
  static_assert(D / 16 == 2 * WNW, "two 16-column tiles of a D-wide GEMM per wave");
  static_assert(DH == 64 && KSD == 8, "planned for D = 512, d_head = 64");
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, B = a.B;
  const int spw = WR / S;  // whole sentences per workgroup

  __shared__ int claimed;
  const int n_wg = (B + spw - 1) / spw;
  int tile = blockIdx.x;
  if (a.ticket) {  // over-subscribed launch: the first workgroups to START take the tiles
    if (tid == 0) claimed = (int)(atomicAdd(a.ticket, 1u) - a.ticket_base);
    __syncthreads();
    tile = claimed;
    if ((unsigned)tile >= (unsigned)n_wg) return;
  }
  const int s0 = tile * spw;  // first sentence
  const int rows_used = spw * S;
  if (tid == 0) occ_trace_event(a.trace, 2, 0);
  // this workgroup's sentence lengths: read once (they may live in pinned host memory)
  __shared__ int slens[WR];
  __shared__ int kv_wide_flag;  // the narrow cache form does not hold this workgroup's accumulators (the K/V phase at the end)
  if (tid < spw) slens[tid] = s0 + tid < B ? sentence_length(a, s0 + tid, S) : 0;

  char *Abuf = smem;                 // x quantised for Q | round 1's attention output | FFN1 / decoder K/V input
  char *Akb = Abuf + WR * LDA;       // x quantised for K
  char *Avb = Akb + WR * LDA;        // x quantised for V
  char *Ob0 = Avb + WR * LDA;        // [WR][LDO] attention output of round 0
  char *Ob1 = Abuf;                  // ... of round 1: Q's operand is dead by then
  char *region = Ob0 + WR * LDO;     // q, k, v of four heads | exchange tile | hidden layer
  static_assert(WR * LDO <= WR * LDA, "round 1's attention output fits Q's operand buffer");
  float *qb = reinterpret_cast<float *>(region);
  float *kb = qb + WR * LDQ;
  float *vb = kb + WR * LDQ;  // rows of LDV floats
  float *Yb = reinterpret_cast<float *>(region);
  char *Hb = region;

  auto row_sentence = [&](int r) __attribute__((always_inline)) { return s0 + r / S; };
  auto row_valid = [&](int r) __attribute__((always_inline)) { return r < rows_used && row_sentence(r) < B; };

  // side job: the batch's shortlisted output layer (used by the decoder launch behind this one)
  const bool gen_here = a.gen.w2o != nullptr;  // the batch's shortlist is generated in this launch (encode_tall.hip): packed at the end
  if (!gen_here) {
    pack_weight_share(a, tile, n_wg, tid, 1024);
  } else {
    // ShortlistGenerator::generate (Shortlist.cc:115-175; Model.cc:117-120) by the workgroup that started first (a merged
    // launch: by the first n, one shortlist per sub-batch), in the still unused LDS, then published for the others (shortlist_device.h)
    shortlists_publish_in_launch(a, reinterpret_cast<uint32_t *>(smem), tile, n_wg, tid);
  }

  // ---- embedding (Model.cc:195-197) into the owner's registers ------------------------------
  float x[2][KSD];
  {
    SLIMT_WPHASE_LANE;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;
      const bool ok = row_valid(r);
      const int sb = row_sentence(r), pos = r % S;
      const SentenceIds sids = ok ? sentence_ids(a, sb, S) : SentenceIds{nullptr, 0};
      const uint32_t tok = pos < sids.n ? embed_row(a.emb, sids.p[pos]) : 0;
#pragma unroll
      for (int i = 0; i < KSD; ++i) {
        float v = 0.0f;
        if (ok) {
          const float e = (float)a.emb.wemb[(size_t)tok * D + lane + 64 * i] * a.emb.inv_mult;
          const float sc = e * a.emb.sqrt_d;
          v = sc + a.emb.pos[(size_t)pos * D + lane + 64 * i];
        }
        x[rr][i] = v;
        if (ok && a.embed_out) a.embed_out[((size_t)sb * S + pos) * D + lane + 64 * i] = v;
      }
    }
  }
  // the owner's rows, quantised for the next affine, into the A buffer
  auto quantise_x = [&](char *A, float aq, int lane) __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int i = 0; i < KSD; ++i) {
        const int row = 2 * wave + rr, col = lane + 64 * i;
        A[row * LDA + ((((col >> 4) ^ row) & 15) << 4 | (col & ~255)) + (col & 15)] = (char)quantize1_byte(x[rr][i], aq);
      }
  };
  // Weight fragments are requested one phase ahead of their use, across the barriers and the
  // LayerNorm / attention phases in between: bw[0] and bw[1] hold one 16-column tile (K = D) each,
  // or -- in FFN2 -- one chunk of 4 k-steps of the wave's two column tiles.
  v4i bw[2][KSD];
  v4i ecs[2];    // FFN1: colsum / prepared bias of the tile in bw[buf], requested with it (loads
  float4 epb[2]; // return in order: constants asked for later would wait behind the next tile)
  auto load_w = [&](v4i (&f)[KSD], const PreparedWeight &w, int ct, int lane) __attribute__((always_inline)) {
    const rsrc_t rw = wrsrc(w.Wp, (unsigned)w.n_tiles * KSD * 1024u);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) f[ks] = wload(rw, lane * 16, (ct * KSD + ks) * 1024);
  };
  // one 16-column tile against the 32 rows of `A`: accumulator lane = rows lr / 16 + lr, columns
  // 4 lg .. 4 lg + 3 of the tile (weights as the MFMA A operand)
  auto mma = [&](const char *A, const v4i (&f)[KSD], int lane, v4i &c0, v4i &c1) __attribute__((always_inline)) {
    const int lr = lane & 15, lg = lane >> 4;
    c0 = v4i{0, 0, 0, 0};
    c1 = v4i{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      const int ch = ks * 4 + lg;  // 16-byte chunk of the row, swizzled
      const int off = ((ch & 16) | ((ch ^ lr) & 15)) << 4;
      const v4i a0 = *reinterpret_cast<const v4i *>(A + lr * LDA + off);
      const v4i a1 = *reinterpret_cast<const v4i *>(A + (16 + lr) * LDA + off);
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks], a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks], a1, c1, 0, 0, 0);
    }
  };
  // the same against the attention output (k-steps 0..3: round 0's heads, 4..7: round 1's; padded rows)
  auto mma_o = [&](const v4i (&f)[KSD], int lane, v4i &c0, v4i &c1) __attribute__((always_inline)) {
    const int lr = lane & 15, lg = lane >> 4;
    c0 = v4i{0, 0, 0, 0};
    c1 = v4i{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      const char *O = ks < KSD / 2 ? Ob0 : Ob1;
      const v4i a0 = *reinterpret_cast<const v4i *>(O + lr * LDO + (ks % (KSD / 2)) * 64 + lg * 16);
      const v4i a1 = *reinterpret_cast<const v4i *>(O + (16 + lr) * LDO + (ks % (KSD / 2)) * 64 + lg * 16);
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks], a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks], a1, c1, 0, 0, 0);
    }
  };
  auto dequant4 = [&](const v4i &c, const Epi4 &e, float u) __attribute__((always_inline)) { return wdequant4(c, e.cs, u, e.pb); };
  {
    SLIMT_WPHASE_LANE;
    load_w(bw[0], a.L[0].q, wave, lane);
  }

  for (int l = 0; l < a.Le; ++l) {
    const FusedEncLayerW &L = a.L[l];
    SLIMT_WSTAMP(0);
    // ---- Attention::forward (Modules.cc:287-319), four heads per round -----------------------
    {  // x quantised with the three projections' multipliers, once per layer
      SLIMT_WPHASE_LANE;
      // (no barrier: the A buffers were last read by FFN1 and round 1's projections, barriers ago; the region -- still the
      // exchange tile the LayerNorm before read -- is not written until the barrier below; encode_tall.hip)
      quantise_x(Abuf, L.q.a_quant, lane);
      quantise_x(Akb, L.k.a_quant, lane);
      quantise_x(Avb, L.v.a_quant, lane);
    }
    for (int hr = 0; hr < NR; ++hr) {
      // Q, K, V projections of this round's heads: wave = column tile, one after the other (their
      // outputs are disjoint: no barrier in between)
      lds_barrier();  // round 0: the operands are complete; round 1: round 0's attention has read q / k / v
      if (hr == 1) SLIMT_WSTAMP(2);
      for (int p = 0; p < 3; ++p) {
        SLIMT_WPHASE_LANE;
        const PreparedWeight &w = p == 0 ? L.q : (p == 1 ? L.k : L.v);
        const int ct = hr * WNW + wave;
        const Epi4 e = load_epi4(w, ct, lg);
        v4i c0, c1;
        mma(p == 0 ? Abuf : (p == 1 ? Akb : Avb), bw[0], lane, c0, c1);
        __builtin_amdgcn_sched_barrier(0);
        if (p < 2)
          load_w(bw[0], p == 0 ? L.k : L.v, ct, lane);
        else if (hr + 1 < NR)
          load_w(bw[0], L.q, ct + WNW, lane);
        else
          load_w(bw[0], L.o, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        const int col = wave * 16 + lg * 4;  // column inside the round
        float *dst = p == 0 ? qb : (p == 1 ? kb : vb);
        const int ldd = p == 2 ? LDV : LDQ;
        *reinterpret_cast<float4 *>(dst + lr * ldd + col) = dequant4(c0, e, w.u);
        *reinterpret_cast<float4 *>(dst + (16 + lr) * ldd + col) = dequant4(c1, e, w.u);
      }
      lds_barrier();
      SLIMT_WSTAMP(hr == 0 ? 1 : 3);
      // scaled_dot_product_attention (Modules.cc:24-86) on the f32 matrix cores: one wave per
      // (sentence, head of the round, 16 queries) -- 8 jobs for one 32-token sentence, on waves 0 .. 7 = two per SIMD
      // (a job per half of the head's columns filled all 16 waves, but the two halves' waves computed the same
      // scores and the same softmax, four to a SIMD: 6.7 -> 5.1 us per round without the copy). v_mfma_f32_16x16x4_f32 chains over ascending k (bit-identical to the
      // ascending fmaf chain, tools/probe_mfma_f32.py); operand maps, butterfly order and the
      // lane-group transpose of P as in encode_fused.hip.
      {
        SLIMT_WPHASE_LANE;
        typedef float v4f __attribute__((ext_vector_type(4)));
        const int n = lane & 15, g = lane >> 4;
        const float minus_inf = -99999999.0f;  // Input.cc:56-61
        const float lowest = -3.402823466e+38f;
        const int nqh = S > 16 ? 2 : 1;  // 16-query halves of a sentence
        for (int job = wave; job < spw * HR * nqh; job += WNW) {
          const int qh = job % nqh, hl = (job / nqh) % HR, sl = job / (nqh * HR);
          const int sb = s0 + sl;
          if (sb >= B) continue;
          const int base = sl * S;
          const int len = slens[sl];
          const int qr = 16 * qh + n;
          const float *qp = qb + (base + (qr < S ? qr : S - 1)) * LDQ + hl * DH + g;
          float sc[2][4];
          {
            // operands first, chains second: fetched where they are used, every pair of MFMAs waited out an LDS round trip
            // (ds_read2, s_waitcnt lgkmcnt(0), two v_mfma: eight times per key tile), and the two key tiles' chains ran one
            // after the other. Both tiles' chains are independent: interleaved, one's latency hides behind the other's issue.
            const float *kp0 = kb + (base + (n < S ? n : S - 1)) * LDQ + hl * DH + g;
            const float *kp1 = kb + (base + (16 + n < S ? 16 + n : S - 1)) * LDQ + hl * DH + g;
            v4f st0 = {0.0f, 0.0f, 0.0f, 0.0f}, st1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int half = 0; half < 2; ++half) {  // (all 48 operands at once cost the phases around this one 30 spilled registers)
              float qv[DH / 8], kv0[DH / 8], kv1[DH / 8];
#pragma unroll
              for (int i = 0; i < DH / 8; ++i) {
                qv[i] = qp[4 * (half * (DH / 8) + i)];
                kv0[i] = kp0[4 * (half * (DH / 8) + i)];
                kv1[i] = kp1[4 * (half * (DH / 8) + i)];
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i = 0; i < DH / 8; ++i) {
                st0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kv0[i], qv[i], st0, 0, 0, 0);
                st1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kv1[i], qv[i], st1, 0, 0, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int m = 16 * kt + 4 * g + r;  // key of this register
                float v = kt ? st1[r] : st0[r];
                v = a.alpha * v;  // (alpha == 1: the product is v, bit for bit)
                v = v + (1.0f - (m < len ? 1.0f : 0.0f)) * minus_inf;
                if (m >= S) v = lowest;
                sc[kt][r] = v;
              }
          }
          float mx = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])),
                           fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
          mx = bf_max<32>(bf_max<16>(mx));
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // (unconditional: a select, not eight exec-mask branches -- encode_tall.hip)
              const float e = exp_p_select(sc[kt][r] - mx);
              sc[kt][r] = (16 * kt + 4 * g + r) < S ? e : 0.0f;
            }
          float t[2];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
            t[kt] = bf_add<32>(bf_add<16>((sc[kt][0] + sc[kt][1]) + (sc[kt][2] + sc[kt][3])));  // masks 1, 2 | 4 | 8
          const float sum = t[0] + t[1];                                                          // mask 16
          float pa[2][4];  // pa[kt][j] on lane (n, g) = P[query n][key 16 kt + 4 j + g]
          {  // e / sum for this lane's eight keys: one refined reciprocal (device_common.h, SharedDiv)
            const SharedDiv dv(sum, SLIMT_DIV_SM_D);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) dv.quot<4, true>(sc[kt], SLIMT_DIV_SM_N);  // keys >= S: exactly 0
          }
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            const slimt_u2 s01 = __builtin_amdgcn_permlane16_swap(__float_as_int(sc[kt][0]), __float_as_int(sc[kt][1]), false, false);
            const slimt_u2 s23 = __builtin_amdgcn_permlane16_swap(__float_as_int(sc[kt][2]), __float_as_int(sc[kt][3]), false, false);
            const slimt_u2 ac = __builtin_amdgcn_permlane32_swap(s01.x, s23.x, false, false);
            const slimt_u2 bd = __builtin_amdgcn_permlane32_swap(s01.y, s23.y, false, false);
            pa[kt][0] = __int_as_float(ac.x);
            pa[kt][1] = __int_as_float(bd.x);
            pa[kt][2] = __int_as_float(ac.y);
            pa[kt][3] = __int_as_float(bd.y);
          }
#pragma unroll
          for (int np = 0; np < DH / 32; ++np) {  // two column tiles at a time: their V operands first, their chains interleaved
            const int dcol = hl * DH + 32 * np + n;  // column inside the round (+ 16: the second tile)
            float v0[8], v1[8];
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {  // keys >= S contribute fma(0, v, o) == o
              const int key = 4 * s4 + g;
              const float *vr = vb + (base + (key < S ? key : S - 1)) * LDV + dcol;
              v0[s4] = vr[0];
              v1[s4] = vr[16];
            }
            __builtin_amdgcn_sched_barrier(0);
            v4f o0 = {0.0f, 0.0f, 0.0f, 0.0f}, o1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
              o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s4 >> 2][s4 & 3], v0[s4], o0, 0, 0, 0);
              o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s4 >> 2][s4 & 3], v1[s4], o1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int q = 16 * qh + 4 * g + r;  // query of this register
              if (q < S) {
                char *orow = (hr == 0 ? Ob0 : Ob1) + (base + q) * LDO + dcol;
                orow[0] = (char)quantize1_byte(o0[r], L.o.a_quant);
                orow[16] = (char)quantize1_byte(o1[r], L.o.a_quant);
              }
            }
          }
        }
        // rows that belong to no sentence keep a defined A operand
        for (int r = rows_used + wave; r < WR; r += WNW)
#pragma unroll
          for (int i = 0; i < RC / 64; ++i) (hr == 0 ? Ob0 : Ob1)[r * LDO + lane + 64 * i] = 0;
      }
    }
    lds_barrier();  // attention of the last round is complete: q / k / v are dead
    SLIMT_WSTAMP(4);
    {  // O projection (Modules.cc:308-314): two column tiles per wave -> exchange tile
      SLIMT_WPHASE_LANE;
      load_w(bw[1], L.o, wave + WNW, lane);
      const Epi4 e0 = load_epi4(L.o, wave, lg);
      const Epi4 e1 = load_epi4(L.o, wave + WNW, lg);
      __builtin_amdgcn_sched_barrier(0);
      v4i c0, c1;
      mma_o(bw[0], lane, c0, c1);
      __builtin_amdgcn_sched_barrier(0);
      load_w(bw[0], L.ffn1, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      *reinterpret_cast<float4 *>(Yb + lr * LDY + wave * 16 + lg * 4) = dequant4(c0, e0, L.o.u);
      *reinterpret_cast<float4 *>(Yb + (16 + lr) * LDY + wave * 16 + lg * 4) = dequant4(c1, e0, L.o.u);
      mma_o(bw[1], lane, c0, c1);
      __builtin_amdgcn_sched_barrier(0);
      load_w(bw[1], L.ffn1, wave + WNW, lane);
      __builtin_amdgcn_sched_barrier(0);
      *reinterpret_cast<float4 *>(Yb + lr * LDY + (wave + WNW) * 16 + lg * 4) = dequant4(c0, e1, L.o.u);
      *reinterpret_cast<float4 *>(Yb + (16 + lr) * LDY + (wave + WNW) * 16 + lg * 4) = dequant4(c1, e1, L.o.u);
    }
    lds_barrier();
    SLIMT_WSTAMP(5);
    {  // x = LN(x + O(...)); quantised for FFN1
      SLIMT_WPHASE_LANE;
      float lsc[KSD], lbi[KSD];
      load_ln_regs<KSD>(L.attn_ln_s, L.attn_ln_b, lane, lsc, lbi);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
        for (int i = 0; i < KSD; ++i) x[rr][i] = x[rr][i] + Yb[(2 * wave + rr) * LDY + lane + 64 * i];
        ln_regs<KSD>(x[rr], lsc, lbi, a.eps);
      }
      quantise_x(Abuf, L.ffn1.a_quant, lane);
      const rsrc_t r1c = wrsrc(L.ffn1.colsum, (unsigned)L.ffn1.n_tiles * 64u);
      const rsrc_t r1p = wrsrc(L.ffn1.pb, (unsigned)L.ffn1.n_tiles * 64u);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        ecs[t2] = wload(r1c, lg * 16, (wave + WNW * t2) * 64);
        epb[t2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r1p, lg * 16, (wave + WNW * t2) * 64, 0));
      }
    }
    lds_barrier();  // the exchange tile is dead, FFN1's input is complete
    SLIMT_WSTAMP(6);
    // ---- FFN (Modules.cc:326-331) ----------------------------------------------------------
    // one descriptor per FFN2 column tile, ending with the tile: a prefetch past the last chunk
    // returns zeros without touching memory
    const char *w2 = reinterpret_cast<const char *>(L.ffn2.Wp);
    const rsrc_t r2[2] = {wrsrc(w2 + (size_t)wave * KSF * 1024, KSF * 1024u),
                          wrsrc(w2 + (size_t)(wave + WNW) * KSF * 1024, KSF * 1024u)};
    // FFN2 chunk c (k-steps 4 c .. 4 c + 3 of both column tiles) into buffer `buf`
    auto load2 = [&](int buf, int c, int lane) __attribute__((always_inline)) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bw[buf][t2 * 4 + ks] = wload(r2[t2], lane * 16, (c * 4 + ks) * 1024);
    };
    {  // FFN1: 8 column tiles per wave, two in flight; relu, requantised into the hidden layer
      SLIMT_WPHASE_LANE;
      constexpr int NT1 = (F / 16) / WNW;
      const rsrc_t r1c = wrsrc(L.ffn1.colsum, (unsigned)L.ffn1.n_tiles * 64u);
      const rsrc_t r1p = wrsrc(L.ffn1.pb, (unsigned)L.ffn1.n_tiles * 64u);
#pragma unroll
      for (int i = 0; i < NT1; ++i) {
        const int buf = i & 1, t = wave + WNW * i;
        v4i c0, c1;
        mma(Abuf, bw[buf], lane, c0, c1);
        const v4i cs4 = ecs[buf];
        const float4 pb4 = epb[buf];
        __builtin_amdgcn_sched_barrier(0);
        if (i + 2 < NT1) {
          load_w(bw[buf], L.ffn1, t + 2 * WNW, lane);
          ecs[buf] = wload(r1c, lg * 16, (t + 2 * WNW) * 64);
          epb[buf] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r1p, lg * 16, (t + 2 * WNW) * 64, 0));
        } else
          load2(buf, i + 2 - NT1, lane);  // the first two chunks of FFN2
        __builtin_amdgcn_sched_barrier(0);
        const float pbv[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
        const int csv[4] = {cs4[0], cs4[1], cs4[2], cs4[3]};
        const int col = t * 16 + lg * 4;
        *reinterpret_cast<int *>(Hb + lr * LDH + col) = wrelu_quant4(wdequant4(c0, csv, L.ffn1.u, pbv), L.ffn2.a_quant);
        *reinterpret_cast<int *>(Hb + (16 + lr) * LDH + col) = wrelu_quant4(wdequant4(c1, csv, L.ffn1.u, pbv), L.ffn2.a_quant);
      }
    }
    lds_barrier();  // the hidden layer is complete
    SLIMT_WSTAMP(7);
    {  // FFN2: this wave's two column tiles over K = F, chunks of 4 k-steps, two in flight
      SLIMT_WPHASE_LANE;
      constexpr int NC2 = KSF / 4;
      static_assert(NC2 % 2 == 0, "two chunk buffers");
      v4i f[2][2];  // [column tile][row tile]
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) f[t2][rt] = v4i{0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NC2; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const v4i h0 = *reinterpret_cast<const v4i *>(Hb + lr * LDH + (c * 4 + ks) * 64 + lg * 16);
          const v4i h1 = *reinterpret_cast<const v4i *>(Hb + (16 + lr) * LDH + (c * 4 + ks) * 64 + lg * 16);
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            f[t2][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[buf][t2 * 4 + ks], h0, f[t2][0], 0, 0, 0);
            f[t2][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[buf][t2 * 4 + ks], h1, f[t2][1], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NC2) {
          load2(buf, c + 2, lane);
        } else if (c + 2 == NC2) {  // buffer 0 is free: the next projection's first tile
          if (l + 1 < a.Le)
            load_w(bw[0], a.L[l + 1].q, wave, lane);
          else
            load_w(bw[0], a.dec_k[0], wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const Epi4 e0 = load_epi4(L.ffn2, wave, lg);
      const Epi4 e1 = load_epi4(L.ffn2, wave + WNW, lg);
      lds_barrier();  // every wave has read the hidden layer: the region becomes the exchange tile
      SLIMT_WSTAMP(8);
      *reinterpret_cast<float4 *>(Yb + lr * LDY + wave * 16 + lg * 4) = dequant4(f[0][0], e0, L.ffn2.u);
      *reinterpret_cast<float4 *>(Yb + (16 + lr) * LDY + wave * 16 + lg * 4) = dequant4(f[0][1], e0, L.ffn2.u);
      *reinterpret_cast<float4 *>(Yb + lr * LDY + (wave + WNW) * 16 + lg * 4) = dequant4(f[1][0], e1, L.ffn2.u);
      *reinterpret_cast<float4 *>(Yb + (16 + lr) * LDY + (wave + WNW) * 16 + lg * 4) = dequant4(f[1][1], e1, L.ffn2.u);
    }
    lds_barrier();
    SLIMT_WSTAMP(9);
    {  // x = LN(FFN2(...) + x)
      SLIMT_WPHASE_LANE;
      float lsc[KSD], lbi[KSD];
      load_ln_regs<KSD>(L.ffn_ln_s, L.ffn_ln_b, lane, lsc, lbi);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
#pragma unroll
        for (int i = 0; i < KSD; ++i) x[rr][i] = Yb[r * LDY + lane + 64 * i] + x[rr][i];
        ln_regs<KSD>(x[rr], lsc, lbi, a.eps);
        if (a.layer_out && row_valid(r)) {
          float *dst = a.layer_out + ((size_t)l * B * S + (size_t)row_sentence(r) * S + r % S) * D;
#pragma unroll
          for (int i = 0; i < KSD; ++i) dst[lane + 64 * i] = x[rr][i];
        }
      }
    }
    SLIMT_WSTAMP(10);
    // (the next phase starts with a barrier before it touches the A buffer or the region)
  }

  // ---- encoder output + decoder cross-attention K/V (Modules.cc:248-249, once per batch) ------
  {
    SLIMT_WPHASE_LANE;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;
      if (a.enc_out && row_valid(r)) {
        float *dst = a.enc_out + ((size_t)row_sentence(r) * S + r % S) * D;
#pragma unroll
        for (int i = 0; i < KSD; ++i) dst[lane + 64 * i] = x[rr][i];
      }
    }
  }
  // The packed cache takes one of three forms per workgroup and layer (kernels.h, FusedDecodeArgs::kv_fmt; encode_tall.hip
  // has the same scheme): the tight one -- int16, the SHIFTED accumulator accS = acc + 127 colsum less its column's centre --
  // where the engine allows it (kv_tight_layers) and every such value of this workgroup's rows lies in [-2^15, 2^15); the
  // narrow one -- 20 bits, accS itself -- when every accS lies in [-limit, limit); else 24 bits holding the signed
  // accumulator (accS needs 25 bits at K = 512). The smallest allowed form first; an accumulator that does not fit raises
  // its bit of kv_wide_flag (1: not tight, 2: not narrow) and the layer is done again in the smallest form that holds it.
#ifndef SLIMT_EXP_WIDE_NO_KV  // timing only (no K/V cache is written: wrong results): what the phase costs the LAYERS in registers
  const bool try_narrow = a.kv24 && a.kv_fmt != nullptr;
  for (int l = 0; SLIMT_ONCE_PER_BATCH(l < a.Ld); ++l) {
    const bool try_tight = try_narrow && a.kv_tight_limit > 0 && ((a.kv_tight_layers >> l) & 1u);
    int form = !try_narrow ? 1 : try_tight ? 2 : 0;  // kv_fmt's codes
    for (int attempt = 0; SLIMT_ONCE_PER_BATCH(attempt < 3); ++attempt) {
      const bool wide = form == 1;
      bool redo = false;
      for (int p = 0; SLIMT_ONCE_PER_BATCH(p < 2); ++p) {
        SLIMT_WPHASE_LANE;
        const PreparedWeight &w = p == 0 ? a.dec_k[l] : a.dec_v[l];
        float *out = a.kv + (size_t)(2 * l + p) * B * S * D;
        load_w(bw[1], w, wave + WNW, lane);
        lds_barrier();
        quantise_x(Abuf, w.a_quant, lane);
        // (cleared behind the barrier above, which every thread reaches after its read of the attempt before; raised behind the one below)
        if (!wide && p == 0 && tid == 0) kv_wide_flag = 0;
        lds_barrier();
        unsigned outside = 0;  // an accumulator of a valid row outside the form's range
        const unsigned lim = (unsigned)a.kv_narrow_limit, lim16 = (unsigned)a.kv_tight_limit;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int ct = wave + WNW * t2;
          const Epi4 e = load_epi4(w, ct, lg);
          v4i c0, c1;
          mma(Abuf, bw[t2], lane, c0, c1);
          if (t2 == 0) {  // the next projection's first tile
            __builtin_amdgcn_sched_barrier(0);
            if (p == 0)
              load_w(bw[0], a.dec_v[l], wave, lane);
            else if (l + 1 < a.Ld)
              load_w(bw[0], a.dec_k[l + 1], wave, lane);
            __builtin_amdgcn_sched_barrier(0);
          }
          const int col = ct * 16 + lg * 4;
          if (a.kv24) {  // staged: the signed accumulators (24-bit form), accS (narrow form), or accS - centre (tight form)
            int *stg = reinterpret_cast<int *>(region);  // [WR][LDY] int32
            if (!wide) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                c0[i] += __mul24(127, e.cs[i]);
                c1[i] += __mul24(127, e.cs[i]);
                if (row_valid(lr)) outside |= (unsigned)((unsigned)c0[i] + lim >= 2u * lim) << 1;
                if (row_valid(16 + lr)) outside |= (unsigned)((unsigned)c1[i] + lim >= 2u * lim) << 1;
              }
              if (form == 2) {
                const v4i ctr = *reinterpret_cast<const v4i *>(a.kv_centre[l][p] + col);
                c0 -= ctr;
                c1 -= ctr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  if (row_valid(lr)) outside |= (unsigned)((unsigned)c0[i] + lim16 >= 2u * lim16);
                  if (row_valid(16 + lr)) outside |= (unsigned)((unsigned)c1[i] + lim16 >= 2u * lim16);
                }
              }
            }
            *reinterpret_cast<v4i *>(stg + lr * LDY + col) = c0;
            *reinterpret_cast<v4i *>(stg + (16 + lr) * LDY + col) = c1;
            continue;
          }
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const int rrow = rt * 16 + lr;
            if (!row_valid(rrow)) continue;
            // the f32 form holds float(accS) = float(acc + 127 colsum), exact (kernels.h, kv24)
            const v4i &cc = rt ? c1 : c0;
            const float4 v = {(float)(cc[0] + __mul24(127, e.cs[0])), (float)(cc[1] + __mul24(127, e.cs[1])),
                              (float)(cc[2] + __mul24(127, e.cs[2])), (float)(cc[3] + __mul24(127, e.cs[3]))};
            if (p == 0) {  // K cache layout [sentence][head][d/4][key][4]: the lane's 4 columns are one d/4 group
              const int hh = col / DH, d = col % DH;
              const size_t chunk = ((size_t)row_sentence(rrow) * H + hh) * (DH / 4) + (d >> 2);
              *reinterpret_cast<float4 *>(out + (chunk * S + rrow % S) * 4) = v;
            } else {
              *reinterpret_cast<float4 *>(out + ((size_t)row_sentence(rrow) * S + rrow % S) * D + col) = v;
            }
          }
        }
        if (!a.kv24) continue;
        if (outside) __hip_atomic_fetch_or(&kv_wide_flag, (int)outside, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_barrier();
        // (uniform: ONE read by every thread behind the barrier; a tight attempt that holds does not care about accS's range)
        const int raised = wide ? 0 : kv_wide_flag;
        if (raised & (form == 2 ? 1 : 2)) {
          redo = true;
          form = (raised & 2) ? 1 : 0;
          break;
        }
        const int *stg = reinterpret_cast<const int *>(region);
        const int Sp = (S + 3) & ~3;
        if (form == 2) {
          // the tight form (decode_attention_packed.inl.h, attention_packed32_d64 with Form16): one thread = 32 values = four quads of int16
          if (p == 0) {  // K [sentence][head][plane 0..7][key][16 B]
            const rsrc_t ro = wrsrc(out, (unsigned)((size_t)B * S * D * 3));
            for (int it = tid; SLIMT_ONCE_PER_BATCH(it < WR * (D / 32)); it += 1024) {
              const int r = it % WR, hf = it / WR;  // hf: half a head (32 columns)
              if (!row_valid(r)) continue;
              const int h = hf >> 1, half = hf & 1;
              const int off = row_sentence(r) * S * D * 3 + (h * 8 * S + r % S) * 16;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const v4i pk = pack16(*reinterpret_cast<const v4i *>(stg + r * LDY + 32 * hf + 8 * q),
                                      *reinterpret_cast<const v4i *>(stg + r * LDY + 32 * hf + 8 * q + 4));
                __builtin_amdgcn_raw_buffer_store_b128(pk, ro, off + (4 * half + q) * S * 16, 0, 0);
              }
            }
          } else {  // V [sentence][key / 8][plane 0..3][column / 4][16 B]
            const rsrc_t ro = wrsrc(out, (unsigned)((size_t)B * Sp * D * 3));
            const int G = (S + 7) >> 3;
            for (int it = tid; SLIMT_ONCE_PER_BATCH(it < spw * G * (D / 4)); it += 1024) {
              const int cl = it % (D / 4), g = (it / (D / 4)) % G, si = (it / (D / 4)) / G;
              if (s0 + si >= B) continue;
              const int off = (s0 + si) * Sp * D * 3 + (g * 4 * (D / 4) + cl) * 16;
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
                const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
                const v4i z = {0, 0, 0, 0};
                const v4i x0 = *reinterpret_cast<const v4i *>(stg + (si * S + (k0 < S ? k0 : 0)) * LDY + 4 * cl);
                const v4i x1 = *reinterpret_cast<const v4i *>(stg + (si * S + (k1 < S ? k1 : 0)) * LDY + 4 * cl);
                const v4i pk = pack16(k0 < S ? x0 : z, k1 < S ? x1 : z);
                __builtin_amdgcn_raw_buffer_store_b128(pk, ro, off + q * (D / 4) * 16, 0, 0);
              }
            }
          }
          continue;
        }
        if (!wide) {
          // the narrow form (decode_attention_packed.inl.h, attention_packed32_d64 with Form20): one thread = 32 values = four quads of hi halves +
          // one quad of lo nibbles
          if (p == 0) {  // K [sentence][head][plane 0..9][key][16 B]
            const rsrc_t ro = wrsrc(out, (unsigned)((size_t)B * S * D * 3));
            for (int it = tid; SLIMT_ONCE_PER_BATCH(it < WR * (D / 32)); it += 1024) {
              const int r = it % WR, hf = it / WR;  // hf: half a head (32 columns)
              if (!row_valid(r)) continue;
              const int h = hf >> 1, half = hf & 1;
              const int off = row_sentence(r) * S * D * 3 + (h * 10 * S + r % S) * 16;
              int lo[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const Packed20 pk = pack20(*reinterpret_cast<const v4i *>(stg + r * LDY + 32 * hf + 8 * q),
                                           *reinterpret_cast<const v4i *>(stg + r * LDY + 32 * hf + 8 * q + 4));
                lo[q] = pk.lo;
                __builtin_amdgcn_raw_buffer_store_b128(pk.hi, ro, off + (4 * half + q) * S * 16, 0, 0);
              }
              const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
              __builtin_amdgcn_raw_buffer_store_b128(lq, ro, off + (8 + half) * S * 16, 0, 0);
            }
          } else {  // V [sentence][key / 8][plane 0..4][column / 4][16 B]
            const rsrc_t ro = wrsrc(out, (unsigned)((size_t)B * Sp * D * 3));
            const int G = (S + 7) >> 3;
            for (int it = tid; SLIMT_ONCE_PER_BATCH(it < spw * G * (D / 4)); it += 1024) {
              const int cl = it % (D / 4), g = (it / (D / 4)) % G, si = (it / (D / 4)) / G;
              if (s0 + si >= B) continue;
              const int off = (s0 + si) * Sp * D * 3 + (g * 5 * (D / 4) + cl) * 16;
              int lo[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
                const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
                const v4i z = {0, 0, 0, 0};
                const v4i x0 = *reinterpret_cast<const v4i *>(stg + (si * S + (k0 < S ? k0 : 0)) * LDY + 4 * cl);
                const v4i x1 = *reinterpret_cast<const v4i *>(stg + (si * S + (k1 < S ? k1 : 0)) * LDY + 4 * cl);
                const Packed20 pk = pack20(k0 < S ? x0 : z, k1 < S ? x1 : z);
                lo[q] = pk.lo;
                __builtin_amdgcn_raw_buffer_store_b128(pk.hi, ro, off + q * (D / 4) * 16, 0, 0);
              }
              const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
              __builtin_amdgcn_raw_buffer_store_b128(lq, ro, off + 4 * (D / 4) * 16, 0, 0);
            }
          }
          continue;
        }
        // kernels.h, FusedDecodeArgs::kv24 -- here with the signed accumulator (the decoder adds the
        // column's 127 colsum term): one thread = 16 values = 48 bytes = three 16-byte stores
        if (p == 0) {  // K [sentence][column / 16][plane][key][16 B]
          const rsrc_t ro = wrsrc(out, (unsigned)((size_t)B * S * D * 3));
          for (int it = tid; SLIMT_ONCE_PER_BATCH(it < WR * (D / 16)); it += 1024) {
            const int r = it % WR, ci = it / WR;
            if (!row_valid(r)) continue;
            wv3i wd[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wd[g] = wpack24(*reinterpret_cast<const v4i *>(stg + r * LDY + 16 * ci + 4 * g));
            const int off = row_sentence(r) * S * D * 3 + (ci * 3 * S + r % S) * 16;
            const v4i p0 = {wd[0].x, wd[0].y, wd[0].z, wd[1].x}, p1 = {wd[1].y, wd[1].z, wd[2].x, wd[2].y},
                      p2 = {wd[2].z, wd[3].x, wd[3].y, wd[3].z};
            __builtin_amdgcn_raw_buffer_store_b128(p0, ro, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p1, ro, off + S * 16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p2, ro, off + 2 * S * 16, 0, 0);
          }
        } else {  // V [sentence][key / 4][plane][column / 4][16 B]: 4 keys x 4 columns, key-major
          const rsrc_t ro = wrsrc(out, (unsigned)((size_t)B * Sp * D * 3));
          for (int it = tid; SLIMT_ONCE_PER_BATCH(it < spw * (Sp / 4) * (D / 4)); it += 1024) {
            const int cl = it % (D / 4), g = (it / (D / 4)) % (Sp / 4), si = (it / (D / 4)) / (Sp / 4);
            if (s0 + si >= B) continue;
            wv3i wd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
              const int key = 4 * g + i;
              const v4i xv = *reinterpret_cast<const v4i *>(stg + (si * S + (key < S ? key : 0)) * LDY + 4 * cl);
              const v4i z = {0, 0, 0, 0};
              wd[i] = wpack24(key < S ? xv : z);
            }
            const int off = (((s0 + si) * (Sp / 4) + g) * 3 * (D / 4) + cl) * 16;
            const v4i p0 = {wd[0].x, wd[0].y, wd[0].z, wd[1].x}, p1 = {wd[1].y, wd[1].z, wd[2].x, wd[2].y},
                      p2 = {wd[2].z, wd[3].x, wd[3].y, wd[3].z};
            __builtin_amdgcn_raw_buffer_store_b128(p0, ro, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p1, ro, off + (D / 4) * 16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p2, ro, off + 2 * (D / 4) * 16, 0, 0);
          }
        }
      }
      if (!redo) break;
      {  // bw[0] holds the tile of whatever came next: back to this layer's K
        SLIMT_WPHASE_LANE;
        load_w(bw[0], a.dec_k[l], wave, lane);
      }
    }
    if (a.kv_fmt && a.kv24 && tid < spw && s0 + tid < B) {
      a.kv_fmt[(size_t)l * B + s0 + tid] = (unsigned char)form;
      if (form == 1 && a.kv_wide_count) __hip_atomic_fetch_add(a.kv_wide_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (try_tight && form != 2 && a.kv_not16_count)
        __hip_atomic_fetch_add(a.kv_not16_count + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#endif
  if (gen_here) {
    if (shortlists_await_in_launch(a, tid))  // (never published: nothing to pack from)
      pack_weight_share(a, tile, n_wg, tid, 1024);
  }
  if (tid == 0) occ_trace_event(a.trace, 2, 1);
}

size_t wide_encode_lds_bytes() {
  constexpr int D = 512, F = 2048, RC = 256;
  const size_t qkv = 2 * (size_t)WR * (RC + 4) * 4 + (size_t)WR * (RC + 16) * 4;
  const size_t y = (size_t)WR * (D + 4) * 4, h = (size_t)WR * (F + 32);
  size_t region = qkv > y ? qkv : y;
  region = region > h ? region : h;
  return 3 * (size_t)WR * D + (size_t)WR * (RC + 16) + region;
}

bool wide_encode_supported(int D, int F, int H, int Le, int Ld, int S) {
  if (S < 1 || S > WR || Le < 1 || Le > 6 || Ld < 1 || Ld > 4) return false;
  return D == 512 && F == 2048 && H == 8 && wide_encode_lds_bytes() <= 160 * 1024;
}

hipError_t launch_encode_wide(const FusedEncodeArgs &a, hipStream_t st) {
  if (a.gen.w2o && (!a.ticket || !a.gen_flag || !a.pack_tiles ||
                    shortlist_in_launch_lds_bytes(a.gen.src_vocab, a.gen.tgt_vocab) > wide_encode_lds_bytes()))
    return hipErrorInvalidValue;  // (in-launch shortlist generation: see launch_encode_tall)
  const dim3 grid(fused_encode_grid(a.B, a.S, a.ticket != nullptr));
  const size_t lds = wide_encode_lds_bytes();
  auto k = encode_wide_kernel<8, 32, 64>;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, dim3(1024), lds, st, a);
  return hipGetLastError();
}

}  // namespace slimt_hip
