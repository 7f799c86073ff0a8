// Persistent fused encoder for D = 256 (tiny11: 8 heads of 32) on 64-row tiles: embedding + all
// encoder layers + the decoder's cross-attention K/V cache in ONE launch, 64 rows
// (= floor(64 / S) whole sentences for S <= 32, one sentence of 33..64 tokens) per workgroup.
//
// encode_fused.hip keeps 32 rows per workgroup with everything resident in LDS; every workgroup
// then streams the layer's 1.05 MB of weights for 32 rows, and the CU's L2 path (64 B/clk) is
// what its GEMM phases wait for. Here a weight fragment feeds FOUR row tiles: the O projection
// and the FFN (0.85 MB of the 1.05 MB) cross the L2 path once per 64 rows. To fit 160 KiB:
//
//   * the residual stream lives in REGISTERS (wave w owns rows 4 w .. 4 w + 3, lane L holds
//     columns L + 64 i -- the element-to-lane map of the canonical row sum): LayerNorm, residual
//     adds, the embedding and every quantisation of x run on the owner without a layout change;
//   * heads are staged in two rounds of four: q / k / v of four heads in f32 (102 KiB), attention
//     on the f32 matrix cores, one wave per (sentence, head, 16 queries) -- 16 jobs per round for
//     two 32-token sentences -- exactly the arithmetic of encode_fused.hip;
//   * three int8 A-operand buffers (x quantised with the multipliers of Q, K and V, once per
//     layer), unpadded and XOR-swizzled instead -- with 16 pad bytes per row the third would not
//     fit; round 0's attention output has a narrow buffer of its own, round 1's reuses Q's;
//   * GEMM outputs that meet the residual (O projection, FFN2) cross from the column-tile owner
//     to the row owner through an f32 exchange tile that time-shares the q / k / v region with
//     the FFN's hidden layer (64 x F int8).
//
// Weights are the MFMA A operand (an accumulator lane holds 4 consecutive columns of one row).
// Arithmetic is bit-identical to encode_fused.hip, the layer-by-layer kernels and the oracle's
// portable order. Reference: Model.cc:195-201, Transformer.cc:57-69, Modules.cc:287-334,
// TensorOps.cc:542-580.
#include <cstdlib>

#include "device_common.h"
#include "shortlist_device.h"
#include "kernels.h"

namespace slimt_hip {

namespace {

constexpr int TNW = 16;  // waves per workgroup
constexpr int TR = 64;   // rows per workgroup
constexpr int TRT = 4;   // MFMA row tiles

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t trsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ v4i tload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// epilogue constants of column tile `tile` for this lane's 4 columns (4 lg .. 4 lg + 3)
struct TEpi {
  v4i cs;
  float4 pb;
};
__device__ __forceinline__ TEpi tload_epi(const PreparedWeight &w, int tile, int lg) {
  const rsrc_t rc = trsrc(w.colsum, (unsigned)w.n_tiles * 64u), rp = trsrc(w.pb, (unsigned)w.n_tiles * 64u);
  TEpi e;
  e.cs = tload(rc, lg * 16, tile * 64);
  e.pb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp, lg * 16, tile * 64, 0));
  return e;
}
// y = float(acc + 127 colsum) * u + pb (Intgemm.inl.cc:146-153; |colsum| <= 127 K < 2^23), four
// columns at once, two per packed instruction (v_pk_mul_f32 / v_pk_add_f32 are the same
// IEEE operations as the scalar forms: multiply and add stay separate roundings).
typedef float tf2 __attribute__((ext_vector_type(2)));
// The shift 127 colsum is the accumulator's INITIAL value (tshift, once per column tile: the four row tiles of a
// column share it) instead of an addition in every epilogue: integer arithmetic, the same sum -- `c` below is
// already acc + 127 colsum.
__device__ __forceinline__ v4i tshift(const TEpi &e) {
  return v4i{__mul24(127, e.cs[0]), __mul24(127, e.cs[1]), __mul24(127, e.cs[2]), __mul24(127, e.cs[3])};
}
__device__ __forceinline__ float4 tdequant4(const v4i &c, const TEpi &e, float u) {
  tf2 lo = {(float)c[0], (float)c[1]};
  tf2 hi = {(float)c[2], (float)c[3]};
  const tf2 uu = {u, u};
  lo = lo * uu;
  hi = hi * uu;
  const tf2 pl = {e.pb.x, e.pb.y}, ph = {e.pb.z, e.pb.w};
  lo = lo + pl;
  hi = hi + ph;
  float4 v;
  v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
  return v;
}
// relu, then PrepareA with `aq` (Intgemm.inl.cc:29-34): four int8 in one register (the FFN's hidden
// layer). For aq > 0 (a quantisation multiplier is 127 / max|x|), clamp(rint(max(v, 0) * aq), -127,
// 127) == clamp(rint(v * aq), 0, 127) for every float v: the product keeps the sign, rint(-0) and
// rint(+0) both convert to 0, and a NaN falls to the lower bound of either form (maxNum / minNum
// semantics) -- so the relu rides in the clamp. v_cvt_pk_u8_f32 rounds to nearest even itself, saturates at
// 0 and at 255 and converts a NaN to 0 (tools/probes/cvt_u8_probe.hip, all 2^32 bit patterns:
// profiles/r03_cvt_u8_probe.txt): it is the rint, the lower clamp and the pack in one instruction per value; the
// upper clamp 127 is taken on the four packed bytes at once (a byte >= 128 becomes 127).
__device__ __forceinline__ int trelu_quant4(const v4i &c, const TEpi &e, float u, float aq) {
  const float4 v = tdequant4(c, e, u);
  tf2 lo = {v.x, v.y}, hi = {v.z, v.w};
  const tf2 qq = {aq, aq};
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

// canonical LayerNorm of one row held in registers (v[i] = column lane + 64 i), in place
__device__ __forceinline__ void tln_regs(float (&v)[4], const float (&scale)[4], const float (&bias)[4], float eps) {
  constexpr int D = 256;
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += v[i];
  s = wave_sum(s);
  const float mean = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float d = v[i] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  const float sigma = __builtin_sqrtf(q / (float)D + eps);
  float tq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) tq[i] = v[i] - mean;
  SharedDiv(sigma, SLIMT_DIV_LN_D).quot<4, false>(tq, SLIMT_DIV_LN_N);  // (v - mean) / sigma, correctly rounded
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = tq[i];
    const float m = scale[i] * t;
    v[i] = m + bias[i];
  }
}
__device__ __forceinline__ void tload_ln(const float *scale, const float *bias, int lane, float (&sc)[4], float (&bi)[4]) {
  const rsrc_t rs = trsrc(scale, 1024u), rb = trsrc(bias, 1024u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, i * 256, 0));
    bi[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, i * 256, 0));
  }
}

// four 24-bit two's-complement integers, little endian, in 12 bytes (the packed K/V cache)
typedef int v3i __attribute__((ext_vector_type(3)));
__device__ __forceinline__ v3i tpack24(v4i x) {
  v3i o;
  o.x = (x.x & 0xffffff) | (x.y << 24);
  o.y = ((x.y >> 8) & 0xffff) | (x.z << 16);
  o.z = ((x.z >> 16) & 0xff) | (x.w << 8);
  return o;
}

}  // namespace

// keeps lane-derived offsets from being hoisted out of the layer loop and spilled (decode_fused.hip)
#define SLIMT_TPHASE_LANE                               \
  int lane = lane0;                                     \
  asm volatile("" : "+v"(lane));                        \
  const int lr = lane & 15, lg = lane >> 4;             \
  (void)lr;                                             \
  (void)lg

// Diagnostic phase stamps (100 MHz wall clock) of workgroup 0 in one layer (slots 0..10).
#define SLIMT_TSTAMP(id)                                                              \
  do {                                                                                \
    if (a.stamps && s0 == 0 && tid == 0 && l == a.stamp_layer)                        \
      a.stamps[(id)] = wall_clock64();                                                \
  } while (0)

// NKT = key tiles of 16 per sentence: 2 (S <= 32, floor(64 / S) sentences per workgroup) or 4
// (33 <= S <= 64: one sentence per workgroup -- what encode_long16_kernel does through global
// scratch tensors stays in LDS here).
// NRT = row tiles of 16 the workgroup's sentences occupy: 4, or 3 when they end within 48 rows (two sentences of 22..24
// tokens, one of 33..48) -- the fourth tile's MFMAs, epilogues, LayerNorm rows and quantisations would be padding only.
template <int KSF, int NKT = 2, int NRT = TRT>
__global__ __launch_bounds__(1024) void encode_tall_kernel(FusedEncodeArgs a) {
  static_assert(NRT == TRT || NRT == TRT - 1, "all four row tiles, or the first three");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KSD = 4, D = 256, DH = 32, F = 64 * KSF;
  constexpr int HR = 4;        // heads per round
  constexpr int RC = HR * DH;  // q / k / v columns per round (128)
  constexpr int NR = 2;        // rounds
  // int8 A rows: unpadded, the 16-byte chunks of a row XOR-swizzled with the row (quantise_x below)
  // -- a fragment read (16 rows x one chunk) then covers all 64 banks without the 16 pad bytes
  // per row, which is what lets THREE A buffers (x quantised for Q, K, V) fit beside the rest
  constexpr int LDA = D;
  constexpr int LDO = RC + 16; // int8 attention output rows of one round
  // f32 q / k / v rows: the attention's 16x16x4 operands are read by lanes (row n = lane % 16,
  // k index g = lane / 16) -- q, k at [row n][d + g]: stride = 4 mod 64 words is conflict-free;
  // v at [key g][d + n]: stride = 16 mod 64 is
  // q / k rows: + 2 floats. The attention reads them one float per lane (row n = lane % 16, k index
  // g = lane / 16, ds_read_b32: 32 banks, half a wave per LDS cycle): 2 n + g is a different bank for every
  // lane of a half wave; with + 4 rows n and n + 8 shared banks (every operand read 2-way conflicted).
  // Rows are then 8-byte aligned: the projections store two float2 instead of one float4.
  constexpr int LDQ = RC + 2;
  constexpr int LDV = RC + 16;
  constexpr int LDY = D + 4;   // f32 exchange rows (also the int32 rows of the K/V staging tile)
  // int8 hidden rows: + 32 bytes, so that an FFN2 fragment read (ds_read_b128: 16 rows x 16 B per
  // hardware lane group) covers all 64 banks -- with + 16 the dword index 4 (lr + lg) put rows lr and lr + 1 of
  // neighbouring lane groups on the same banks: every read 2-way conflicted (8 instead of 4 LDS cycles,
  // 1.5 MB of such reads per layer; SQ_LDS_BANK_CONFLICT was 29 % of the kernel's LDS cycles)
  constexpr int LDH = F + 32;
  constexpr int NT1 = (F / 16) / TNW;  // FFN1 column tiles per wave
  constexpr int NC2 = KSF / 4;         // FFN2 chunks of four k-steps
  static_assert((F / 16) % TNW == 0 && KSF % 4 == 0, "whole tiles / whole chunks per wave");
  static_assert(D / 16 == TNW, "one 16-column tile of a D-wide GEMM per wave");
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, B = a.B;
  const int spw = TR / S;  // whole sentences per workgroup

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
  // this wave owns rows 4 wave .. 4 wave + 3 in the row-wise phases (embedding, LayerNorm, quantisation): with NRT = 3 the
  // last four waves' rows are beyond every sentence, and what they would read there (the fourth row tile of the exchange
  // tile) is never written
  const bool owner = NRT == TRT || wave < 4 * NRT;
  if (tid == 0) occ_trace_event(a.trace, 2, 0);
  // this workgroup's sentence lengths: read once (they may live in pinned host memory, and every
  // attention job needs its sentence's; ordered by the first barrier below)
  __shared__ int slens[TR];
  // the narrow cache form does not hold this workgroup's accumulators (the K/V phase at the end). TWO words, used in turn
  // (kvf): an attempt's flag is cleared in front of its first barrier while a lagging wave may still be reading the flag of
  // the attempt before -- a different word, whose next clearing lies behind two barriers every reader has passed by then.
  __shared__ int kv_wide_flag[2];
  int kvf = 0;
  if (tid < spw) slens[tid] = s0 + tid < B ? sentence_length(a, s0 + tid, S) : 0;

  char *Aq = smem;                       // x quantised for Q | round 1's attention output | for FFN1 | for the decoder's K / V
  char *Ak = Aq + TR * LDA;              // x quantised for K
  char *Av = Ak + TR * LDA;              // x quantised for V
  char *Ob0 = Av + TR * LDA;             // [TR][LDO] attention output of round 0, int8
  char *Ob1 = Aq;                        // ... of round 1: Q's operand is dead by then
  char *region = Ob0 + TR * LDO;         // q, k, v of four heads | exchange tile | hidden layer
  static_assert(TR * LDO <= TR * LDA, "round 1's attention output fits Q's operand buffer");
  float *qb = reinterpret_cast<float *>(region);
  float *kb = qb + TR * LDQ;
  float *vb = kb + TR * LDQ;  // rows of LDV floats
  float *Yb = reinterpret_cast<float *>(region);
  char *Hb = region;
  static_assert((size_t)TR * LDY * 4 <= (size_t)TR * (2 * LDQ + LDV) * 4, "exchange tile fits q / k / v");
  static_assert((size_t)TR * LDH <= (size_t)TR * (2 * LDQ + LDV) * 4, "hidden layer fits q / k / v");

  auto row_sentence = [&](int r) { return s0 + r / S; };
  auto row_valid = [&](int r) { return r < rows_used && row_sentence(r) < B; };

  // side job: the batch's shortlisted output layer (used by the decoder launch behind this one)
  const bool gen_here = a.gen.w2o != nullptr;  // ... of a shortlist this launch generates itself: packed at the end
  if (!gen_here) {
    pack_weight_share(a, tile, n_wg, tid, 1024);
  } else {
    // ShortlistGenerator::generate (Shortlist.cc:115-175; Model.cc:117-120) by the workgroup that started first (a merged
    // launch: by the first n, one shortlist per sub-batch), in the still unused LDS, then published for the others (shortlist_device.h)
    shortlists_publish_in_launch(a, reinterpret_cast<uint32_t *>(smem), tile, n_wg, tid);
  }

  // A round's projections are 3 x 8 column tiles (Q, K, V of four heads) x 4 row tiles on 16 waves:
  // waves 0..7 take Q tile w and V tile w, waves 8..15 K tile w - 8, all four row tiles each -- every
  // weight tile crosses the L2 path once. The first tile (Q or K) is requested a phase ahead: under
  // the last LayerNorm of the layer before (round 0) or at the top of round 1.
  v4i w1[KSD];
  auto load_first = [&](const FusedEncLayerW &Lw, int hr, int lane) {
    const PreparedWeight &W = wave < 8 ? Lw.q : Lw.k;
    const int ct = hr * (RC / 16) + (wave & 7);
    const rsrc_t rw = trsrc(W.Wp, (unsigned)W.n_tiles * KSD * 1024u);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) w1[ks] = tload(rw, lane * 16, (ct * KSD + ks) * 1024);
  };
  {
    SLIMT_TPHASE_LANE;
    load_first(a.L[0], 0, lane);
  }

  // ---- embedding (Model.cc:195-197) into the owner's registers ------------------------------
  float x[4][KSD];
  {
    SLIMT_TPHASE_LANE;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wave + rr;
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
  // the owner's rows, quantised for the next affine, into an A buffer.
  // Byte offset of (row, column): row LDA + (((column / 16) ^ f(row)) << 4) + column % 16 with
  //   f(r) = ((r & 3) << 2) | (((r >> 2) + 1) & 3)   (r = row & 15: the 2-bit halves swapped, the upper one + 1).
  // A fragment read (ds_read_b128, serviced in the hardware's four groups of 16 lanes: tools/lds_conflicts.py) must find 16
  // different chunks per group; of the bijections that do, this one ALSO puts the owner's stores at one lane-dependent
  // address plus compile-time offsets: row 4 w + rr, column L + 64 i -> chunk ((L / 16) ^ ((w + 1) & 3)) | ((i ^ rr) << 2), i.e.
  //   A + 1024 w + (((L / 16) ^ ((w + 1) & 3)) << 4) + L % 16   (tq_base: one register per phase)   + 256 rr + ((i ^ rr) << 6).
  // (f = identity: the i-term is (i ^ (w & 3)) << 6, two to three address instructions in front of every byte store; the plain
  // swap of the halves: every fragment read 2-way conflicted, SQ_LDS_BANK_CONFLICT 15 -> 35 % of the LDS cycles.)
  typedef __attribute__((address_space(3))) char *lds_bptr;
  auto tq_base = [&](int lane) {
    lds_bptr p = (lds_bptr)(smem + 1024 * wave + ((((lane >> 4) ^ (wave + 1)) & 3) << 4) + (lane & 15));
    asm volatile("" : "+v"(p));
    return p;
  };
  // PrepareA (Intgemm.inl.cc:29-34) for a BYTE store: clamp(rint(x aq), -127, 127) == rint(clamp(x aq, -127, 127)) (the bounds
  // are integers, rint is monotonic), and adding 1.5 * 2^23 rounds to nearest even and leaves the integer's two's complement
  // in the low mantissa bits: multiply, v_med3 (a NaN product -> -127, like the max / min pair), add -- the store takes
  // the low byte. (quantize1: multiply, v_rndne, v_med3, v_cvt_i32.)
  auto quant_byte = [](float v, float aq) {
    const float t = __builtin_amdgcn_fmed3f(v * aq, -127.0f, 127.0f);
    return __float_as_int(t + 12582912.0f);
  };
  // aoff: the A buffer's byte offset in smem (a constant at every call)
  auto quantise_x = [&](int aoff, float aq, lds_bptr qb0) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int i = 0; i < KSD; ++i) qb0[aoff + 256 * rr + ((i ^ rr) << 6)] = (char)quant_byte(x[rr][i], aq);
  };
  auto load_w = [&](v4i (&f)[KSD], const PreparedWeight &w, int ct, int lane) {
    const rsrc_t rw = trsrc(w.Wp, (unsigned)w.n_tiles * KSD * 1024u);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) f[ks] = tload(rw, lane * 16, (ct * KSD + ks) * 1024);
  };
  // one 16-column weight tile (K = D) against row tile rt of `A`: accumulator lane = row
  // 16 rt + lr, columns 4 lg .. 4 lg + 3 of the tile
  // This lane's fragment offsets inside a row tile of an A buffer, one per k-step: the XOR swizzle depends
  // on (lr, lg, ks) only, a row tile adds the constant 16 rt LDA -- computed ONCE per phase (AFrag) and
  // pinned, the reads then carry an immediate offset. Recomputed per read (what the compiler does
  // with the plain expression) it is three VALU instructions per fragment, 12 per (tile, row tile) next
  // to 4 MFMAs and a 29-instruction epilogue.
  typedef const __attribute__((address_space(3))) char *lds_cptr;
  struct AFrag {
    lds_cptr p[KSD];
  };
  auto a_frag = [&](const char *A, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    AFrag o;
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      o.p[ks] = (lds_cptr)(A + lr * LDA + (((ks * 4 + lg) ^ (((lr & 3) << 2) | (((lr >> 2) + 1) & 3))) << 4));
      asm volatile("" : "+v"(o.p[ks]));
    }
    return o;
  };
  auto mma_rt = [&](const v4i (&f)[KSD], int rt, const AFrag &o, const v4i &init) {
    v4i c = init;
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      const v4i av = *reinterpret_cast<const __attribute__((address_space(3))) v4i *>(o.p[ks] + 16 * rt * LDA);
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks], av, c, 0, 0, 0);
    }
    return c;
  };

  // One weight tile against every row tile of `A`, epi(rt, c) per row tile: a row tile's four fragments are requested
  // together, the next row tile's right behind this one's MFMAs (their round trip runs under this one's epilogue).
  // Written out because the plain loop (mma_rt per row tile) compiles to read -> wait -> MFMA, fragment by fragment.
  auto gemm_rts = [&](const v4i (&f)[KSD], const AFrag &o, const v4i &init, auto &&epi) {
    typedef const __attribute__((address_space(3))) v4i *lds_v4i;
    v4i A[KSD];
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) A[ks] = *(lds_v4i)(o.p[ks]);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      v4i c = init;
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks], A[ks], c, 0, 0, 0);
      if (rt + 1 < NRT) {
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) A[ks] = *(lds_v4i)(o.p[ks] + 16 * (rt + 1) * LDA);
      }
      __builtin_amdgcn_sched_barrier(0);
      epi(rt, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int l = 0; l < a.Le; ++l) {
    const FusedEncLayerW &L = a.L[l];
    SLIMT_TSTAMP(0);
    // ---- Attention::forward (Modules.cc:287-319), four heads per round -----------------------
    // (24 registers across an attention round: with four key tiles -- sentences of 33..64 tokens -- they spill)
    constexpr bool R1_AHEAD = NKT == 2;
    TEpi e1_next;
#pragma unroll
    for (int hr = 0; hr < NR; ++hr) {
#ifdef SLIMT_EXP_QKV_STAMPS  // diagnosis only: every wave's clock at four points of round SLIMT_EXP_QKV_STAMPS (slots wave * 3 + i, i < 3;
                             // 48 + 11 + wave is taken: the fourth goes without)
#define SLIMT_WSTAMP(i)                                                                                             \
  do {                                                                                                              \
    if (a.stamps && s0 == 0 && (tid & 63) == 0 && l == a.stamp_layer && hr == SLIMT_EXP_QKV_STAMPS)                 \
      (a.stamps - 48)[wave * 3 + (i)] = wall_clock64();                                                             \
  } while (0)
#else
#define SLIMT_WSTAMP(i)
#endif
      const int ctl = wave & 7;
      const int ct = hr * (RC / 16) + ctl;
      {  // Q, K, V projections of this round's heads
        SLIMT_TPHASE_LANE;
        const bool qv = wave < 8;  // this wave: Q and V tiles, else the K tile
        const PreparedWeight &W1 = qv ? L.q : L.k;
        // round 1's first tile and its epilogue constants were requested before round 0's attention where the registers
        // allow it (R1_AHEAD); else here, at the top of the round
        if (hr > 0 && !R1_AHEAD) {
          load_first(L, hr, lane);
          __builtin_amdgcn_sched_barrier(0);
        }
        TEpi e1;
        if (hr > 0 && R1_AHEAD)
          e1 = e1_next;
        else
          e1 = tload_epi(W1, ct, lg);
        // V's tile: loaded by every wave (an unconditional definition keeps its registers out of the
        // K waves' way; their copy is never used, its 8 redundant fetches hit in L2)
        v4i wv[KSD];
        load_w(wv, L.v, ct, lane);
        const TEpi ev = tload_epi(L.v, ct, lg);
        __builtin_amdgcn_sched_barrier(0);
        // round 1: the region is free (round 0's attention has read its q / k / v). Round 0 of a later layer needs no barrier
        // here: the A buffers were last read by FFN1 (two barriers ago), and nothing below touches the region -- still the
        // exchange tile the LayerNorm before read -- until the barrier behind the quantisation. (Layer 0: the in-launch
        // shortlist generation used this memory.)
        SLIMT_WSTAMP(0);
        if (hr > 0 || l == 0) lds_barrier();
        SLIMT_WSTAMP(1);
        if (hr == 0) {  // x quantised with the three projections' multipliers, once per layer
          const lds_bptr qb0 = tq_base(lane);
          if (owner) {
            quantise_x(0, L.q.a_quant, qb0);
            quantise_x(TR * LDA, L.k.a_quant, qb0);
            quantise_x(2 * TR * LDA, L.v.a_quant, qb0);
          }
          SLIMT_WSTAMP(1);
          lds_barrier();
        }
        SLIMT_WSTAMP(2);
        const char *A1 = qv ? Aq : Ak;
        float *dst1 = qv ? qb : kb;
        const AFrag af = a_frag(A1, lane);
        const v4i s1 = tshift(e1);
        gemm_rts(w1, af, s1, [&](int rt, const v4i &c) {
          const float4 qk = tdequant4(c, e1, W1.u);
          float *qd = dst1 + (16 * rt + lr) * LDQ + ctl * 16 + lg * 4;
          *reinterpret_cast<float2 *>(qd) = float2{qk.x, qk.y};
          *reinterpret_cast<float2 *>(qd + 2) = float2{qk.z, qk.w};
        });
        __builtin_amdgcn_sched_barrier(0);
        if (qv) {
          const AFrag afv = a_frag(Av, lane);
          const v4i sv = tshift(ev);
          gemm_rts(wv, afv, sv, [&](int rt, const v4i &cv) {
            *reinterpret_cast<float4 *>(vb + (16 * rt + lr) * LDV + ctl * 16 + lg * 4) = tdequant4(cv, ev, L.v.u);
          });
        }
      }
      if (hr == 0 && R1_AHEAD) {  // their round trip runs under round 0's attention instead of in front of round 1's first MFMA
        SLIMT_TPHASE_LANE;
        load_first(L, 1, lane);
        e1_next = tload_epi(wave < 8 ? L.q : L.k, (RC / 16) + (wave & 7), lg);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (hr == 1 && R1_AHEAD) {  // ... and the O projection's tile (the same registers) under round 1's
        SLIMT_TPHASE_LANE;
        load_w(w1, L.o, wave, lane);
        e1_next = tload_epi(L.o, wave, lg);
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
      SLIMT_TSTAMP(hr == 0 ? 1 : 3);
      // scaled_dot_product_attention (Modules.cc:24-86) on the f32 matrix cores: one wave per
      // (sentence, head of the round, 16 queries). v_mfma_f32_16x16x4_f32 chains over ascending k
      // (bit-identical to the ascending fmaf chain, tools/probe_mfma_f32.py); operand maps,
      // butterfly order and the lane-group transpose of P as in encode_fused.hip.
      {
        SLIMT_TPHASE_LANE;
        typedef float v4f __attribute__((ext_vector_type(4)));
        char *Or = hr == 0 ? Ob0 : Ob1;
        const int n = lane & 15, g = lane >> 4;
        const float minus_inf = -99999999.0f;  // Input.cc:56-61
        const float lowest = -3.402823466e+38f;
        const int nqh = (S + 15) >> 4;  // 16-query tiles of a sentence
        for (int job = wave; job < spw * HR * nqh; job += TNW) {
          // job = (sl HR + hl) nqh + qh without a division: nqh is 1 or 2 (S <= 32), 3 or 4 (one sentence per workgroup) --
          // the general form kept a dozen reciprocal constants in scalar registers that spilled into this loop
          int qh, rest;
          if constexpr (NKT == 2) {
            qh = nqh == 2 ? (job & 1) : 0;
            rest = nqh == 2 ? (job >> 1) : job;
          } else {
            rest = nqh == 4 ? (job >> 2) : (job * 11) >> 5;  // job / 3 for job < 16
            qh = job - rest * nqh;
          }
          const int hl = rest & (HR - 1), sl = rest / HR;
          static_assert(HR == 4, "hl = rest & 3");
          const int sb = s0 + sl;
          if (sb >= B) continue;
          const int base = sl * S;
          const int len = slens[sl];
          const int qr = 16 * qh + n;
          const float *qp = qb + __mul24(base + (qr < S ? qr : S - 1), LDQ) + hl * DH + g;
          float sc[NKT][4];
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            const int kr = 16 * kt + n;
            const float *kp = kb + __mul24(base + (kr < S ? kr : S - 1), LDQ) + hl * DH + g;
            v4f st = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k0 = 0; k0 < DH; k0 += 4) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[k0], qp[k0], st, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int m = 16 * kt + 4 * g + r;  // key of this register
              float v = st[r];
              v = a.alpha * v;  // (alpha == 1: the product is v itself, bit for bit -- no select per score)
              v = v + (1.0f - (m < len ? 1.0f : 0.0f)) * minus_inf;
              if (m >= S) v = lowest;
              sc[kt][r] = v;
            }
          }
          float mx = lowest;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
            mx = fmaxf(mx, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
          mx = bf_max<32>(bf_max<16>(mx));
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = exp_p_select(sc[kt][r] - mx);
              sc[kt][r] = (16 * kt + 4 * g + r) < S ? e : 0.0f;
            }
          // the canonical butterfly over the key index 16 kt + 4 g + r: masks 1, 2 inside the four
          // registers, 4 and 8 across lane groups, 16 and 32 across the key tiles
          float t[NKT];
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
            t[kt] = bf_add<32>(bf_add<16>((sc[kt][0] + sc[kt][1]) + (sc[kt][2] + sc[kt][3])));
          float sum = t[0] + t[1];
          if constexpr (NKT == 4) sum = sum + (t[2] + t[3]);
          float pa[NKT][4];  // pa[kt][j] on lane (n, g) = P[query n][key 16 kt + 4 j + g]
          {  // e / sum for this lane's 4 NKT keys: one refined reciprocal (device_common.h, SharedDiv)
            const SharedDiv dv(sum, SLIMT_DIV_SM_D);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) dv.quot<4, true>(sc[kt], SLIMT_DIV_SM_N);  // keys >= S: exactly 0
          }
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            const slimt_u2 s01 = __builtin_amdgcn_permlane16_swap(__float_as_int(sc[kt][0]), __float_as_int(sc[kt][1]), false, false);
            const slimt_u2 s23 = __builtin_amdgcn_permlane16_swap(__float_as_int(sc[kt][2]), __float_as_int(sc[kt][3]), false, false);
            const slimt_u2 ac = __builtin_amdgcn_permlane32_swap(s01.x, s23.x, false, false);
            const slimt_u2 bd = __builtin_amdgcn_permlane32_swap(s01.y, s23.y, false, false);
            pa[kt][0] = __int_as_float(ac.x);
            pa[kt][1] = __int_as_float(bd.x);
            pa[kt][2] = __int_as_float(ac.y);
            pa[kt][3] = __int_as_float(bd.y);
          }
          // V operands: this lane's key rows (4 s4 + g), both column tiles, requested before the first MFMA -- with the row
          // address inside the chain the compiler put a quarter-rate 32-bit multiply, a read and its wait in front of every MFMA
          // (rows are < 2^24: 24-bit multiplies, full rate)
          typedef const __attribute__((address_space(3))) float *lds_fp;
          const lds_fp vcol = (lds_fp)(vb + hl * DH + n);
          float vv[DH / 16][4 * NKT];
#pragma unroll
          for (int s4 = 0; s4 < 4 * NKT; ++s4) {  // keys >= S contribute fma(0, v, o) == o
            const int key = 4 * s4 + g;
            const int off = __mul24(base + (key < S ? key : S - 1), LDV);
#pragma unroll
            for (int nt = 0; nt < DH / 16; ++nt) vv[nt][s4] = vcol[off + 16 * nt];
          }
          const float aq_o = L.o.a_quant;
          typedef __attribute__((address_space(3))) char *lds_wp;
          const lds_wp orow = (lds_wp)(Or + hl * DH + n);
#pragma unroll
          for (int nt = 0; nt < DH / 16; ++nt) {
            v4f o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s4 = 0; s4 < 4 * NKT; ++s4) o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s4 >> 2][s4 & 3], vv[nt][s4], o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int q = 16 * qh + 4 * g + r;  // query of this register
              if (q < S) orow[__mul24(base + q, LDO) + 16 * nt] = (char)quant_byte(o[r], aq_o);
            }
          }
        }
        // rows that belong to no sentence keep a defined A operand
        for (int r = rows_used + wave; r < TR; r += TNW)
#pragma unroll
          for (int i = 0; i < RC / 64; ++i) Or[r * LDO + lane + 64 * i] = 0;
      }
      if (hr == 0) SLIMT_TSTAMP(2);
    }
    // ---- O projection (Modules.cc:308-314): wave = column tile, four row tiles -> exchange tile
    {
      SLIMT_TPHASE_LANE;
      v4i wo_here[KSD];
      TEpi eo;
      if (!R1_AHEAD) {
        load_w(wo_here, L.o, wave, lane);
        eo = tload_epi(L.o, wave, lg);
      } else {
        eo = e1_next;
      }
      const v4i(&wo)[KSD] = R1_AHEAD ? w1 : wo_here;
      lds_barrier();  // attention of the last round is complete: q / k / v are dead
      SLIMT_TSTAMP(4);
      const v4i so = tshift(eo);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        v4i c = so;
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {  // k-steps 0, 1: round 0's heads; 2, 3: round 1's
          const v4i av = *reinterpret_cast<const v4i *>((ks < 2 ? Ob0 : Ob1) + (16 * rt + lr) * LDO + (ks & 1) * 64 + lg * 16);
          c = __builtin_amdgcn_mfma_i32_16x16x64_i8(wo[ks], av, c, 0, 0, 0);
        }
        *reinterpret_cast<float4 *>(Yb + (16 * rt + lr) * LDY + wave * 16 + lg * 4) = tdequant4(c, eo, L.o.u);
      }
    }
    // FFN1's first column tiles travel under the LayerNorm
    v4i bw[3][KSD];
    auto load1 = [&](int buf, int i, int lane) { load_w(bw[buf], L.ffn1, wave + TNW * i, lane); };
    // FFN1's epilogue constants (column sums and prepared biases of its F columns, 8 F bytes) wait in LDS -- K's operand
    // buffer is idle from the last projection round to the next layer -- instead of in 24 registers per wave (three tiles
    // in flight): those registers hold a row tile's four A fragments, requested at once and one row tile ahead (below)
    int *ecs = reinterpret_cast<int *>(Ak);
    float *epb = reinterpret_cast<float *>(Ak + F * 4);
    static_assert((size_t)F * 8 <= (size_t)TR * LDA, "FFN1's epilogue constants fit K's operand buffer");
    {
      SLIMT_TPHASE_LANE;
      float lsc[4], lbi[4];
      tload_ln(L.attn_ln_s, L.attn_ln_b, lane, lsc, lbi);
      // (F / 1024 words of each array per thread; requested here, stored behind the LayerNorm)
      constexpr int EPT = (F + 1023) / 1024;
      int ecv[EPT];
      float epv[EPT];
      {
        const rsrc_t rc = trsrc(L.ffn1.colsum, (unsigned)F * 4u), rp = trsrc(L.ffn1.pb, (unsigned)F * 4u);
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          ecv[i] = __builtin_amdgcn_raw_buffer_load_b32(rc, (tid + 1024 * i) * 4, 0, 0);  // (past F: zeros nobody reads)
          epv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (tid + 1024 * i) * 4, 0, 0));
        }
      }
      lds_barrier();
      SLIMT_TSTAMP(5);
      // x = LN(x + O(...)); quantised for FFN1
      if (owner) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
          for (int i = 0; i < KSD; ++i) x[rr][i] = x[rr][i] + Yb[(4 * wave + rr) * LDY + lane + 64 * i];
          tln_regs(x[rr], lsc, lbi, a.eps);
        }
        quantise_x(0, L.ffn1.a_quant, tq_base(lane));
      }
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (tid + 1024 * i < F) {
          ecs[tid + 1024 * i] = ecv[i];
          epb[tid + 1024 * i] = epv[i];
        }
#pragma unroll
      for (int i = 0; i < 3 && i < NT1; ++i) {
        load1(i, i, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_barrier();  // the exchange tile is dead, FFN1's input is complete
    SLIMT_TSTAMP(6);
    // ---- FFN (Modules.cc:326-331) ----------------------------------------------------------
    const rsrc_t r2 = trsrc(reinterpret_cast<const char *>(L.ffn2.Wp) + (size_t)wave * KSF * 1024, (unsigned)KSF * 1024u);
    v4i b2[3][4];
    auto load2 = [&](int buf, int c, int lane) {  // FFN2 chunk c: k-steps 4 c .. 4 c + 3 of this wave's column tile
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b2[buf][ks] = tload(r2, lane * 16, (c * 4 + ks) * 1024);
    };
    {  // FFN1: NT1 column tiles per wave, three in flight; relu, requantised into the hidden layer
      SLIMT_TPHASE_LANE;
      const AFrag af = a_frag(Aq, lane);
      typedef const __attribute__((address_space(3))) v4i *lds_v4i;
      // Measured by ablation (timing-only builds, round 3): this phase was 9.2 us = the weight stream alone
      // (3.0: 384 KB at the CU's 64 B/clk) + the MFMAs and their A-fragment reads (3.1) + the epilogues (3.1). The ISA showed why
      // they added up: with 3 x 8 registers of epilogue constants held per wave, the compiler had two register quads for A
      // fragments and serialised read -> wait -> MFMA three times per row tile, then the epilogue, then the next row tile's reads.
      // Now a row tile's four fragments are requested together, the next row tile's right behind this one's MFMAs: their
      // round trip runs under this one's epilogue.
      v4i A[KSD];
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks) A[ks] = *(lds_v4i)(af.p[ks]);
#pragma unroll
      for (int i = 0; i < NT1; ++i) {
        const int buf = i % 3, t = wave + TNW * i;
        TEpi e;
        e.cs = *(lds_v4i)((lds_cptr)(ecs + t * 16 + lg * 4));
        e.pb = __builtin_bit_cast(float4, *(lds_v4i)((lds_cptr)(epb + t * 16 + lg * 4)));
        const v4i sh = tshift(e);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          v4i c = sh;
#pragma unroll
          for (int ks = 0; ks < KSD; ++ks) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[buf][ks], A[ks], c, 0, 0, 0);
          // the next row tile's fragments (the next column tile starts over at row tile 0)
          if (rt + 1 < NRT || i + 1 < NT1) {
#pragma unroll
            for (int ks = 0; ks < KSD; ++ks) A[ks] = *(lds_v4i)(af.p[ks] + 16 * ((rt + 1) % NRT) * LDA);
          }
          __builtin_amdgcn_sched_barrier(0);
          *reinterpret_cast<int *>(Hb + (16 * rt + lr) * LDH + t * 16 + lg * 4) = trelu_quant4(c, e, L.ffn1.u, L.ffn2.a_quant);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i + 3 < NT1) load1(buf, i + 3, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      // FFN2's first weight chunks do not depend on the hidden layer: requested before the barrier
#pragma unroll
      for (int c = 0; c < 3 && c < NC2; ++c) {
        load2(c, c, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_barrier();  // the hidden layer is complete
    SLIMT_TSTAMP(7);
    {  // FFN2: this wave's column tile over K = F, chunks of 4 k-steps, three in flight
      SLIMT_TPHASE_LANE;
      const TEpi e2 = tload_epi(L.ffn2, wave, lg);  // its shift starts the accumulators
      v4i f[NRT];
      {
        const v4i s2 = tshift(e2);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) f[rt] = s2;
      }
      // A k-step's four hidden-layer fragments (one per row tile) are requested together, the next k-step's right behind this
      // one's four MFMAs (independent accumulators: they issue back to back). The plain loop compiled to read -> wait -> MFMA
      // 96 times per wave (one register quad for the fragment); the phase went from 4.3-4.5 to 4.0-4.3 us.
      typedef const __attribute__((address_space(3))) v4i *lds_v4i;
      typedef const __attribute__((address_space(3))) char *lds_c;
      const lds_c hrow = (lds_c)(Hb + lr * LDH + lg * 16);
      v4i H[NRT];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) H[rt] = *(lds_v4i)(hrow + 16 * rt * LDH);
#pragma unroll
      for (int c = 0; c < NC2; ++c) {
        const int buf = c % 3;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) f[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b2[buf][ks], H[rt], f[rt], 0, 0, 0);
          if (c * 4 + ks + 1 < KSF) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) H[rt] = *(lds_v4i)(hrow + 16 * rt * LDH + (c * 4 + ks + 1) * 64);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 3 < NC2) load2(buf, c + 3, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next layer's first Q / K tiles: under the LayerNorm (unconditional, so that the
      // registers are dead between the projections and here)
      load_first(a.L[l + 1 < a.Le ? l + 1 : l], 0, lane);
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();  // every wave has read the hidden layer: the region becomes the exchange tile
      SLIMT_TSTAMP(8);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        *reinterpret_cast<float4 *>(Yb + (16 * rt + lr) * LDY + wave * 16 + lg * 4) = tdequant4(f[rt], e2, L.ffn2.u);
    }
    {  // x = LN(FFN2(...) + x)
      SLIMT_TPHASE_LANE;
      float lsc[4], lbi[4];
      tload_ln(L.ffn_ln_s, L.ffn_ln_b, lane, lsc, lbi);
      lds_barrier();
      SLIMT_TSTAMP(9);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (!owner) break;
        const int r = 4 * wave + rr;
#pragma unroll
        for (int i = 0; i < KSD; ++i) x[rr][i] = Yb[r * LDY + lane + 64 * i] + x[rr][i];
        tln_regs(x[rr], lsc, lbi, a.eps);
        if (a.layer_out && row_valid(r)) {
          float *dst = a.layer_out + ((size_t)l * B * S + (size_t)row_sentence(r) * S + r % S) * D;
#pragma unroll
          for (int i = 0; i < KSD; ++i) dst[lane + 64 * i] = x[rr][i];
        }
      }
    }
    SLIMT_TSTAMP(10);
    // (the next phase starts with a barrier before it touches an A buffer or the region)
  }

  // ---- encoder output + decoder cross-attention K/V (Modules.cc:248-249, once per batch) ------
  {
    SLIMT_TPHASE_LANE;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wave + rr;
      if (a.enc_out && row_valid(r)) {
        float *dst = a.enc_out + ((size_t)row_sentence(r) * S + r % S) * D;
#pragma unroll
        for (int i = 0; i < KSD; ++i) dst[lane + 64 * i] = x[rr][i];
      }
    }
  }
  // The packed cache takes one of three forms per workgroup and layer (kernels.h, FusedDecodeArgs::kv_fmt): the tight one,
  // the SIGNED accumulators as int16, when every K and V accumulator of this workgroup's rows lies in [-2^15, 2^15) (tried
  // only where the decoder has a reader for it and the engine's watch says sentences mostly take it: kv_tight_limit > 0);
  // the narrow one, 20 bits per value, when every shifted accumulator lies in [-limit, limit); else 24 bits. The smallest
  // form allowed is tried first; an accumulator that does not fit (seen by its lane while the tile is staged) raises its
  // bit of kv_wide_flag (1: not tight, 2: not narrow), every thread reads the flag behind the staging barrier, and the
  // layer's K and V are produced again in the smallest form that holds them, over whatever the attempt wrote. All forms
  // hold the same integers.
  const bool try_narrow = a.kv24 && a.kv_fmt != nullptr;
  for (int l = 0; l < a.Ld; ++l) {
    const bool try_tight = try_narrow && a.kv_tight_limit > 0 && ((a.kv_tight_layers >> l) & 1u);
    int form = !try_narrow ? 1 : try_tight ? 2 : 0;  // kv_fmt's codes
    for (int attempt = 0; attempt < 3; ++attempt) {
      const bool wide = form == 1;
      bool redo = false;
      for (int p = 0; p < 2; ++p) {
        SLIMT_TPHASE_LANE;
        const PreparedWeight &w = p == 0 ? a.dec_k[l] : a.dec_v[l];
        float *out = a.kv + (size_t)(2 * l + p) * B * S * D;
        v4i wf[KSD];
        load_w(wf, w, wave, lane);
        const TEpi e = tload_epi(w, wave, lg);
        // (packed cache: no barrier here -- every wave's reads of the A buffer lie in front of the barrier between the products and
        // the packing loop of the projection before, or of FFN1's; the staging tile is not written until the barrier below)
        if (!a.kv24) lds_barrier();  // the A buffer and the region are free
        if (owner) quantise_x(0, w.a_quant, tq_base(lane));
        // (the flag is raised behind the barrier below only)
        if (!wide && p == 0) {
          kvf ^= 1;
          if (tid == 0) kv_wide_flag[kvf] = 0;
        }
        lds_barrier();
        const int col = wave * 16 + lg * 4;
        const AFrag af = a_frag(Aq, lane);
        const v4i skv = tshift(e);
        int *stg = reinterpret_cast<int *>(region);  // packed cache: [TR][LDY] shifted accumulators (tight: the signed ones)
        unsigned outside = 0;  // an accumulator of a valid row outside the form's range
        const unsigned lim = (unsigned)a.kv_narrow_limit, lim16 = (unsigned)a.kv_tight_limit;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          const v4i c = mma_rt(wf, rt, af, skv);
          const int rrow = 16 * rt + lr;
          if (a.kv24) {
            if (form == 2) {
              const v4i sg = c - *reinterpret_cast<const v4i *>(a.kv_centre[l][p] + col);  // less the columns' centres
              *reinterpret_cast<v4i *>(stg + rrow * LDY + col) = sg;
              if (row_valid(rrow)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  outside |= (unsigned)((unsigned)sg[i] + lim16 >= 2u * lim16) | (unsigned)((unsigned)c[i] + lim >= 2u * lim) << 1;
#ifdef SLIMT_EXP_TIGHT_ALL  // timing only (wrong results): every sentence-layer takes the tight form
                outside &= 2u;
#endif
              }
            } else {
              *reinterpret_cast<v4i *>(stg + rrow * LDY + col) = c;
              if (!wide && row_valid(rrow)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) outside |= (unsigned)((unsigned)c[i] + lim >= 2u * lim) << 1;
              }
            }
          } else if (row_valid(rrow)) {
            // c is the shifted accumulator accS: the cache holds float(accS), exact (kernels.h, kv24)
            const float4 v = {(float)c[0], (float)c[1], (float)c[2], (float)c[3]};
            if (p == 0) {  // K cache layout [sentence][head][d/4][key][4]: the lane's 4 columns are one d/4 group
              const size_t chunk = (size_t)row_sentence(rrow) * (D / 4) + (col >> 2);
              *reinterpret_cast<float4 *>(out + (chunk * S + rrow % S) * 4) = v;
            } else {
              *reinterpret_cast<float4 *>(out + ((size_t)row_sentence(rrow) * S + rrow % S) * D + col) = v;
            }
          }
        }
        if (!a.kv24) continue;
        if (outside) __hip_atomic_fetch_or(&kv_wide_flag[kvf], (int)outside, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_barrier();
        // (uniform: ONE read by every thread behind the barrier. A tight attempt looks at both ranges at once to know where
        // to go if it fails; when it holds, the shifted accumulators' range does not matter)
        const int raised = wide ? 0 : kv_wide_flag[kvf];
        if (raised & (form == 2 ? 1 : 2)) {
          redo = true;
          form = (raised & 2) ? 1 : 0;
          break;
        }
        const int Sp = (S + 3) & ~3;
        if (form == 2) {
          // the tight form (decode_attention_packed.inl.h, attention_packed32 with Form16): one thread = 32 values = four quads of int16
          if (p == 0) {  // K [sentence][head][plane 0..3][key][16 B]: consecutive lanes = consecutive keys
            const rsrc_t ro = trsrc(out, (unsigned)((size_t)B * S * D * 3));
            for (int it = tid; it < TR * (D / 32); it += 1024) {
              const int r = it % TR, h = it / TR;
              if (!row_valid(r)) continue;
              const int off = row_sentence(r) * S * D * 3 + (h * 4 * S + r % S) * 16;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const v4i pk = pack16(*reinterpret_cast<const v4i *>(stg + r * LDY + 32 * h + 8 * q),
                                      *reinterpret_cast<const v4i *>(stg + r * LDY + 32 * h + 8 * q + 4));
                if (a.kv_store_nt) __builtin_amdgcn_raw_buffer_store_b128(pk, ro, off + q * S * 16, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(pk, ro, off + q * S * 16, 0, 0);
              }
            }
          } else {  // V [sentence][key / 8][plane 0..3][column / 4][16 B]: key pairs x 4 columns, key-major
            const rsrc_t ro = trsrc(out, (unsigned)((size_t)B * Sp * D * 3));
            const int G = (S + 7) >> 3;
            for (int it = tid; it < spw * G * 64; it += 1024) {
              const int cl = it & 63, g = (it >> 6) % G, si = (it >> 6) / G;
              if (s0 + si >= B) continue;
              const int off = (s0 + si) * Sp * D * 3 + (g * 4 * 64 + cl) * 16;
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // keys past the sentence: acc = -127 colsum, accS = 0 (finite, weight 0)
                const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
                const v4i z = {0, 0, 0, 0};
                const v4i x0 = *reinterpret_cast<const v4i *>(stg + (si * S + (k0 < S ? k0 : 0)) * LDY + 4 * cl);
                const v4i x1 = *reinterpret_cast<const v4i *>(stg + (si * S + (k1 < S ? k1 : 0)) * LDY + 4 * cl);
                const v4i pk = pack16(k0 < S ? x0 : z, k1 < S ? x1 : z);
                if (a.kv_store_nt) __builtin_amdgcn_raw_buffer_store_b128(pk, ro, off + q * 1024, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(pk, ro, off + q * 1024, 0, 0);
              }
            }
          }
          continue;
        }
        if (!wide) {
          // the narrow form (decode_attention_packed.inl.h, attention_packed32 with Form20): one thread = 32 values = four quads of hi halves + one
          // quad of lo nibbles
          if (p == 0) {  // K [sentence][head][plane 0..4][key][16 B]: consecutive lanes = consecutive keys
            const rsrc_t ro = trsrc(out, (unsigned)((size_t)B * S * D * 3));
            for (int it = tid; it < TR * (D / 32); it += 1024) {
              const int r = it % TR, h = it / TR;
              if (!row_valid(r)) continue;
              const int off = row_sentence(r) * S * D * 3 + (h * 5 * S + r % S) * 16;
              int lo[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const Packed20 pk = pack20(*reinterpret_cast<const v4i *>(stg + r * LDY + 32 * h + 8 * q),
                                           *reinterpret_cast<const v4i *>(stg + r * LDY + 32 * h + 8 * q + 4));
                lo[q] = pk.lo;
                if (a.kv_store_nt) __builtin_amdgcn_raw_buffer_store_b128(pk.hi, ro, off + q * S * 16, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(pk.hi, ro, off + q * S * 16, 0, 0);
              }
              const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
              if (a.kv_store_nt) __builtin_amdgcn_raw_buffer_store_b128(lq, ro, off + 4 * S * 16, 0, 2);
              else __builtin_amdgcn_raw_buffer_store_b128(lq, ro, off + 4 * S * 16, 0, 0);
            }
          } else {  // V [sentence][key / 8][plane 0..4][column / 4][16 B]: key pairs x 4 columns, key-major
            const rsrc_t ro = trsrc(out, (unsigned)((size_t)B * Sp * D * 3));
            const int G = (S + 7) >> 3;
            for (int it = tid; it < spw * G * 64; it += 1024) {
              const int cl = it & 63, g = (it >> 6) % G, si = (it >> 6) / G;
              if (s0 + si >= B) continue;
              const int off = (s0 + si) * Sp * D * 3 + (g * 5 * 64 + cl) * 16;
              int lo[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
                const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
                const v4i z = {0, 0, 0, 0};
                const v4i x0 = *reinterpret_cast<const v4i *>(stg + (si * S + (k0 < S ? k0 : 0)) * LDY + 4 * cl);
                const v4i x1 = *reinterpret_cast<const v4i *>(stg + (si * S + (k1 < S ? k1 : 0)) * LDY + 4 * cl);
                const Packed20 pk = pack20(k0 < S ? x0 : z, k1 < S ? x1 : z);
                lo[q] = pk.lo;
                if (a.kv_store_nt) __builtin_amdgcn_raw_buffer_store_b128(pk.hi, ro, off + q * 1024, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(pk.hi, ro, off + q * 1024, 0, 0);
              }
              const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
              if (a.kv_store_nt) __builtin_amdgcn_raw_buffer_store_b128(lq, ro, off + 4 * 1024, 0, 2);
              else __builtin_amdgcn_raw_buffer_store_b128(lq, ro, off + 4 * 1024, 0, 0);
            }
          }
          continue;
        }
        // the 24-bit form (kernels.h, FusedDecodeArgs::kv24): one thread = 16 values = 48 bytes =
        // three 16-byte stores, one per plane
        if (p == 0) {  // K [sentence][column / 16][plane][key][16 B]: consecutive lanes = consecutive keys
          const rsrc_t ro = trsrc(out, (unsigned)((size_t)B * S * D * 3));
          for (int it = tid; it < TR * (D / 16); it += 1024) {
            const int r = it % TR, ci = it / TR;
            if (!row_valid(r)) continue;
            v3i wd[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wd[g] = tpack24(*reinterpret_cast<const v4i *>(stg + r * LDY + 16 * ci + 4 * g));
            const int off = row_sentence(r) * S * D * 3 + (ci * 3 * S + r % S) * 16;
            const v4i p0 = {wd[0].x, wd[0].y, wd[0].z, wd[1].x}, p1 = {wd[1].y, wd[1].z, wd[2].x, wd[2].y},
                      p2 = {wd[2].z, wd[3].x, wd[3].y, wd[3].z};
            if (a.kv_store_nt) {  // this batch's decoder will stream its cache past the Infinity Cache: so do the stores
              __builtin_amdgcn_raw_buffer_store_b128(p0, ro, off, 0, 2);
              __builtin_amdgcn_raw_buffer_store_b128(p1, ro, off + S * 16, 0, 2);
              __builtin_amdgcn_raw_buffer_store_b128(p2, ro, off + 2 * S * 16, 0, 2);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(p0, ro, off, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(p1, ro, off + S * 16, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(p2, ro, off + 2 * S * 16, 0, 0);
            }
          }
        } else {  // V [sentence][key / 4][plane][column / 4][16 B]: 4 keys x 4 columns, key-major
          const rsrc_t ro = trsrc(out, (unsigned)((size_t)B * Sp * D * 3));
          for (int it = tid; it < spw * (Sp / 4) * 64; it += 1024) {
            const int cl = it & 63, g = (it >> 6) % (Sp / 4), si = (it >> 6) / (Sp / 4);
            if (s0 + si >= B) continue;
            v3i wd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
              const int key = 4 * g + i;
              const v4i xv = *reinterpret_cast<const v4i *>(stg + (si * S + (key < S ? key : 0)) * LDY + 4 * cl);
              const v4i z = {0, 0, 0, 0};
              wd[i] = tpack24(key < S ? xv : z);
            }
            const int off = ((s0 + si) * (Sp / 4) + g) * 3 * 64 * 16 + cl * 16;
            const v4i p0 = {wd[0].x, wd[0].y, wd[0].z, wd[1].x}, p1 = {wd[1].y, wd[1].z, wd[2].x, wd[2].y},
                      p2 = {wd[2].z, wd[3].x, wd[3].y, wd[3].z};
            if (a.kv_store_nt) {
              __builtin_amdgcn_raw_buffer_store_b128(p0, ro, off, 0, 2);
              __builtin_amdgcn_raw_buffer_store_b128(p1, ro, off + 1024, 0, 2);
              __builtin_amdgcn_raw_buffer_store_b128(p2, ro, off + 2048, 0, 2);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(p0, ro, off, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(p1, ro, off + 1024, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(p2, ro, off + 2048, 0, 0);
            }
          }
        }
      }
      if (!redo) break;
    }
    if (a.kv_fmt && a.kv24 && tid < spw && s0 + tid < B) {
      a.kv_fmt[(size_t)l * B + s0 + tid] = (unsigned char)form;
      if (form == 1 && a.kv_wide_count) __hip_atomic_fetch_add(a.kv_wide_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (try_tight && form != 2 && a.kv_not16_count)
        __hip_atomic_fetch_add(a.kv_not16_count + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (gen_here) {
    // the shortlist of this launch: wait for its publisher (running since before this workgroup started, and
    // waiting for nobody: ~40 us of work against this workgroup's ~250); then this workgroup's share of the output layer
    if (shortlists_await_in_launch(a, tid))  // (never published: nothing to pack from)
      pack_weight_share(a, tile, n_wg, tid, 1024);
  }
  if (tid == 0) occ_trace_event(a.trace, 2, 1);
}

size_t tall_encode_lds_bytes(int F) {
  const size_t region_qkv = (size_t)TR * (2 * (128 + 2) + (128 + 16)) * 4;
  const size_t region_h = (size_t)TR * (F + 32);
  const size_t region = region_qkv > region_h ? region_qkv : region_h;
  return 3 * (size_t)TR * 256 + (size_t)TR * (128 + 16) + region;
}

bool tall_encode_supported(int D, int F, int H, int Le, int Ld, int S) {
  if (S < 1 || S > TR || Le < 1 || Le > 6 || Ld < 1 || Ld > 4) return false;  // S <= 32: several sentences per
  if (D != 256 || H != 8) return false;                                      // workgroup; 33..64: one
  if (F != 1536 && F != 1024) return false;  // (F = 2048: the hidden layer of 64 rows does not fit the region)
  return tall_encode_lds_bytes(F) <= 160 * 1024;
}

// workgroups launched for 64-row tiles (see fused_encode_grid)
int tall_encode_grid(int B, int S, bool tickets) {
  const int spw = TR / S;
  const int tiles = (B + spw - 1) / spw;
  if (!tickets) return tiles;
  const int extra = tiles / 4 > 32 ? tiles / 4 : 32;
  return (tiles + extra + 31) / 32 * 32;
}

hipError_t launch_encode_tall(const FusedEncodeArgs &a, int F, hipStream_t st) {
  // in-launch shortlist generation: tiles must be claimed in START order (the publisher is then running before
  // any waiter exists), and the two bitmaps + the scan array must fit the LDS the encoder is not using yet
  if (a.gen.w2o && (!a.ticket || !a.gen_flag || !a.pack_tiles ||
                    shortlist_in_launch_lds_bytes(a.gen.src_vocab, a.gen.tgt_vocab) > tall_encode_lds_bytes(F)))
    return hipErrorInvalidValue;
  const dim3 grid(tall_encode_grid(a.B, a.S, a.ticket != nullptr));
  const size_t lds = tall_encode_lds_bytes(F);
  hipError_t e = hipSuccess;
  static const bool no_row_tile_skip = std::getenv("SLIMT_ENC_ROW_TILES") && std::getenv("SLIMT_ENC_ROW_TILES")[0] == '4';  // A/B
#define SLIMT_TALL_CASE(KSF_)                                                                  \
  if (F == 64 * KSF_) {                                                                        \
    /* three row tiles where the workgroup's sentences end within 48 rows (S = 22..24, 33..48) */ \
    const bool three = (64 / a.S) * a.S <= 48 && !no_row_tile_skip;                             \
    auto k = a.S > 32 ? (three ? encode_tall_kernel<KSF_, 4, 3> : encode_tall_kernel<KSF_, 4>)  \
                      : (three ? encode_tall_kernel<KSF_, 2, 3> : encode_tall_kernel<KSF_, 2>); \
    e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds);             \
    if (e != hipSuccess) return e;                                                             \
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, st, a);                                       \
    return hipGetLastError();                                                                  \
  }
  SLIMT_TALL_CASE(24) SLIMT_TALL_CASE(16)
#undef SLIMT_TALL_CASE
  return hipErrorInvalidValue;
}

}  // namespace slimt_hip
