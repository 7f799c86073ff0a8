// Persistent fused encoder: embedding + all encoder layers + the decoder's
// cross-attention K/V cache in ONE launch (Model.cc:195-201,
// Transformer.cc:57-69, Modules.cc:287-334).
//
// Encoder rows only interact inside a sentence (self-attention), so one
// workgroup owns 32 rows = floor(32 / S) whole sentences (S <= 32) for all
// layers: the residual stream (f32), the three int8 A operands of Q/K/V and
// the f32 q/k/v of all heads live in LDS (~155 KiB of the CU's 160 KiB);
// weights stream from L2 in MFMA-fragment order. Arithmetic is bit-identical
// to the layer-by-layer kernels (kernels.hip).
#include "device_common.h"
#include "shortlist_device.h"
#include "kernels.h"

namespace slimt_hip {

namespace {

constexpr int ENW = 16;  // waves per workgroup
constexpr int ER = 32;   // rows per workgroup (2 MFMA row tiles)

#define SLIMT_GLOBAL __attribute__((address_space(1)))
#define SLIMT_LDS __attribute__((address_space(3)))
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float edequant(int acc, int colsum, float u, float pb) {
  const float v = (float)(acc + __mul24(127, colsum)) * u;  // |colsum| <= 127 K < 2^23
  return v + pb;
}

// four 24-bit two's-complement integers, little endian, in 12 bytes (the packed K/V cache)
typedef int v3i __attribute__((ext_vector_type(3)));
__device__ __forceinline__ v3i pack24(v4i x) {
  v3i o;
  o.x = (x.x & 0xffffff) | (x.y << 24);
  o.y = ((x.y >> 8) & 0xffff) | (x.z << 16);
  o.z = ((x.z >> 16) & 0xff) | (x.w << 8);
  return o;
}

// one 16-column tile of a weight against both row tiles of A (32 x K int8 in LDS)
template <int KS>
__device__ __forceinline__ void tile_mma2(const char *A, int lda, const v4i (&bf)[KS], int lr, int lg,
                                          v4i &acc0, v4i &acc1, int ks0 = 0) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const v4i a0 = *reinterpret_cast<const v4i *>(A + lr * lda + (ks0 + ks) * 64 + lg * 16);
    const v4i a1 = *reinterpret_cast<const v4i *>(A + (16 + lr) * lda + (ks0 + ks) * 64 + lg * 16);
    acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bf[ks], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bf[ks], acc1, 0, 0, 0);
  }
}

// Weight streams use buffer loads (see decode_fused.hip): descriptor and tile
// offset are wave-uniform SGPRs, the only vector operand is lane * 16.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <int KS>
__device__ __forceinline__ void load_frags(v4i (&bf)[KS], const PreparedWeight &w, int tile, int ks0,
                                           int lane) {
  const int KST = w.K >> 6;
  const rsrc_t r = make_rsrc(w.Wp, (unsigned)w.n_tiles * KST * 1024u);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    bf[ks] = __builtin_bit_cast(
        v4i, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (tile * KST + ks0 + ks) * 1024, 0));
}

// k-steps [ks0, ks0 + KS) of ONE column tile, through a descriptor that ends with the
// tile: a prefetch past the last k-step returns zeros without touching memory (through
// the whole-matrix descriptor of load_frags it would fetch the next tile's fragments --
// 128 KB of useless traffic per layer at the end of the FFN loop, queued ahead of the
// LayerNorm's loads in the in-order return path).
template <int KS>
__device__ __forceinline__ void load_frags_k(v4i (&bf)[KS], const PreparedWeight &w, int tile, int ks0,
                                             int lane) {
  const int KST = w.K >> 6;
  const rsrc_t r = make_rsrc(reinterpret_cast<const char *>(w.Wp) + (size_t)tile * KST * 1024,
                             (unsigned)KST * 1024u);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    bf[ks] = __builtin_bit_cast(
        v4i, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (ks0 + ks) * 1024, 0));
}

// epilogue constants of column tile `tile` for this lane's column (lane & 15)
__device__ __forceinline__ void load_epi(const PreparedWeight &w, int tile, int lr, int &cs, float &pb) {
  const rsrc_t rc = make_rsrc(w.colsum, (unsigned)w.n_tiles * 64u);
  const rsrc_t rp = make_rsrc(w.pb, (unsigned)w.n_tiles * 64u);
  cs = __builtin_amdgcn_raw_buffer_load_b32(rc, lr * 4, tile * 64, 0);
  pb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, lr * 4, tile * 64, 0));
}

// canonical in-place LayerNorm of LDS row x[0..D) by one wave
template <int DPL>
__device__ __forceinline__ void load_ln(const float *scale, const float *bias, int lane,
                                        float (&sc)[DPL], float (&bi)[DPL]) {
  const rsrc_t rs = make_rsrc(scale, 64u * DPL * 4u), rb = make_rsrc(bias, 64u * DPL * 4u);
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    sc[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, i * 256, 0));
    bi[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, i * 256, 0));
  }
}

// scale / bias of this lane's columns are passed in registers: they are the same
// for every row, and loading them first keeps them ahead of any weight prefetch
// in the (in-order) vector memory queue
template <int DPL>
__device__ __forceinline__ void eln_row(float *x, const float (&scale)[DPL], const float (&bias)[DPL],
                                        float eps, int lane) {
  constexpr int D = 64 * DPL;
  float v[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) v[i] = x[lane + 64 * i];
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
    x[lane + 64 * i] = m + bias[i];
  }
}

}  // namespace

// Diagnostic phase stamps (100 MHz wall clock) of workgroup 0 in one layer.
#define SLIMT_ESTAMP(id)                                                              \
  do {                                                                                \
    if (a.stamps && s0 == 0 && tid == 0 && l == a.stamp_layer)                        \
      a.stamps[(id)] = wall_clock64();                                                \
  } while (0)

// see decode_fused.hip: keeps lane-derived offsets from being hoisted and spilled
#define SLIMT_PHASE_LANE                                 \
  int lane = lane0;                                      \
  asm volatile("" : "+v"(lane));                         \
  const int lr = lane & 15, lg = lane >> 4;              \
  (void)lr;                                              \
  (void)lg

template <int KSD, int KSF, int DH>
__global__ __launch_bounds__(1024) void encode_fused_kernel(FusedEncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 64 * KSD, H = D / DH;
  constexpr int LDX = D + 4;   // f32 residual rows
  constexpr int LDA = D + 16;  // int8 A rows
  // f32 q / k / v rows. The attention's 16x16x4 MFMA operands are read by lanes (row n = lane % 16,
  // k index g = lane / 16): q and k at [row n][d + g] -- stride = 4 mod 64 words puts the 64 lanes
  // on 64 banks --, v at [key g][d + n] -- stride = 16 mod 64 does.
  constexpr int LDQQ = D + 4;
  constexpr int LDK = D + 4;
  constexpr int LDV = D + 16;
  static_assert(D / 16 == ENW, "one 16-column tile of a D-wide GEMM per wave");
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  SLIMT_PHASE_LANE;
  const int S = a.S, B = a.B;
  const int spw = ER / S;              // whole sentences per workgroup
  // Over-subscribed launch (as in decode_fused.hip): the grid holds more workgroups than
  // tiles, the first to START claim the tiles, the rest leave. The tail of an encoder
  // launch then goes to whichever CUs free up first instead of waiting for CUs of the
  // shader engines its last workgroups were bound to at dispatch.
  __shared__ int claimed;
  const int n_tiles = (B + spw - 1) / spw;
  int tile = blockIdx.x;
  if (a.ticket) {
    if (tid == 0) claimed = (int)(atomicAdd(a.ticket, 1u) - a.ticket_base);
    __syncthreads();
    tile = claimed;
    if ((unsigned)tile >= (unsigned)n_tiles) return;
  }
  const int s0 = tile * spw;           // first sentence
  const int rows_used = spw * S;
  if (tid == 0) occ_trace_event(a.trace, 2, 0);
  // this workgroup's sentence lengths: read once (they may live in pinned host memory)
  __shared__ int slens[ER];
  // the narrow cache form does not hold this workgroup's accumulators (kernels.h, kv_fmt); two words used in turn, so that a
  // layer's clearing never meets a lagging wave's read of the layer before (encode_tall.hip)
  __shared__ int kv_wide_flag[2];
  if (tid < spw) slens[tid] = s0 + tid < B ? sentence_length(a, s0 + tid, S) : 0;

  float *xs = reinterpret_cast<float *>(smem);
  char *Aq = reinterpret_cast<char *>(xs + ER * LDX);
  char *Ak = Aq + ER * LDA;
  char *Av = Ak + ER * LDA;
  float *qb = reinterpret_cast<float *>(Av + ER * LDA);
  float *kb = qb + ER * LDQQ;
  float *vb = kb + ER * LDK;

  // row r of this workgroup: sentence s0 + r / S, position r % S
  auto row_sentence = [&](int r) { return s0 + r / S; };
  auto row_valid = [&](int r) { return r < rows_used && row_sentence(r) < B; };

  // Q/K/V weight fragments of this wave's column tile; those of layer l+1 are
  // issued before the last LayerNorm of layer l (they do not depend on data)
  v4i bq[KSD], bk[KSD], bv[KSD];
  load_frags<KSD>(bq, a.L[0].q, wave, 0, lane);
  load_frags<KSD>(bk, a.L[0].k, wave, 0, lane);
  load_frags<KSD>(bv, a.L[0].v, wave, 0, lane);
  // the three int8 A operands of a layer's Q/K/V projections from row r of xs
  // (lane = 4 consecutive columns: one 16-byte read and three 4-byte writes per row
  // instead of 4 + 12 single-element LDS operations; the row was written by this wave)
  static_assert(KSD == 4, "one float4 per lane covers the row");
  auto quantise_row = [&](int r, const FusedEncLayerW &L) {
    const float4 v = *reinterpret_cast<const float4 *>(xs + r * LDX + 4 * lane);
    *reinterpret_cast<int *>(Aq + r * LDA + 4 * lane) =
        pack4(quantize1_byte(v.x, L.q.a_quant), quantize1_byte(v.y, L.q.a_quant), quantize1_byte(v.z, L.q.a_quant),
              quantize1_byte(v.w, L.q.a_quant));
    *reinterpret_cast<int *>(Ak + r * LDA + 4 * lane) =
        pack4(quantize1_byte(v.x, L.k.a_quant), quantize1_byte(v.y, L.k.a_quant), quantize1_byte(v.z, L.k.a_quant),
              quantize1_byte(v.w, L.k.a_quant));
    *reinterpret_cast<int *>(Av + r * LDA + 4 * lane) =
        pack4(quantize1_byte(v.x, L.v.a_quant), quantize1_byte(v.y, L.v.a_quant), quantize1_byte(v.z, L.v.a_quant),
              quantize1_byte(v.w, L.v.a_quant));
  };

  // ---- side job: the batch's shortlisted output layer (used by the decoder
  // launch that follows this one in the stream; independent of the encoder) ----
  const bool gen_here = a.gen.w2o != nullptr;  // the batch's shortlist is generated in this launch (encode_tall.hip): packed at the end
  if (!gen_here) {
    pack_weight_share(a, tile, n_tiles, tid, 1024);
  } else {
    // ShortlistGenerator::generate (Shortlist.cc:115-175; Model.cc:117-120) by the workgroup that started first (a merged
    // launch: by the first n, one shortlist per sub-batch), in the still unused LDS, then published for the others (shortlist_device.h)
    shortlists_publish_in_launch(a, reinterpret_cast<uint32_t *>(smem), tile, n_tiles, tid);
  }

  // ---- embedding (Model.cc:195-197) ----------------------------------------
  for (int r = wave; r < ER; r += ENW) {
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
      xs[r * LDX + lane + 64 * i] = v;
      if (ok && a.embed_out) a.embed_out[((size_t)sb * S + pos) * D + lane + 64 * i] = v;
    }
    quantise_row(r, a.L[0]);  // same wave, same row: no barrier in between
  }
  __syncthreads();

  for (int l = 0; l < a.Le; ++l) {
    SLIMT_PHASE_LANE;
    const FusedEncLayerW &L = a.L[l];
    SLIMT_ESTAMP(0);
    // ---- Attention::forward (Modules.cc:287-319) ---------------------------
    // (x was quantised into Aq/Ak/Av by the producer of xs)
    SLIMT_ESTAMP(1);
    {  // Q, K, V projections: wave = column tile of each
      const int col = wave * 16 + lr;
      {
        v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        tile_mma2<KSD>(Aq, LDA, bq, lr, lg, c0, c1);
        int cs;
        float pb;
        load_epi(L.q, wave, lr, cs, pb);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qb[(lg * 4 + r) * LDQQ + col] = edequant(c0[r], cs, L.q.u, pb);
          qb[(16 + lg * 4 + r) * LDQQ + col] = edequant(c1[r], cs, L.q.u, pb);
        }
      }
      {
        v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        tile_mma2<KSD>(Ak, LDA, bk, lr, lg, c0, c1);
        int cs;
        float pb;
        load_epi(L.k, wave, lr, cs, pb);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          kb[(lg * 4 + r) * LDK + col] = edequant(c0[r], cs, L.k.u, pb);
          kb[(16 + lg * 4 + r) * LDK + col] = edequant(c1[r], cs, L.k.u, pb);
        }
      }
      {
        v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        tile_mma2<KSD>(Av, LDA, bv, lr, lg, c0, c1);
        int cs;
        float pb;
        load_epi(L.v, wave, lr, cs, pb);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vb[(lg * 4 + r) * LDV + col] = edequant(c0[r], cs, L.v.u, pb);
          vb[(16 + lg * 4 + r) * LDV + col] = edequant(c1[r], cs, L.v.u, pb);
        }
      }
    }
    // the O projection's fragments and epilogue constants travel under the attention (bq is
    // dead until the next layer's prefetch)
    load_frags<KSD>(bq, L.o, wave, 0, lane);
    int cs_o;
    float pb_o;
    load_epi(L.o, wave, lr, cs_o, pb_o);
    lds_barrier();
    SLIMT_ESTAMP(2);
    // scaled_dot_product_attention (Modules.cc:24-86) on the f32 matrix cores: one wave per
    // (sentence, head, 16 queries) -- 16 jobs for one 32-token sentence, so all 16 waves work.
    // A chain of v_mfma_f32_16x16x4_f32 over ascending k is bit-identical to the ascending fmaf
    // chain the other kernels and the oracle use, like the 32x32x2 form the stage kernel keeps
    // (tools/probe_mfma_f32.py: 16384 of 16384 elements, magnitudes 1e-38 .. 1e18):
    //   S^T = K Q^T        two key tiles of 16; A lane (n, g) <- K[16 kt + n][d = k0 + g],
    //                      B lane (n, g) <- Q[query n][d = k0 + g]; accumulator register r of lane
    //                      (n, g): key 16 kt + 4 g + r, query n
    //   softmax over keys  the canonical 32-key butterfly: masks 1, 2 inside the four registers,
    //                      mask 4 = lanes g ^ 1 (xor 16), mask 8 = g ^ 2 (xor 32), mask 16 = the
    //                      two key tiles
    //   O = P V            keys 4 s + g per step s: the A operand wants P[query n][4 s + g] on lane
    //                      (n, g), the accumulators hold P[query n][4 g + r] -- a 4 x 4 transpose
    //                      between lane groups and registers (v_permlane16_swap, then 32)
    // Output quantised for the O projection into Aq (dead since the projections).
    {
      typedef float v4f __attribute__((ext_vector_type(4)));
      const int n = lane & 15, g = lane >> 4;
      const float minus_inf = -99999999.0f;  // Input.cc:56-61
      const float lowest = -3.402823466e+38f;
      const int nqh = S > 16 ? 2 : 1;  // 16-query halves of a sentence
      for (int job = wave; job < spw * H * nqh; job += ENW) {
        const int qh = job % nqh, h = (job / nqh) % H, sl = job / (nqh * H);
        const int sb = s0 + sl;
        if (sb >= B) continue;
        const int base = sl * S;
        const int len = slens[sl];
        const int qr = 16 * qh + n;
        const float *qp = qb + (base + (qr < S ? qr : S - 1)) * LDQQ + h * DH + g;
        float sc[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int kr = 16 * kt + n;
          const float *kp = kb + (base + (kr < S ? kr : S - 1)) * LDK + h * DH + g;
          v4f st = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int k0 = 0; k0 < DH; k0 += 4) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[k0], qp[k0], st, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * kt + 4 * g + r;  // key of this register
            float v = st[r];
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
          for (int r = 0; r < 4; ++r) sc[kt][r] = (16 * kt + 4 * g + r) < S ? exp_p_select(sc[kt][r] - mx) : 0.0f;
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
          // T[g][r] -> T[r][g]: registers (0, 1) and (2, 3) trade between lane groups g ^ 1,
          // then the pairs trade between g ^ 2
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
        for (int nt = 0; nt < DH / 16; ++nt) {
          v4f o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int s4 = 0; s4 < 8; ++s4) {  // keys >= S contribute fma(0, v, o) == o
            const int key = 4 * s4 + g;
            const float vv = vb[(base + (key < S ? key : S - 1)) * LDV + h * DH + 16 * nt + n];
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s4 >> 2][s4 & 3], vv, o, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = 16 * qh + 4 * g + r;  // query of this register
            if (q < S) Aq[(base + q) * LDA + h * DH + 16 * nt + n] = (char)quantize1_byte(o[r], L.o.a_quant);
          }
        }
      }
    }
    // rows that belong to no sentence keep a defined A operand
    for (int r = rows_used + wave; r < ER; r += ENW)
#pragma unroll
      for (int i = 0; i < KSD; ++i) Aq[r * LDA + lane + 64 * i] = 0;
    lds_barrier();
    SLIMT_ESTAMP(3);
    float lsc[KSD], lbi[KSD];
    load_ln<KSD>(L.attn_ln_s, L.attn_ln_b, lane, lsc, lbi);  // needed behind the next barrier
    {  // O projection + residual: x = x + yo (Modules.cc:308-314)
      v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      tile_mma2<KSD>(Aq, LDA, bq, lr, lg, c0, c1);
      const int col = wave * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *p0 = xs + (lg * 4 + r) * LDX + col;
        float *p1 = xs + (16 + lg * 4 + r) * LDX + col;
        *p0 = *p0 + edequant(c0[r], cs_o, L.o.u, pb_o);
        *p1 = *p1 + edequant(c1[r], cs_o, L.o.u, pb_o);
      }
    }
    // FFN1's first column tiles: requested here, used behind the LayerNorm
    constexpr int NT1 = (KSF * 4) / ENW;  // FFN1 column tiles per wave
    const rsrc_t r1 = make_rsrc(L.ffn1.Wp, (unsigned)L.ffn1.n_tiles * KSD * 1024u);
    const rsrc_t r1c = make_rsrc(L.ffn1.colsum, (unsigned)L.ffn1.n_tiles * 64u);
    const rsrc_t r1p = make_rsrc(L.ffn1.pb, (unsigned)L.ffn1.n_tiles * 64u);
    v4i bw[3][KSD], cs4[3];
    float4 pb4[3];
    auto load1 = [&](int buf, int i) {
      const int tile = wave + ENW * i;
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks)
        bw[buf][ks] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r1, lane * 16, (tile * KSD + ks) * 1024, 0));
      cs4[buf] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r1c, lg * 16, tile * 64, 0));
      pb4[buf] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r1p, lg * 16, tile * 64, 0));
    };
#pragma unroll
    for (int i = 0; i < 3 && i < NT1; ++i) {
      load1(i, i);
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    SLIMT_ESTAMP(4);
    for (int r = wave; r < ER; r += ENW) {
      eln_row<KSD>(xs + r * LDX, lsc, lbi, a.eps, lane);
      const float4 v = *reinterpret_cast<const float4 *>(xs + r * LDX + 4 * lane);
      *reinterpret_cast<int *>(Aq + r * LDA + 4 * lane) =
          pack4(quantize1_byte(v.x, L.ffn1.a_quant), quantize1_byte(v.y, L.ffn1.a_quant),
                quantize1_byte(v.z, L.ffn1.a_quant), quantize1_byte(v.w, L.ffn1.a_quant));
    }
    lds_barrier();
    SLIMT_ESTAMP(5);
    // ---- FFN (Modules.cc:326-331). The whole hidden layer (32 x F int8, 49 KiB for F = 1536)
    // lives in the dead q/k/v buffers, so the phase is two streams with ONE barrier between
    // them (FFN1: 6 column tiles per wave, three in flight; FFN2: 24 k-steps of this wave's
    // column tile, three chunks of four in flight) instead of a barrier per 256 hidden columns
    // (13 per layer before: every barrier re-aligned the waves and drained the weight stream).
    // The MFMA operands are swapped (weights as A): an accumulator lane holds 4 consecutive
    // columns of one row, so the requantised hidden values leave as one 4-byte LDS store per
    // row tile and the FFN2 result meets the residual as float4.
    {
      constexpr int LDH = 64 * KSF + 32;  // hidden row stride (bytes; + 32: conflict-free FFN2 fragment reads, encode_tall.hip)
      char *Hb = reinterpret_cast<char *>(qb);
      static_assert((size_t)ER * LDH <= (size_t)ER * (LDQQ + LDK + LDV) * 4, "hidden layer fits q/k/v");
      static_assert((KSF * 4) % ENW == 0 && KSF % 4 == 0, "whole tiles / whole chunks per wave");
      {
        v4i a0[KSD], a1[KSD];  // this wave's view of the 32 input rows, all of K
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
          a0[ks] = *reinterpret_cast<const v4i *>(Aq + lr * LDA + ks * 64 + lg * 16);
          a1[ks] = *reinterpret_cast<const v4i *>(Aq + (16 + lr) * LDA + ks * 64 + lg * 16);
        }
#pragma unroll
        for (int i = 0; i < NT1; ++i) {
          const int buf = i % 3;
          v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
          for (int ks = 0; ks < KSD; ++ks) {  // lane: row lr (c0) / 16 + lr (c1), columns 4 lg .. 4 lg + 3 of the tile
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[buf][ks], a0[ks], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[buf][ks], a1[ks], c1, 0, 0, 0);
          }
          const float pbv[4] = {pb4[buf].x, pb4[buf].y, pb4[buf].z, pb4[buf].w};
          int q0[4], q1[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v0 = edequant(c0[r], cs4[buf][r], L.ffn1.u, pbv[r]);
            float v1 = edequant(c1[r], cs4[buf][r], L.ffn1.u, pbv[r]);
            v0 = v0 > 0.0f ? v0 : 0.0f;
            v1 = v1 > 0.0f ? v1 : 0.0f;
            q0[r] = quantize1_byte(v0, L.ffn2.a_quant);
            q1[r] = quantize1_byte(v1, L.ffn2.a_quant);
          }
          const int col = (wave + ENW * i) * 16 + lg * 4;
          *reinterpret_cast<int *>(Hb + lr * LDH + col) = pack4(q0[0], q0[1], q0[2], q0[3]);
          *reinterpret_cast<int *>(Hb + (16 + lr) * LDH + col) = pack4(q1[0], q1[1], q1[2], q1[3]);
          if (i + 3 < NT1) load1(buf, i + 3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // FFN2's first weight chunks do not depend on the hidden layer: request them before the barrier
      const rsrc_t r2 = make_rsrc(reinterpret_cast<const char *>(L.ffn2.Wp) + (size_t)wave * KSF * 1024, (unsigned)KSF * 1024u);
      constexpr int NC2 = KSF / 4;  // chunks of four k-steps
      v4i b2[3][4];
      auto load2 = [&](int buf, int c) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          b2[buf][ks] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r2, lane * 16, (c * 4 + ks) * 1024, 0));
      };
#pragma unroll
      for (int c = 0; c < 3 && c < NC2; ++c) {
        load2(c, c);
        __builtin_amdgcn_sched_barrier(0);
      }
      int cs2i[4];
      float pb2v[4];
      {
        const rsrc_t r2c = make_rsrc(L.ffn2.colsum, (unsigned)L.ffn2.n_tiles * 64u);
        const rsrc_t r2p = make_rsrc(L.ffn2.pb, (unsigned)L.ffn2.n_tiles * 64u);
        const v4i c4 = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r2c, lg * 16, wave * 64, 0));
        const float4 p4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r2p, lg * 16, wave * 64, 0));
        cs2i[0] = c4[0]; cs2i[1] = c4[1]; cs2i[2] = c4[2]; cs2i[3] = c4[3];
        pb2v[0] = p4.x; pb2v[1] = p4.y; pb2v[2] = p4.z; pb2v[3] = p4.w;
      }
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();  // the hidden layer is complete
      v4i f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NC2; ++c) {
        const int buf = c % 3;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const v4i h0 = *reinterpret_cast<const v4i *>(Hb + lr * LDH + (c * 4 + ks) * 64 + lg * 16);
          const v4i h1 = *reinterpret_cast<const v4i *>(Hb + (16 + lr) * LDH + (c * 4 + ks) * 64 + lg * 16);
          f0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b2[buf][ks], h0, f0, 0, 0, 0);
          f1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b2[buf][ks], h1, f1, 0, 0, 0);
        }
        if (c + 3 < NC2) load2(buf, c + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
      load_ln<KSD>(L.ffn_ln_s, L.ffn_ln_b, lane, lsc, lbi);  // needed behind the next barrier
      // x = FFN2(...) + x: this lane's 4 columns of rows lr and 16 + lr
      const int col = wave * 16 + lg * 4;
      float4 *p0 = reinterpret_cast<float4 *>(xs + lr * LDX + col);
      float4 *p1 = reinterpret_cast<float4 *>(xs + (16 + lr) * LDX + col);
      float4 x0 = *p0, x1 = *p1;
      x0.x = edequant(f0[0], cs2i[0], L.ffn2.u, pb2v[0]) + x0.x;
      x0.y = edequant(f0[1], cs2i[1], L.ffn2.u, pb2v[1]) + x0.y;
      x0.z = edequant(f0[2], cs2i[2], L.ffn2.u, pb2v[2]) + x0.z;
      x0.w = edequant(f0[3], cs2i[3], L.ffn2.u, pb2v[3]) + x0.w;
      x1.x = edequant(f1[0], cs2i[0], L.ffn2.u, pb2v[0]) + x1.x;
      x1.y = edequant(f1[1], cs2i[1], L.ffn2.u, pb2v[1]) + x1.y;
      x1.z = edequant(f1[2], cs2i[2], L.ffn2.u, pb2v[2]) + x1.z;
      x1.w = edequant(f1[3], cs2i[3], L.ffn2.u, pb2v[3]) + x1.w;
      *p0 = x0;
      *p1 = x1;
    }
    lds_barrier();
    SLIMT_ESTAMP(6);
    {  // next layer's projection weights, under the LayerNorm (unconditional, so
       // that the registers are dead between the projections and here)
      const FusedEncLayerW &Ln = a.L[l + 1 < a.Le ? l + 1 : l];
      load_frags<KSD>(bq, Ln.q, wave, 0, lane);
      load_frags<KSD>(bk, Ln.k, wave, 0, lane);
      load_frags<KSD>(bv, Ln.v, wave, 0, lane);
    }
    for (int r = wave; r < ER; r += ENW) {
      eln_row<KSD>(xs + r * LDX, lsc, lbi, a.eps, lane);
      if (l + 1 < a.Le) quantise_row(r, a.L[l + 1]);
      if (a.layer_out && row_valid(r)) {
        float *dst = a.layer_out + ((size_t)l * B * S + (size_t)row_sentence(r) * S + r % S) * D;
#pragma unroll
        for (int i = 0; i < KSD; ++i) dst[lane + 64 * i] = xs[r * LDX + lane + 64 * i];
      }
    }
    lds_barrier();
    SLIMT_ESTAMP(7);
  }

  // ---- encoder output + decoder cross-attention K/V (Modules.cc:248-249,
  // computed once per batch instead of every step) ----------------------------
  for (int r = wave; r < ER; r += ENW) {
    if (a.enc_out && row_valid(r)) {
      float *dst = a.enc_out + ((size_t)row_sentence(r) * S + r % S) * D;
#pragma unroll
      for (int i = 0; i < KSD; ++i) dst[lane + 64 * i] = xs[r * LDX + lane + 64 * i];
    }
  }
  for (int l = 0; l < a.Ld; ++l) {
    SLIMT_PHASE_LANE;
    const PreparedWeight &wk = a.dec_k[l], &wv = a.dec_v[l];
    for (int r = wave; r < ER; r += ENW) {
#pragma unroll
      for (int i = 0; i < KSD; ++i) {
        const float v = xs[r * LDX + lane + 64 * i];
        Ak[r * LDA + lane + 64 * i] = (char)quantize1_byte(v, wk.a_quant);
        Av[r * LDA + lane + 64 * i] = (char)quantize1_byte(v, wv.a_quant);
      }
    }
    if (tid == 0) kv_wide_flag[l & 1] = 0;  // (raised behind this barrier, read ONCE behind the staging barrier, by every thread)
    lds_barrier();
    v4i bk[KSD], bv[KSD];
    load_frags<KSD>(bk, wk, wave, 0, lane);
    load_frags<KSD>(bv, wv, wave, 0, lane);
    const int col = wave * 16 + lr;
    float *kout = a.kv + (size_t)(2 * l) * B * S * D;
    float *vout = a.kv + (size_t)(2 * l + 1) * B * S * D;
    if (a.kv24) {
      // Packed cache (kernels.h, FusedEncodeArgs::kv24): the shifted accumulators accS = acc +
      // 127 colsum (|accS| <= 254 * 128 * 256 < 2^23) as 24-bit integers; the decoder rebuilds
      // float(accS) * u + pb in registers. The accumulator tiles cross to the packing threads
      // through the (dead) q / k buffers.
      int *stk = reinterpret_cast<int *>(qb), *stv = reinterpret_cast<int *>(kb);
      static_assert(LDQQ == LDK, "both staging tiles use the q row stride");
      unsigned outside = 0;  // an accumulator of a valid row outside a form's range: bit 0 the tight one's, bit 1 the narrow one's
      const unsigned lim = a.kv_fmt ? (unsigned)a.kv_narrow_limit : 0x40000000u;
      const bool try_tight = a.kv_fmt && a.kv_tight_limit > 0 && ((a.kv_tight_layers >> l) & 1u);
      const unsigned lim16 = try_tight ? (unsigned)a.kv_tight_limit : 0x40000000u;
      {
        v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        tile_mma2<KSD>(Ak, LDA, bk, lr, lg, c0, c1);
        int cs;
        float pb;
        load_epi(wk, wave, lr, cs, pb);
        (void)pb;
        const int ctr = try_tight ? a.kv_centre[l][0][col] : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s0v = c0[r] + __mul24(127, cs), s1v = c1[r] + __mul24(127, cs);
          stk[(lg * 4 + r) * LDQQ + col] = s0v;
          stk[(16 + lg * 4 + r) * LDQQ + col] = s1v;
          if (row_valid(lg * 4 + r))
            outside |= (unsigned)((unsigned)s0v + lim >= 2u * lim) << 1 | (unsigned)((unsigned)(s0v - ctr) + lim16 >= 2u * lim16);
          if (row_valid(16 + lg * 4 + r))
            outside |= (unsigned)((unsigned)s1v + lim >= 2u * lim) << 1 | (unsigned)((unsigned)(s1v - ctr) + lim16 >= 2u * lim16);
        }
      }
      {
        v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        tile_mma2<KSD>(Av, LDA, bv, lr, lg, c0, c1);
        int cs;
        float pb;
        load_epi(wv, wave, lr, cs, pb);
        (void)pb;
        const int ctr = try_tight ? a.kv_centre[l][1][col] : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s0v = c0[r] + __mul24(127, cs), s1v = c1[r] + __mul24(127, cs);
          stv[(lg * 4 + r) * LDQQ + col] = s0v;
          stv[(16 + lg * 4 + r) * LDQQ + col] = s1v;
          if (row_valid(lg * 4 + r))
            outside |= (unsigned)((unsigned)s0v + lim >= 2u * lim) << 1 | (unsigned)((unsigned)(s0v - ctr) + lim16 >= 2u * lim16);
          if (row_valid(16 + lg * 4 + r))
            outside |= (unsigned)((unsigned)s1v + lim >= 2u * lim) << 1 | (unsigned)((unsigned)(s1v - ctr) + lim16 >= 2u * lim16);
        }
      }
      if (outside) __hip_atomic_fetch_or(&kv_wide_flag[l & 1], (int)outside, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      lds_barrier();
      const int raised = kv_wide_flag[l & 1];
      // The form of this workgroup's caches of this layer (kernels.h, FusedDecodeArgs::kv_fmt): the tight one (int16 less the
      // columns' centres) where the engine allows it and every such value of its valid rows lies in [-2^15, 2^15); else the
      // narrow one, 20 bits per value, when every K and V accumulator lies in [-limit, limit); else 24 bits -- both tiles are
      // staged (as accS), so the choice is made before anything is written. The same integers either way.
      const int form = !a.kv_fmt ? 1 : (try_tight && !(raised & 1)) ? 2 : (raised & 2) ? 1 : 0;  // kv_fmt's codes
      const bool wide = form == 1;
      if (a.kv_fmt && tid < spw && s0 + tid < B) {
        a.kv_fmt[(size_t)l * B + s0 + tid] = (unsigned char)form;
        if (wide && a.kv_wide_count) __hip_atomic_fetch_add(a.kv_wide_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (try_tight && form != 2 && a.kv_not16_count)
          __hip_atomic_fetch_add(a.kv_not16_count + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (form == 2) {
        // decode_attention_packed.inl.h, attention_packed32 with Form16: one thread = 32 values = four quads of int16 (accS - centre)
        //   K [sentence][head][plane 0..3][key][16 B],  V [sentence][key / 8][plane 0..3][column / 4][16 B]
        const int Sp = (S + 3) & ~3, G = (S + 7) >> 3;
        const rsrc_t rko = make_rsrc(kout, (unsigned)((size_t)B * S * D * 3));
        const rsrc_t rvo = make_rsrc(vout, (unsigned)((size_t)B * Sp * D * 3));
        for (int it = tid; it < ER * (D / 32); it += 1024) {
          const int r = it % ER, h = it / ER;
          if (!row_valid(r)) continue;
          const int off = row_sentence(r) * S * D * 3 + (h * 4 * S + r % S) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4i ca = *reinterpret_cast<const v4i *>(a.kv_centre[l][0] + 32 * h + 8 * q), cb = *reinterpret_cast<const v4i *>(a.kv_centre[l][0] + 32 * h + 8 * q + 4);
            const v4i pk = pack16(*reinterpret_cast<const v4i *>(stk + r * LDQQ + 32 * h + 8 * q) - ca,
                                  *reinterpret_cast<const v4i *>(stk + r * LDQQ + 32 * h + 8 * q + 4) - cb);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rko, off + q * S * 16, 0, 0);
          }
        }
        for (int it = tid; it < spw * G * 64; it += 1024) {
          const int cl = it & 63, g = (it >> 6) % G, si = (it >> 6) / G;
          if (s0 + si >= B) continue;
          const int off = (s0 + si) * Sp * D * 3 + (g * 4 * 64 + cl) * 16;
          const v4i cc = *reinterpret_cast<const v4i *>(a.kv_centre[l][1] + 4 * cl);
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
            const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
            const v4i z = {0, 0, 0, 0};
            const v4i x0 = *reinterpret_cast<const v4i *>(stv + (si * S + (k0 < S ? k0 : 0)) * LDQQ + 4 * cl) - cc;
            const v4i x1 = *reinterpret_cast<const v4i *>(stv + (si * S + (k1 < S ? k1 : 0)) * LDQQ + 4 * cl) - cc;
            const v4i pk = pack16(k0 < S ? x0 : z, k1 < S ? x1 : z);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rvo, off + q * 1024, 0, 0);
          }
        }
        lds_barrier();
        continue;
      }
      if (!wide) {
        // decode_attention_packed.inl.h, attention_packed32 with Form20: one thread = 32 values = four quads of hi halves + one quad of lo nibbles
        //   K [sentence][head][plane 0..4][key][16 B],  V [sentence][key / 8][plane 0..4][column / 4][16 B]
        const int Sp = (S + 3) & ~3, G = (S + 7) >> 3;
        const rsrc_t rko = make_rsrc(kout, (unsigned)((size_t)B * S * D * 3));
        const rsrc_t rvo = make_rsrc(vout, (unsigned)((size_t)B * Sp * D * 3));
        for (int it = tid; it < ER * (D / 32); it += 1024) {
          const int r = it % ER, h = it / ER;
          if (!row_valid(r)) continue;
          const int off = row_sentence(r) * S * D * 3 + (h * 5 * S + r % S) * 16;
          int lo[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const Packed20 pk = pack20(*reinterpret_cast<const v4i *>(stk + r * LDQQ + 32 * h + 8 * q),
                                       *reinterpret_cast<const v4i *>(stk + r * LDQQ + 32 * h + 8 * q + 4));
            lo[q] = pk.lo;
            __builtin_amdgcn_raw_buffer_store_b128(pk.hi, rko, off + q * S * 16, 0, 0);
          }
          const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
          __builtin_amdgcn_raw_buffer_store_b128(lq, rko, off + 4 * S * 16, 0, 0);
        }
        for (int it = tid; it < spw * G * 64; it += 1024) {
          const int cl = it & 63, g = (it >> 6) % G, si = (it >> 6) / G;
          if (s0 + si >= B) continue;
          const int off = (s0 + si) * Sp * D * 3 + (g * 5 * 64 + cl) * 16;
          int lo[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
            const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
            const v4i z = {0, 0, 0, 0};
            const v4i x0 = *reinterpret_cast<const v4i *>(stv + (si * S + (k0 < S ? k0 : 0)) * LDQQ + 4 * cl);
            const v4i x1 = *reinterpret_cast<const v4i *>(stv + (si * S + (k1 < S ? k1 : 0)) * LDQQ + 4 * cl);
            const Packed20 pk = pack20(k0 < S ? x0 : z, k1 < S ? x1 : z);
            lo[q] = pk.lo;
            __builtin_amdgcn_raw_buffer_store_b128(pk.hi, rvo, off + q * 1024, 0, 0);
          }
          const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
          __builtin_amdgcn_raw_buffer_store_b128(lq, rvo, off + 4 * 1024, 0, 0);
        }
        lds_barrier();
        continue;
      }
      // One thread = 16 values = 48 bytes = three 16-byte stores, one per plane, so that the
      // decoder's 16-byte loads stay contiguous across lanes:
      //   K [sentence][column / 16][plane][key][16 B]      (consecutive lanes = consecutive keys)
      //   V [sentence][key / 4][plane][column / 4][16 B]   (4 keys x 4 columns, key-major;
      //                                                      consecutive lanes = consecutive columns)
      const int Sp = (S + 3) & ~3;
      const rsrc_t rko = make_rsrc(kout, (unsigned)((size_t)B * S * D * 3));
      const rsrc_t rvo = make_rsrc(vout, (unsigned)((size_t)B * Sp * D * 3));
      for (int it = tid; it < ER * (D / 16); it += 1024) {
        const int r = it % ER, ci = it / ER;
        if (!row_valid(r)) continue;
        v3i w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = pack24(*reinterpret_cast<const v4i *>(stk + r * LDQQ + 16 * ci + 4 * g));
        const int off = row_sentence(r) * S * D * 3 + (ci * 3 * S + r % S) * 16;
        const v4i p0 = {w[0].x, w[0].y, w[0].z, w[1].x}, p1 = {w[1].y, w[1].z, w[2].x, w[2].y},
                  p2 = {w[2].z, w[3].x, w[3].y, w[3].z};
        __builtin_amdgcn_raw_buffer_store_b128(p0, rko, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(p1, rko, off + S * 16, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(p2, rko, off + 2 * S * 16, 0, 0);
      }
      for (int it = tid; it < spw * (Sp / 4) * 64; it += 1024) {
        const int cl = it & 63, g = (it >> 6) % (Sp / 4), si = (it >> 6) / (Sp / 4);
        if (s0 + si >= B) continue;
        v3i w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
          const int key = 4 * g + i;
          const v4i x = *reinterpret_cast<const v4i *>(stv + (si * S + (key < S ? key : 0)) * LDQQ + 4 * cl);
          const v4i z = {0, 0, 0, 0};
          w[i] = pack24(key < S ? x : z);
        }
        const int off = ((s0 + si) * (Sp / 4) + g) * 3 * 64 * 16 + cl * 16;
        const v4i p0 = {w[0].x, w[0].y, w[0].z, w[1].x}, p1 = {w[1].y, w[1].z, w[2].x, w[2].y},
                  p2 = {w[2].z, w[3].x, w[3].y, w[3].z};
        __builtin_amdgcn_raw_buffer_store_b128(p0, rvo, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(p1, rvo, off + 1024, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(p2, rvo, off + 2048, 0, 0);
      }
      lds_barrier();
      continue;
    }
    {
      v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      tile_mma2<KSD>(Ak, LDA, bk, lr, lg, c0, c1);
      int cs;
      float pb;
      load_epi(wk, wave, lr, cs, pb);
      const int hh = col / DH, d = col % DH;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = t * 16 + lg * 4 + r;
          if (row_valid(rr)) {  // K cache layout [sentence][head][d/4][key][4]
            const size_t chunk = ((size_t)row_sentence(rr) * H + hh) * (DH / 4) + (d >> 2);
            // the cache holds float(accS), exact; the decoder applies u and pb after its sums (kernels.h, kv24)
            kout[(chunk * S + rr % S) * 4 + (d & 3)] = (float)((t ? c1[r] : c0[r]) + __mul24(127, cs));
          }
        }
    }
    {
      v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      tile_mma2<KSD>(Av, LDA, bv, lr, lg, c0, c1);
      int cs;
      float pb;
      load_epi(wv, wave, lr, cs, pb);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = t * 16 + lg * 4 + r;
          if (row_valid(rr))
            vout[((size_t)row_sentence(rr) * S + rr % S) * D + col] = (float)((t ? c1[r] : c0[r]) + __mul24(127, cs));
        }
    }
    lds_barrier();
  }
  if (gen_here) {
    if (shortlists_await_in_launch(a, tid))  // (never published: nothing to pack from)
      pack_weight_share(a, tile, n_tiles, tid, 1024);
  }
  if (tid == 0) occ_trace_event(a.trace, 2, 1);
}

// =============================================================================
// Sentences of 33..128 tokens: one 16-wave workgroup per sentence.
//
// A sentence longer than 32 rows does not fit the LDS-resident kernel above
// (f32 residual + all-head q/k/v). Here only the int8 MFMA operands live in LDS
// (the quantised rows of the current GEMM, the FFN's hidden chunks, and two
// heads' q/k/v slices at a time for the attention); the f32 tensors of the
// sentence -- residual, pre-LayerNorm sums, q, k, v: 5 x S x 1 KiB -- go through
// the context's scratch buffers, which stay in L2 between the phases of the one
// workgroup that touches them. Tiling as above: wave w owns column tile w of
// every 256-wide output, with up to 8 row tiles of accumulators, so a weight
// fragment is used for up to 128 rows. Same arithmetic, same order, same bits.
namespace {

constexpr int LR = 128;  // rows of LDS operands (a sentence)

// row-wise LayerNorm of a register-held row (the arithmetic of eln_row)
template <int DPL>
__device__ __forceinline__ void ln_regs(float (&v)[DPL], const float (&scale)[DPL],
                                        const float (&bias)[DPL], float eps) {
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

}  // namespace

// NG: key tiles of 32 the attention is compiled for (2: S <= 64, 4: S <= 128)
template <int NG>
__global__ __launch_bounds__(1024) void encode_long16_kernel(LongEncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef float v16f __attribute__((ext_vector_type(16)));
  constexpr int KSD = 4, D = 256, DH = 32, H = D / DH;
  constexpr int LDA = D + 16;   // int8 operand rows
  constexpr int LDH = DH + 1;   // staged q / k / v head slices (f32)
  constexpr int NRT = 2 * NG;  // row tiles at most (32 rows per key tile)
  const FusedEncodeArgs &f = a.f;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  SLIMT_PHASE_LANE;
  const int b = blockIdx.x, S = f.S;
  const int row0 = b * S;
  const int nrt = (S + 15) >> 4;  // row tiles in use (uniform)
  const int len = checked_length(f.lengths[b], S);

  char *A = smem;                                          // [128][272] int8
  char *H0 = A + LR * LDA, *H1 = H0 + LR * LDA;            // FFN hidden chunks
  float *stage = reinterpret_cast<float *>(A + LR * LDA);  // attention: [2 heads][q,k,v][128][33] (aliases H0/H1)
  float *X = a.x + (size_t)row0 * D, *Y = a.y + (size_t)row0 * D;
  float *Qg = a.q + (size_t)row0 * D, *Kg = a.k + (size_t)row0 * D, *Vg = a.v + (size_t)row0 * D;
  // The GEMM epilogues address the sentence's f32 tensors through buffer descriptors: one
  // 32-bit lane offset + a compile-time row offset per access (32 per-lane 64-bit addresses,
  // one per accumulator row, used to be hoisted and spilled), and rows >= S fall outside the
  // descriptor: their stores are dropped, their loads return 0.
  const unsigned sbytes = (unsigned)S * D * 4u;
  const rsrc_t rX = make_rsrc(X, sbytes), rY = make_rsrc(Y, sbytes);
  auto bstore = [](rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
  };
  auto bload = [](rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };

  // side job: the batch's shortlisted output layer (see above)
  for (int tile = blockIdx.x; tile < f.pack_tiles; tile += gridDim.x) pack_weight_tile(f.pack, tile, tid, 1024);

  // quantise the sentence's rows of `src` (global f32) with aq into A
  auto quantise_rows = [&](const float *src, float aq) {
    for (int r = wave; r < 16 * nrt; r += ENW) {
#pragma unroll
      for (int i = 0; i < KSD; ++i) {
        const float v = r < S ? src[(size_t)r * D + lane + 64 * i] : 0.0f;
        A[r * LDA + lane + 64 * i] = (char)quantize1_byte(v, aq);
      }
    }
  };
  // acc[rt] += A(row tile rt) x fragments of this wave's column tile
  auto mma_rows = [&](const char *Aop, const v4i(&bf)[KSD], v4i(&acc)[NRT]) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      if (rt < nrt) {
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
          const v4i av = *reinterpret_cast<const v4i *>(Aop + (16 * rt + lr) * LDA + ks * 64 + lg * 16);
          acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bf[ks], acc[rt], 0, 0, 0);
        }
      }
    }
  };
  auto zero_acc = [&](v4i(&acc)[NRT]) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) acc[rt] = v4i{0, 0, 0, 0};
  };

  // ---- embedding (Model.cc:195-197) ----------------------------------------
  for (int r = wave; r < S; r += ENW) {
    const uint32_t tok = embed_row(f.emb, f.ids[(size_t)row0 + r]);
#pragma unroll
    for (int i = 0; i < KSD; ++i) {
      const float e = (float)f.emb.wemb[(size_t)tok * D + lane + 64 * i] * f.emb.inv_mult;
      const float sc = e * f.emb.sqrt_d;
      const float v = sc + f.emb.pos[(size_t)r * D + lane + 64 * i];
      X[(size_t)r * D + lane + 64 * i] = v;
      if (f.embed_out) f.embed_out[((size_t)row0 + r) * D + lane + 64 * i] = v;
    }
  }
  __syncthreads();

  for (int l = 0; l < f.Le; ++l) {
    SLIMT_PHASE_LANE;
    const FusedEncLayerW &L = f.L[l];
    const int col = wave * 16 + lr;
    // ---- Q, K, V projections (Modules.cc:287-300) -> global f32 ----------------
    for (int which = 0; which < 3; ++which) {
      const PreparedWeight &W = which == 0 ? L.q : which == 1 ? L.k : L.v;
      const rsrc_t rd = make_rsrc(which == 0 ? Qg : which == 1 ? Kg : Vg, sbytes);
      const int voff = (lg * 4 * D + col) * 4;
      v4i bf[KSD];
      load_frags<KSD>(bf, W, wave, 0, lane);
      int cs;
      float pb;
      load_epi(W, wave, lr, cs, pb);
      quantise_rows(X, W.a_quant);
      __syncthreads();
      v4i acc[NRT];
      zero_acc(acc);
      mma_rows(A, bf, acc);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        if (rt < nrt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bstore(rd, voff, (16 * rt + r) * D * 4, edequant(acc[rt][r], cs, W.u, pb));
        }
      __syncthreads();
    }
    // ---- scaled_dot_product_attention (Modules.cc:24-86), two heads staged at a
    // time, a wave per (head, 32-query tile); f32 MFMA chains as above, four key
    // tiles (see attention_mfma_long in kernels.hip). Output -> A, quantised for
    // the O projection.
    {
      const int n = lane & 31, hh = lane >> 5;
      const int ng = (S + 31) >> 5;
      const float minus_inf = -99999999.0f;  // Input.cc:56-61
      const float lowest = -3.402823466e+38f;
      auto tree32 = [&](const float(&x)[16], auto op, auto op_halves) -> float {
        float t4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) t4[g] = op(op(x[4 * g], x[4 * g + 1]), op(x[4 * g + 2], x[4 * g + 3]));
#pragma unroll
        for (int g = 0; g < 4; ++g) t4[g] = op_halves(t4[g]);
        return op(op(t4[0], t4[1]), op(t4[2], t4[3]));
      };
      auto fadd = [](float x, float y) { return x + y; };
      auto fmax_ = [](float x, float y) { return fmaxf(x, y); };
      auto add_halves = [](float x) { return bf_add<32>(x); };
      auto max_halves = [](float x) { return bf_max<32>(x); };
      for (int hp = 0; hp < H / 2; ++hp) {
        // stage q / k / v of heads 2 hp, 2 hp + 1: [slot][matrix][row][33]
        for (int i = tid; i < S * 64; i += 1024) {
          const int j = i >> 6, c = i & 63;  // c: 2 heads x 32 dims, contiguous in the row
          const size_t src = (size_t)j * D + hp * 64 + c;
          float *dst = stage + ((c >> 5) * 3) * LR * LDH + j * LDH + (c & 31);
          dst[0] = Qg[src];
          dst[LR * LDH] = Kg[src];
          dst[2 * LR * LDH] = Vg[src];
        }
        __syncthreads();
        const int slot = wave & 1, qt = wave >> 1;  // job of this wave
        if (qt * 32 < S) {
          const int h = 2 * hp + slot;
          const float *Qs = stage + (slot * 3) * LR * LDH, *Ks = Qs + LR * LDH, *Vs = Ks + LR * LDH;
          // opaque per head pair: otherwise the 64 clamped V-row offsets and the 128 key masks are
          // hoisted out of the head loop as loop invariants and spilled (VGPRs and SGPRs)
          int nv = n, hv = hh;
          asm volatile("" : "+v"(nv), "+v"(hv));
          const int qrow = (qt * 32 + nv) < S ? (qt * 32 + nv) : S - 1;
          // The scores of a key tile are recomputed in each of the three passes (max, sum,
          // P V) instead of being held for all NG tiles: 16 MFMAs per tile and pass are cheap,
          // 64 more live registers under the 128 of a 16-wave workgroup are not (they spilled).
          // Same values every time, so the results are those of the one-pass form.
          auto scores = [&](int g, float(&sc)[16]) {
            v16f st = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            const int krow = (32 * g + nv) < S ? (32 * g + nv) : S - 1;
#pragma unroll
            for (int k0 = 0; k0 < DH; k0 += 2)
              st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[krow * LDH + k0 + hv], Qs[qrow * LDH + k0 + hv], st, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = 32 * g + 8 * (r >> 2) + 4 * hv + (r & 3);
              float v = st[r];
              v = f.alpha * v;  // (alpha == 1: the product is v, bit for bit)
              v = v + (1.0f - (key < len ? 1.0f : 0.0f)) * minus_inf;
              if (key >= S) v = lowest;
              sc[r] = v;
            }
          };
          float m;
          {
            float mx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[r] = lowest;
#pragma unroll
            for (int g = 0; g < NG; ++g)
              if (g < ng) {
                float sc[16];
                scores(g, sc);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx[r] = fmaxf(mx[r], sc[r]);
              }
            m = tree32(mx, fmax_, max_halves);
          }
          float sum;
          {
            float u[2][16];  // keys L, L + 64 | keys L + 32, L + 96 (0 + e and e + 0 are exact)
#pragma unroll
            for (int r = 0; r < 16; ++r) u[0][r] = u[1][r] = 0.0f;
#pragma unroll
            for (int g = 0; g < NG; ++g)
              if (g < ng) {
                float sc[16];
                scores(g, sc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const int key = 32 * g + 8 * (r >> 2) + 4 * hv + (r & 3);
                  const float e = key < S ? exp_p_select(sc[r] - m) : 0.0f;
                  u[g & 1][r] = u[g & 1][r] + e;
                }
              }
            const float t0 = tree32(u[0], fadd, add_halves);
            sum = t0 + tree32(u[1], fadd, add_halves);
          }
          v16f o = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            if (g < ng) {
              float sc[16], pa[16];
              scores(g, sc);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int key = 32 * g + 8 * (r >> 2) + 4 * hv + (r & 3);
                const float e = key < S ? exp_p_select(sc[r] - m) : 0.0f;
                sc[r] = e / sum;  // keys >= S: exactly 0
              }
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const slimt_u2 s01 = __builtin_amdgcn_permlane32_swap(__float_as_int(sc[4 * q4 + 0]),
                                                                      __float_as_int(sc[4 * q4 + 1]), false, false);
                const slimt_u2 s23 = __builtin_amdgcn_permlane32_swap(__float_as_int(sc[4 * q4 + 2]),
                                                                      __float_as_int(sc[4 * q4 + 3]), false, false);
                pa[4 * q4 + 0] = __int_as_float(s01.x);
                pa[4 * q4 + 1] = __int_as_float(s23.x);
                pa[4 * q4 + 2] = __int_as_float(s01.y);
                pa[4 * q4 + 3] = __int_as_float(s23.y);
              }
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int key = 32 * g + 2 * i + hv;
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[i], Vs[(key < S ? key : S - 1) * LDH + nv], o, 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qi = qt * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
            if (qi < S) A[qi * LDA + h * DH + n] = (char)quantize1_byte(o[r], L.o.a_quant);
          }
        }
        __syncthreads();
      }
      // rows past the sentence in the last row tile keep a defined operand
      for (int r = S + wave; r < 16 * nrt; r += ENW)
#pragma unroll
        for (int i = 0; i < KSD; ++i) A[r * LDA + lane + 64 * i] = 0;
      __syncthreads();
    }
    float lsc[KSD], lbi[KSD];
    {  // O projection + residual (Modules.cc:308-314): Y = X + O(att)
      v4i bf[KSD];
      load_frags<KSD>(bf, L.o, wave, 0, lane);
      int cs;
      float pb;
      load_epi(L.o, wave, lr, cs, pb);
      v4i acc[NRT];
      zero_acc(acc);
      mma_rows(A, bf, acc);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        if (rt < nrt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int voff = (lg * 4 * D + col) * 4, soff = (16 * rt + r) * D * 4;
            bstore(rY, voff, soff, bload(rX, voff, soff) + edequant(acc[rt][r], cs, L.o.u, pb));
          }
        }
    }
    __syncthreads();
    // LayerNorm -> X (the FFN's residual source), quantised for FFN1 into A
    load_ln<KSD>(L.attn_ln_s, L.attn_ln_b, lane, lsc, lbi);
    for (int r = wave; r < 16 * nrt; r += ENW) {
      float v[KSD];
#pragma unroll
      for (int i = 0; i < KSD; ++i) v[i] = r < S ? Y[(size_t)r * D + lane + 64 * i] : 0.0f;
      ln_regs<KSD>(v, lsc, lbi, f.eps);
#pragma unroll
      for (int i = 0; i < KSD; ++i) {
        if (r < S) X[(size_t)r * D + lane + 64 * i] = v[i];
        A[r * LDA + lane + 64 * i] = r < S ? (char)quantize1_byte(v[i], L.ffn1.a_quant) : (char)0;
      }
    }
    __syncthreads();
    // ---- FFN (Modules.cc:326-331): hidden columns in chunks of 256 ---------------
    {
      const int NC = a.F / 256;
      const int KSF = a.F / 64;
      v4i f2[NRT];
      zero_acc(f2);
      v4i b1[KSD], b2[4];
      int cs1;
      float pb1;
      auto ffn1_chunk = [&](int fc, char *Hb) {
        v4i c1[NRT];
        zero_acc(c1);
        mma_rows(A, b1, c1);
        const int cs = cs1;
        const float pb = pb1;
        // fragments consumed: request the next chunk's (reads past the last tile return zeros)
        load_frags<KSD>(b1, L.ffn1, (fc + 1) * 16 + wave, 0, lane);
        load_epi(L.ffn1, (fc + 1) * 16 + wave, lr, cs1, pb1);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (rt < nrt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = edequant(c1[rt][r], cs, L.ffn1.u, pb);
              v = v > 0.0f ? v : 0.0f;
              Hb[(16 * rt + lg * 4 + r) * LDA + wave * 16 + lr] = (char)quantize1_byte(v, L.ffn2.a_quant);
            }
          }
      };
      load_frags<KSD>(b1, L.ffn1, wave, 0, lane);
      load_epi(L.ffn1, wave, lr, cs1, pb1);
      load_frags<4>(b2, L.ffn2, wave, 0, lane);
      ffn1_chunk(0, H0);
      __syncthreads();
      for (int fc = 0; fc < NC; ++fc) {
        char *Hcur = (fc & 1) ? H1 : H0;
        char *Hnext = (fc & 1) ? H0 : H1;
        if (fc + 1 < NC) ffn1_chunk(fc + 1, Hnext);
        mma_rows(Hcur, b2, f2);
        {  // next k-chunk of this wave's FFN2 tile (past the last: clamped, unused)
          const int nk = (fc + 1 < NC ? fc + 1 : fc) * 4;
          (void)KSF;
          load_frags<4>(b2, L.ffn2, wave, nk, lane);
        }
        __syncthreads();
      }
      int cs;
      float pb;
      load_epi(L.ffn2, wave, lr, cs, pb);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        if (rt < nrt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int voff = (lg * 4 * D + col) * 4, soff = (16 * rt + r) * D * 4;
            bstore(rY, voff, soff, edequant(f2[rt][r], cs, L.ffn2.u, pb) + bload(rX, voff, soff));
          }
        }
    }
    __syncthreads();
    load_ln<KSD>(L.ffn_ln_s, L.ffn_ln_b, lane, lsc, lbi);
    for (int r = wave; r < S; r += ENW) {
      float v[KSD];
#pragma unroll
      for (int i = 0; i < KSD; ++i) v[i] = Y[(size_t)r * D + lane + 64 * i];
      ln_regs<KSD>(v, lsc, lbi, f.eps);
#pragma unroll
      for (int i = 0; i < KSD; ++i) {
        X[(size_t)r * D + lane + 64 * i] = v[i];
        if (f.layer_out) f.layer_out[((size_t)l * f.B * S + row0 + r) * D + lane + 64 * i] = v[i];
      }
    }
    __syncthreads();
  }

  // ---- encoder output + decoder cross-attention K/V (Modules.cc:248-249) ---------
  if (f.enc_out && f.enc_out != a.x)
    for (int r = wave; r < S; r += ENW)
#pragma unroll
      for (int i = 0; i < KSD; ++i) f.enc_out[((size_t)row0 + r) * D + lane + 64 * i] = X[(size_t)r * D + lane + 64 * i];
  const size_t M = (size_t)f.B * S;
  // The packed cache of this sentence takes one of three forms per layer (kernels.h, FusedDecodeArgs::kv_fmt; encode_tall.hip
  // has the same scheme): the smallest allowed first -- tight (int16 less the column's centre) where the engine allows it,
  // else narrow --, and when an accumulator of K or V does not fit, the layer is done again in the smallest form that
  // holds it (kv_wide_flag: 1 = not tight, 2 = not narrow; two words used in turn, encode_tall.hip).
  __shared__ int kv_wide_flag[2];
  int kvf = 0;
  const bool try_narrow = f.kv24 && f.kv_fmt != nullptr;
  for (int l = 0; l < f.Ld; ++l) {
    SLIMT_PHASE_LANE;
    const int col = wave * 16 + lr;
    const bool try_tight = try_narrow && f.kv_tight_limit > 0 && ((f.kv_tight_layers >> l) & 1u);
    int form = !try_narrow ? 1 : try_tight ? 2 : 0;  // kv_fmt's codes
   for (int attempt = 0; attempt < 3; ++attempt) {
    const bool wide = form == 1;
    bool redo = false;
    for (int which = 0; which < 2; ++which) {
      const PreparedWeight &W = which == 0 ? f.dec_k[l] : f.dec_v[l];
      float *out = f.kv + (size_t)(2 * l + which) * M * D;
      v4i bf[KSD];
      load_frags<KSD>(bf, W, wave, 0, lane);
      int cs;
      float pb;
      load_epi(W, wave, lr, cs, pb);
      quantise_rows(X, W.a_quant);
      if (!wide && which == 0) {  // (raised behind the staging only; encode_tall.hip)
        kvf ^= 1;
        if (tid == 0) kv_wide_flag[kvf] = 0;
      }
      __syncthreads();
      v4i acc[NRT];
      zero_acc(acc);
      mma_rows(A, bf, acc);
      // K cache layout [sentence][head][d/4][key][4]: this wave's columns lie in one head
      // (hw), whose [d/4][S][4] block is the descriptor; V: the sentence's [S][D] rows
      const int hw = __builtin_amdgcn_readfirstlane((wave * 16) / DH), d = col % DH;
      const rsrc_t ro = which == 0 ? make_rsrc(out + ((size_t)b * H + hw) * DH * S, (unsigned)S * DH * 4u)
                                   : make_rsrc(out + (size_t)row0 * D, sbytes);
      const int voff = which == 0 ? (((d >> 2) * S + lg * 4) * 4 + (d & 3)) * 4 : (lg * 4 * D + col) * 4;
      const int rstep = which == 0 ? 16 : D * 4;  // bytes per row
      if (f.kv24) {
        // The packed cache (kernels.h, FusedDecodeArgs::kv24): the shifted accumulators accS = acc +
        // 127 colsum as 24-bit integers, 16 values = 48 bytes = one 16-byte store in each of three planes.
        // They pass through LDS (the operand and the attention staging are free: [16 nrt][256] int32)
        // so that a thread holds 16 columns of one key (K) or 4 keys x 4 columns (V).
        __syncthreads();  // every wave has read its A fragments
        int *stg = reinterpret_cast<int *>(smem);
        unsigned outside = 0;  // an accumulator of one of the sentence's rows outside the form's range
        const unsigned lim = (unsigned)f.kv_narrow_limit, lim16 = (unsigned)f.kv_tight_limit;
        const int ctr = form == 2 ? f.kv_centre[l][which][col] : 0;  // tight: staged less the column's centre
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (rt < nrt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int v = acc[rt][r] + __mul24(127, cs);
              stg[(16 * rt + lg * 4 + r) * D + col] = v - ctr;
              if (!wide && 16 * rt + lg * 4 + r < S) {
                outside |= (unsigned)((unsigned)v + lim >= 2u * lim) << 1;
                if (form == 2) outside |= (unsigned)((unsigned)(v - ctr) + lim16 >= 2u * lim16);
              }
            }
          }
        if (outside) __hip_atomic_fetch_or(&kv_wide_flag[kvf], (int)outside, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        // (uniform: ONE read by every thread behind the barrier; a tight attempt that holds does not care about accS's range)
        const int raised = wide ? 0 : kv_wide_flag[kvf];
        if (raised & (form == 2 ? 1 : 2)) {
          redo = true;
          form = (raised & 2) ? 1 : 0;
          break;
        }
        const int Sp = (S + 3) & ~3;
        if (form == 2) {
          // the tight form (decode_attention_packed.inl.h, attention_packed128 with Form16): one thread = 32 values = four quads of int16;
          // K [head][plane 0..3][key][16 B], V [key / 8][plane 0..3][column / 4][16 B]
          if (which == 0) {
            const rsrc_t rp = make_rsrc(reinterpret_cast<const char *>(out) + (size_t)b * S * D * 3, (unsigned)(S * D * 3));
            for (int it = tid; it < S * (D / 32); it += 1024) {
              const int r = it % S, h = it / S;
              const int off = (h * 4 * S + r) * 16;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const v4i pk = pack16(*reinterpret_cast<const v4i *>(stg + r * D + 32 * h + 8 * q),
                                      *reinterpret_cast<const v4i *>(stg + r * D + 32 * h + 8 * q + 4));
                __builtin_amdgcn_raw_buffer_store_b128(pk, rp, off + q * S * 16, 0, 0);
              }
            }
          } else {
            const rsrc_t rp = make_rsrc(reinterpret_cast<const char *>(out) + (size_t)b * Sp * D * 3, (unsigned)(Sp * D * 3));
            const int G = (S + 7) >> 3;
            for (int it = tid; it < G * 64; it += 1024) {
              const int cl = it & 63, g = it >> 6;
              const int off = (g * 4 * 64 + cl) * 16;
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
                const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
                const v4i z = {0, 0, 0, 0};
                const v4i x0 = *reinterpret_cast<const v4i *>(stg + (k0 < S ? k0 : 0) * D + 4 * cl);
                const v4i x1 = *reinterpret_cast<const v4i *>(stg + (k1 < S ? k1 : 0) * D + 4 * cl);
                const v4i pk = pack16(k0 < S ? x0 : z, k1 < S ? x1 : z);
                __builtin_amdgcn_raw_buffer_store_b128(pk, rp, off + q * 1024, 0, 0);
              }
            }
          }
          __syncthreads();
          continue;
        }
        if (!wide) {
          // the narrow form (decode_attention_packed.inl.h, attention_packed128 with Form20): one thread = 32 values = four quads of hi halves +
          // one quad of lo nibbles; K [head][plane 0..4][key][16 B], V [key / 8][plane 0..4][column / 4][16 B]
          if (which == 0) {
            const rsrc_t rp = make_rsrc(reinterpret_cast<const char *>(out) + (size_t)b * S * D * 3, (unsigned)(S * D * 3));
            for (int it = tid; it < S * (D / 32); it += 1024) {
              const int r = it % S, h = it / S;
              const int off = (h * 5 * S + r) * 16;
              int lo[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const Packed20 pk = pack20(*reinterpret_cast<const v4i *>(stg + r * D + 32 * h + 8 * q),
                                           *reinterpret_cast<const v4i *>(stg + r * D + 32 * h + 8 * q + 4));
                lo[q] = pk.lo;
                __builtin_amdgcn_raw_buffer_store_b128(pk.hi, rp, off + q * S * 16, 0, 0);
              }
              const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
              __builtin_amdgcn_raw_buffer_store_b128(lq, rp, off + 4 * S * 16, 0, 0);
            }
          } else {
            const rsrc_t rp = make_rsrc(reinterpret_cast<const char *>(out) + (size_t)b * Sp * D * 3, (unsigned)(Sp * D * 3));
            const int G = (S + 7) >> 3;
            for (int it = tid; it < G * 64; it += 1024) {
              const int cl = it & 63, g = it >> 6;
              const int off = (g * 5 * 64 + cl) * 16;
              int lo[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
                const int k0 = 8 * g + 2 * q, k1 = k0 + 1;
                const v4i z = {0, 0, 0, 0};
                const v4i x0 = *reinterpret_cast<const v4i *>(stg + (k0 < S ? k0 : 0) * D + 4 * cl);
                const v4i x1 = *reinterpret_cast<const v4i *>(stg + (k1 < S ? k1 : 0) * D + 4 * cl);
                const Packed20 pk = pack20(k0 < S ? x0 : z, k1 < S ? x1 : z);
                lo[q] = pk.lo;
                __builtin_amdgcn_raw_buffer_store_b128(pk.hi, rp, off + q * 1024, 0, 0);
              }
              const v4i lq = {lo[0], lo[1], lo[2], lo[3]};
              __builtin_amdgcn_raw_buffer_store_b128(lq, rp, off + 4 * 1024, 0, 0);
            }
          }
          __syncthreads();
          continue;
        }
        if (which == 0) {  // K [sentence][column / 16][plane][key][16 B]: consecutive lanes = consecutive keys
          const rsrc_t rp = make_rsrc(reinterpret_cast<const char *>(out) + (size_t)b * S * D * 3, (unsigned)(S * D * 3));
          for (int it = tid; it < S * (D / 16); it += 1024) {
            const int r = it % S, ci = it / S;
            v3i wd[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wd[g] = pack24(*reinterpret_cast<const v4i *>(stg + r * D + 16 * ci + 4 * g));
            const int off = (ci * 3 * S + r) * 16;
            const v4i p0 = {wd[0].x, wd[0].y, wd[0].z, wd[1].x}, p1 = {wd[1].y, wd[1].z, wd[2].x, wd[2].y},
                      p2 = {wd[2].z, wd[3].x, wd[3].y, wd[3].z};
            __builtin_amdgcn_raw_buffer_store_b128(p0, rp, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p1, rp, off + S * 16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p2, rp, off + 2 * S * 16, 0, 0);
          }
        } else {  // V [sentence][key / 4][plane][column / 4][16 B]: 4 keys x 4 columns, key-major
          const rsrc_t rp = make_rsrc(reinterpret_cast<const char *>(out) + (size_t)b * Sp * D * 3, (unsigned)(Sp * D * 3));
          for (int it = tid; it < (Sp / 4) * 64; it += 1024) {
            const int cl = it & 63, g = it >> 6;
            v3i wd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // keys past the sentence: zeros (finite once unpacked, weight 0)
              const int key = 4 * g + i;
              const v4i xv = *reinterpret_cast<const v4i *>(stg + (key < S ? key : 0) * D + 4 * cl);
              const v4i z = {0, 0, 0, 0};
              wd[i] = pack24(key < S ? xv : z);
            }
            const int off = (g * 3 * 64 + cl) * 16;
            const v4i p0 = {wd[0].x, wd[0].y, wd[0].z, wd[1].x}, p1 = {wd[1].y, wd[1].z, wd[2].x, wd[2].y},
                      p2 = {wd[2].z, wd[3].x, wd[3].y, wd[3].z};
            __builtin_amdgcn_raw_buffer_store_b128(p0, rp, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p1, rp, off + 1024, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p2, rp, off + 2048, 0, 0);
          }
        }
        __syncthreads();
        continue;
      }
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        if (rt < nrt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + lg * 4 + r;
            if (row < S) bstore(ro, voff, (16 * rt + r) * rstep, (float)(acc[rt][r] + __mul24(127, cs)));  // float(accS): kernels.h, kv24
          }
        }
      __syncthreads();
    }
    if (!redo) break;
   }
    if (f.kv_fmt && f.kv24 && tid == 0) {
      f.kv_fmt[(size_t)l * f.B + b] = (unsigned char)form;
      if (form == 1 && f.kv_wide_count) __hip_atomic_fetch_add(f.kv_wide_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (try_tight && form != 2 && f.kv_not16_count)
        __hip_atomic_fetch_add(f.kv_not16_count + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

size_t long16_lds_bytes() {
  return (size_t)LR * (256 + 16) + 2 * 3 * (size_t)LR * 33 * sizeof(float);
}

bool long_encode_supported(int D, int F, int H, int Le, int Ld, int S) {
  if (S < 1 || S > LR || Le < 1 || Le > 6 || Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  return D == 256 && F % 256 == 0 && F <= 4096 && D / H == 32 && long16_lds_bytes() <= 160 * 1024;
}

hipError_t launch_encode_long(const LongEncodeArgs &a, hipStream_t st) {
  if (!long_encode_supported(a.D, a.F, a.H, a.f.Le, a.f.Ld, a.f.S)) return hipErrorInvalidValue;
  const size_t lds = long16_lds_bytes();
  auto k = a.f.S <= 64 ? encode_long16_kernel<2> : encode_long16_kernel<4>;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(a.f.B), dim3(1024), lds, st, a);
  return hipGetLastError();
}

// workgroups launched: with tickets a quarter more than tiles (at least 32 more: one
// candidate per shader engine and XCD), rounded up to a multiple of 32
int fused_encode_grid(int B, int S, bool tickets) {
  const int spw = ER / S;
  const int tiles = (B + spw - 1) / spw;
  if (!tickets) return tiles;
  const int extra = tiles / 4 > 32 ? tiles / 4 : 32;
  return (tiles + extra + 31) / 32 * 32;
}

size_t fused_encode_lds_bytes(int D) {
  return (size_t)ER * (D + 4) * 4 + 3 * (size_t)ER * (D + 16) + 2 * (size_t)ER * (D + 4) * 4 +
         (size_t)ER * (D + 16) * 4;
}

bool fused_encode_supported(int D, int F, int H, int Le, int Ld, int S) {
  if (S < 1 || S > ER || Le < 1 || Le > 6 || Ld < 1 || Ld > 4) return false;
  if (H <= 0 || D % H || F % 256) return false;
  if (wide_encode_supported(D, F, H, Le, Ld, S)) return true;
  return D == 256 && D / H == 32 && fused_encode_lds_bytes(D) <= 160 * 1024;
}

hipError_t launch_encode_fused(const FusedEncodeArgs &a, int D, int F, int H, hipStream_t st) {
  if (!fused_encode_supported(D, F, H, a.Le, a.Ld, a.S)) return hipErrorInvalidValue;
  if (wide_encode_supported(D, F, H, a.Le, a.Ld, a.S)) return launch_encode_wide(a, st);
  const dim3 grid(fused_encode_grid(a.B, a.S, a.ticket != nullptr));
  const size_t lds = fused_encode_lds_bytes(D);
  if (a.gen.w2o && (!a.ticket || !a.gen_flag || !a.pack_tiles ||
                    shortlist_in_launch_lds_bytes(a.gen.src_vocab, a.gen.tgt_vocab) > lds))
    return hipErrorInvalidValue;  // (in-launch shortlist generation: see launch_encode_tall)
  hipError_t e = hipSuccess;
#define SLIMT_ENC_CASE(KSF_)                                                                   \
  if (F == 64 * KSF_) {                                                                        \
    auto k = encode_fused_kernel<4, KSF_, 32>;                                                 \
    e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds);             \
    if (e != hipSuccess) return e;                                                             \
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, st, a);                                       \
    return hipGetLastError();                                                                  \
  }
  SLIMT_ENC_CASE(24) SLIMT_ENC_CASE(16) SLIMT_ENC_CASE(32)
#undef SLIMT_ENC_CASE
  return hipErrorInvalidValue;
}

}  // namespace slimt_hip
