// Persistent fused greedy decoder: ONE launch runs the whole decode loop of
// Model::decode (slimt/Model.cc:111-185) for a batch.
//
// The decoder (SSRU + cross-attention + FFN, slimt/Modules.cc:190-259) never
// mixes sentences, so a workgroup can own 16 sentences (one MFMA row tile)
// for ALL steps with no inter-workgroup communication at all:
//   * activations, the SSRU cell states and the int8 A operands live in LDS
//     for the whole loop; nothing but tokens / alignments is written to HBM;
//   * the ~3.4 MB of decoder + shortlist weights are streamed from L2 each step
//     in MFMA-fragment order (1 KiB coalesced per operand) through buffer loads
//     with scalar addressing, 16 waves keeping 3-4 chunks of 4 fragments in
//     flight each (SLIMT_NB_FFN / SLIMT_NB_OUT);
//   * one wave per sentence does LayerNorm, embedding lookup, cross-attention
//     (two heads per wave64 pass when S <= 32, d_head == 32) and the greedy
//     bookkeeping (EOS / lengths / alignments);
//   * no launch, no host round trip and no grid-wide barrier per step, and a
//     batch only occupies ceil(B/16) CUs, so independent batches (slimt's
//     Async workers, Frontend.cc:212-226) overlap on one GPU.
// Arithmetic is bit-identical to the step-wise kernels (decode_kernels.hip).
#include "device_common.h"
#include "kernels.h"

// Bandwidth experiments only (wrong results): shrink the weight / K-V footprint.
#ifdef SLIMT_EXP_WSMALL
#define EXP_TILE(t) ((t) & 3)
#else
#define EXP_TILE(t) (t)
#endif
#ifdef SLIMT_EXP_KVSMALL
#define EXP_SENT(b) ((b) & 1)
#else
#define EXP_SENT(b) (b)
#endif

#ifndef SLIMT_KV_AUX_NT
#define SLIMT_KV_AUX_NT 2  // cache-policy bits of the "streamed" K/V loads (experiments: -DSLIMT_KV_AUX_NT=...)
#endif
#ifndef SLIMT_KV_AUX_KEEP
#define SLIMT_KV_AUX_KEEP 0  // ... of the "kept" ones
#endif

namespace slimt_hip {

namespace {

// prefetch depth (chunks of CH fragments in flight per wave) of the streamed GEMMs
#ifndef SLIMT_NB_FFN
#define SLIMT_NB_FFN 3
#endif
#ifndef SLIMT_NB_OUT
#define SLIMT_NB_OUT 4
#endif

constexpr int NW = 16;   // waves per workgroup
constexpr int CH = 4;    // weight fragments per prefetch chunk (per wave)

struct Frags {
  v4i f[CH];
  // epilogue constants of the chunk's tiles (this lane's column), fetched with
  // the fragments: loading them inside the epilogue costs one exposed memory
  // round trip per tile (measured: 16 of the 20 us of the logits phase)
  int cs[CH];
  float pb[CH];
};

// y = float(acc + 127 colsum) * u + pb   (Intgemm.inl.cc:146-153)
// y = float(acc + 127 colsum) * u + pb (Intgemm.inl.cc:146-153; |colsum| <= 127 K < 2^23) for the four rows an accumulator
// lane holds -- they share their column: one shift, two packed multiplies and two packed adds (v_pk_mul_f32 / v_pk_add_f32:
// the same IEEE operations as the scalar forms, multiply and add stay separate roundings).
typedef float dq2 __attribute__((ext_vector_type(2)));
struct Dequant4 {
  float v[4];
};
__device__ __forceinline__ Dequant4 dequant4(const int (&acc)[4], int colsum, float u, float pb) {
  const int sh = __mul24(127, colsum);
  dq2 lo = {(float)(acc[0] + sh), (float)(acc[1] + sh)};
  dq2 hi = {(float)(acc[2] + sh), (float)(acc[3] + sh)};
  const dq2 uu = {u, u}, pp = {pb, pb};
  lo = lo * uu;
  hi = hi * uu;
  lo = lo + pp;
  hi = hi + pp;
  Dequant4 o;
  o.v[0] = lo.x; o.v[1] = lo.y; o.v[2] = hi.x; o.v[3] = hi.y;
  return o;
}
__device__ __forceinline__ Dequant4 dequant4(const v4i &acc, int colsum, float u, float pb) {
  const int a[4] = {acc[0], acc[1], acc[2], acc[3]};
  return dequant4(a, colsum, u, pb);
}

// Weight streams go through buffer loads: the descriptor and the tile offset
// are wave-uniform (SGPRs), the only vector operand is lane * 16. That keeps
// 64-bit per-lane addresses out of the VGPR budget (they used to spill, and a
// scratch reload inside the loop drains every prefetch in flight), and reads
// past the last tile return zeros, so prefetches need no predicate and the
// compiler's s_waitcnt counts stay exact.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ v4i load_frag(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Stream this wave's tiles (tile = wave + 16 i) of a [K = 64 KS] x N weight,
// A operand (16 x K int8) in LDS. epi(tile, acc, colsum, pb) once per finished
// tile. NB chunks of CH fragments (1 KiB each) are kept in flight per wave.
// `wave` must be wave-uniform in the compiler's eyes (readfirstlane).
// NT > 0: every wave has exactly NT tiles (n_tiles == 16 NT, known at compile time): the
// tile loop is then fully unrolled. Not for speed of the loop itself: in the ROLLED loop the
// compiler's s_waitcnt insertion merges the preheader's and the back edge's pending loads
// and emits s_waitcnt vmcnt(0) at the top of every iteration (seen in the ISA of the FFN
// loops), which drains all NB chunks in flight and exposes a full memory round trip per
// iteration; straight-line code gets exact counts.
// PADDED (with NT == 0, a column count known only at run time: the output layer): the tile
// loop runs whole rounds of NB chunks with no conditional inside -- tiles past the last one
// read zeros (nothing is fetched past a descriptor) and `epi` must ignore them itself (the
// arg-max epilogue does: its columns are >= N). A branch-free single-block loop is what
// the s_waitcnt insertion counts exactly; with the `if (chunk < n)` of the general form
// inside, it falls back to vmcnt(0) at the loop head whenever register allocation shifts.
// RT row tiles (16 RT rows of A): every weight fragment feeds RT MFMAs; epi(tile, rt, ...)
// runs once per finished (column tile, row tile).
// The stream's descriptors and this lane's offsets.
template <int KS, int NT, bool PAIRED = false>
struct StreamSrc {
  rsrc_t rw, rc, rp;
  int voff, eoff, n_tiles;
  __device__ __forceinline__ StreamSrc(const PreparedWeight &w, int lane) {
    n_tiles = NT > 0 ? NT * NW : w.n_tiles;
    rw = make_rsrc(w.Wp, (unsigned)n_tiles * KS * 1024u);
    if constexpr (PAIRED) {  // pair constants (kernels.h, PreparedWeight::cp4): one descriptor, 16 bytes per lane
      rc = make_rsrc(w.cp4, (unsigned)(16 * ((n_tiles + 31) / 32)) * 256u);
      rp = rc;
      eoff = (lane & 15) * 16;
    } else {
      rc = make_rsrc(w.colsum, (unsigned)n_tiles * 64u);
      rp = make_rsrc(w.pb, (unsigned)n_tiles * 64u);
      eoff = (lane & 15) * 4;
    }
    voff = lane * 16;
  }
  // K = 256 (one tile per chunk): chunk c's four fragments; the ODD chunk of a pair (2 p, 2 p + 1) -- or an even
  // last chunk without a partner -- also fetches the pair's epilogue constants as one quad: cs[0], pb[0] = the even
  // chunk's, cs[1], pb[1] = the odd one's (HOLD). With the odd chunk because both chunks' constants are then dead
  // when its buffer is requested again: fetched with the even chunk, the odd chunk's would have to be copied across
  // the loop's back edge, and the copy waits for the load (measured: one drained round trip per round).
  template <bool HOLD>
  __device__ __forceinline__ void load_paired(Frags &bb, int c, int wave) const {
    static_assert(KS == CH && PAIRED, "one tile per chunk");
    const int tile = wave + NW * c;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bb.f[ks] = load_frag(rw, voff, (tile * KS + ks) * 1024);
#ifndef SLIMT_EXP_NOEPI
    if constexpr (HOLD) {
      const v4i q = load_frag(rc, eoff, (NW * (c >> 1) + wave) * 256);
      // (__int_as_float on a copy: __builtin_bit_cast(float, q.y) of a vector ELEMENT reads element 0 with this compiler)
      const int y = q.y, w4 = q.w;
      bb.cs[0] = q.x;
      bb.pb[0] = __int_as_float(y);
      bb.cs[1] = q.z;
      bb.pb[1] = __int_as_float(w4);
    }
#endif
  }
  // chunk c of wave `wave`'s stream: CH fragments + the epilogue constants of its tile(s)
  __device__ __forceinline__ void load(Frags &bb, int c, int wave) const {
    if constexpr (KS <= CH) {
      constexpr int TPC = CH / KS;  // whole tiles per chunk
#pragma unroll
      for (int j = 0; j < TPC; ++j) {
        const int tile = EXP_TILE(wave + NW * (c * TPC + j));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bb.f[j * KS + ks] = load_frag(rw, voff, (tile * KS + ks) * 1024);
#ifdef SLIMT_EXP_NOEPI  // timing experiment: the stream without its epilogue-constant loads (results are wrong)
        bb.cs[j] = tile;
        bb.pb[j] = 0.5f;
#else
        bb.cs[j] = __builtin_amdgcn_raw_buffer_load_b32(rc, eoff, tile * 64, 0);
        bb.pb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, eoff, tile * 64, 0));
#endif
      }
    } else {
      constexpr int CPT = KS / CH;  // chunks per tile
      const int tile = EXP_TILE(wave + NW * (c / CPT));
      const int ks0 = (c % CPT) * CH;
#pragma unroll
      for (int p = 0; p < CH; ++p) bb.f[p] = load_frag(rw, voff, (tile * KS + ks0 + p) * 1024);
      bb.cs[0] = __builtin_amdgcn_raw_buffer_load_b32(rc, eoff, tile * 64, 0);
      bb.pb[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, eoff, tile * 64, 0));
    }
  }
};

// The first NB chunks of a stream, requested ahead of stream_gemm_from: weights do not depend
// on the phase before, so a wave asks for them as soon as its own work in that phase is done
// (or before it, where that pays) and the round trip runs under the barrier wait (lds_barrier
// keeps global loads in flight).
template <int KS, int NB, int NT = 0, bool PADDED = false>
__device__ __forceinline__ void stream_prologue(const PreparedWeight &w, int wave, int lane, Frags (&b)[NB]) {
  constexpr bool PAIRED = KS == CH && (NT > 0 || (PADDED && NB % 2 == 0));  // epilogue constants fetched per two chunks
  const StreamSrc<KS, NT, PAIRED> src(w, lane);
  if constexpr (NT > 0) {
    constexpr int NCH = KS <= CH ? (NT + CH / KS - 1) / (CH / KS) : NT * (KS / CH);
#pragma unroll
    for (int k = 0; k < NB && k < NCH; ++k) {
      if constexpr (PAIRED) {
        if (k % 2 == 1 || k + 1 >= NCH)
          src.template load_paired<true>(b[k], k, wave);
        else
          src.template load_paired<false>(b[k], k, wave);
      } else {
        src.load(b[k], k, wave);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // straight-line code: keep the scheduler from sinking
  } else if constexpr (PADDED) {        // the prefetches next to their uses
    // the chunks are requested in the same order before and inside the loop (the
    // scheduler would otherwise regroup the prologue's loads, and the pending-load
    // orders of the two loop entries would no longer match)
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if constexpr (PAIRED) {
        if (k % 2 == 1)
          src.template load_paired<true>(b[k], k, wave);
        else
          src.template load_paired<false>(b[k], k, wave);
      } else {
        src.load(b[k], k, wave);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NB; ++k) src.load(b[k], k, wave);
  }
}

// The stream proper; b holds the chunks stream_prologue requested.
template <int KS, int NB, int NT = 0, bool PADDED = false, int RT = 1, class Epi>
__device__ __forceinline__ void stream_gemm_from(const char *A, int lda, const PreparedWeight &w, int wave,
                                                 int lane, Frags (&b)[NB], Epi &&epi) {
  constexpr bool PAIRED = KS == CH && (NT > 0 || (PADDED && NB % 2 == 0));
  const StreamSrc<KS, NT, PAIRED> src(w, lane);
  const int n_tiles = src.n_tiles;
  const int lr = lane & 15, lg = lane >> 4;
  const int ntw = n_tiles > wave ? (n_tiles - wave + NW - 1) / NW : 0;  // my tiles
  auto load = [&](Frags &bb, int c) { src.load(bb, c, wave); };
  if constexpr (KS <= CH) {
    constexpr int TPC = CH / KS;  // whole tiles per chunk
    v4i af[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        af[rt][ks] = *reinterpret_cast<const v4i *>(A + (rt * 16 + lr) * lda + ks * 64 + lg * 16);
    const int nch = (ntw + TPC - 1) / TPC;
    auto tile_mma_c = [&](const Frags &bb, int j, int tile, int cs, float pb) {
      v4i acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = v4i{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rt][ks], bb.f[j * KS + ks], acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) epi(tile, rt, acc[rt], cs, pb);
    };
    auto tile_mma = [&](const Frags &bb, int j, int tile) { tile_mma_c(bb, j, tile, bb.cs[j], bb.pb[j]); };
    auto compute = [&](const Frags &bb, int c) {
#pragma unroll
      for (int j = 0; j < TPC; ++j) {
        const int i = c * TPC + j;
        if (i < ntw) tile_mma(bb, j, wave + NW * i);
      }
    };
    // PAIRED: a pair's constants live in the buffer of its odd chunk (StreamSrc::load_paired)
    if constexpr (NT > 0) {
      constexpr int NCH = (NT + TPC - 1) / TPC;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if constexpr (PAIRED) {
          Frags &bb = b[c % NB];
          const bool own = c % 2 == 1 || c + 1 >= NCH;  // this chunk's buffer holds the pair's constants
          const Frags &hold = own ? bb : b[(c + 1) % NB];
          tile_mma_c(bb, 0, wave + NW * c, hold.cs[c % 2], hold.pb[c % 2]);
          if (c + NB < NCH) {
            if ((c + NB) % 2 == 1 || c + NB + 1 >= NCH)
              src.template load_paired<true>(bb, c + NB, wave);
            else
              src.template load_paired<false>(bb, c + NB, wave);
          }
        } else {
          compute(b[c % NB], c);
          if (c + NB < NCH) load(b[c % NB], c + NB);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (PADDED) {
      static_assert(TPC == 1, "one tile per chunk");
      const int nchp = (nch + NB - 1) / NB * NB;
      for (int c = 0; c < nchp; c += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if constexpr (PAIRED) {
            if (k % 2 == 0) {
              tile_mma_c(b[k], 0, wave + NW * (c + k), b[k + 1].cs[0], b[k + 1].pb[0]);
              src.template load_paired<false>(b[k], c + k + NB, wave);
            } else {
              tile_mma_c(b[k], 0, wave + NW * (c + k), b[k].cs[1], b[k].pb[1]);
              src.template load_paired<true>(b[k], c + k + NB, wave);
            }
          } else {
            tile_mma(b[k], 0, wave + NW * (c + k));
            load(b[k], c + k + NB);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      for (int c = 0; c < nch; c += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (c + k < nch) compute(b[k], c + k);
          load(b[k], c + k + NB);
        }
      }
    }
  } else {
    constexpr int CPT = KS / CH;  // chunks per tile
    static_assert(KS % CH == 0, "K/64 must be a multiple of CH here");
    const int nch = ntw * CPT;
    v4i acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = v4i{0, 0, 0, 0};
    // chunk c of the stream: kc = its position inside its tile (compile time where unrolled)
    auto chunk_mma = [&](const Frags &bb, int kc, int tile) {
      const int ks0 = kc * CH;
#pragma unroll
      for (int p = 0; p < CH; ++p)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const v4i af = *reinterpret_cast<const v4i *>(A + (rt * 16 + lr) * lda + (ks0 + p) * 64 + lg * 16);
          acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bb.f[p], acc[rt], 0, 0, 0);
        }
      if (kc == CPT - 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          epi(tile, rt, acc[rt], bb.cs[0], bb.pb[0]);
          acc[rt] = v4i{0, 0, 0, 0};
        }
      }
    };
    if constexpr (NT > 0) {
      constexpr int NCH = NT * CPT;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        chunk_mma(b[c % NB], c % CPT, wave + NW * (c / CPT));
        if (c + NB < NCH) load(b[c % NB], c + NB);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (PADDED) {
      static_assert(NB % CPT == 0, "whole tiles per round");
      const int nchp = (nch + NB - 1) / NB * NB;
      for (int c = 0; c < nchp; c += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {  // c is a multiple of NB, NB of CPT
          chunk_mma(b[k], k % CPT, wave + NW * ((c + k) / CPT));
          load(b[k], c + k + NB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      for (int c = 0; c < nch; c += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (c + k < nch) chunk_mma(b[k], (c + k) % CPT, wave + NW * ((c + k) / CPT));
          load(b[k], c + k + NB);
        }
      }
    }
  }
}

template <int KS, int NB, int NT = 0, bool PADDED = false, int RT = 1, class Epi>
__device__ __forceinline__ void stream_gemm(const char *A, int lda, const PreparedWeight &w,
                                            int wave, int lane, Epi &&epi) {
  Frags b[NB];
  stream_prologue<KS, NB, NT, PADDED>(w, wave, lane, b);
  stream_gemm_from<KS, NB, NT, PADDED, RT>(A, lda, w, wave, lane, b, epi);
}

// canonical LayerNorm of row `src` (LDS) by one wave -> dst (LDS f32) and,
// if A != nullptr, its int8 quantisation with aq.
template <int DPL>
__device__ __forceinline__ void ln_row(const float *src, const float *scale, const float *bias,
                                       float eps, float *dst, char *A, float aq, int lane) {
  constexpr int D = 64 * DPL;
  float v[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) v[i] = src[lane + 64 * i];
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
  SharedDiv(sigma, SLIMT_DIV_LN_D).quot<DPL, false>(tq, SLIMT_DIV_LN_N);  // (v - mean) / sigma, correctly rounded (device_common.h)
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    const float m = scale[lane + 64 * i] * tq[i];
    const float y = m + bias[lane + 64 * i];
    dst[lane + 64 * i] = y;
    if (A) A[lane + 64 * i] = (char)quantize1_byte(y, aq);
  }
}

// Address spaces are spelled out in the per-sentence argument block: through
// generic pointers the compiler falls back to FLAT loads (the generic S > 32
// path still loads through these), which it serialises one by one.
#define SLIMT_GLOBAL __attribute__((address_space(1)))
#define SLIMT_LDS __attribute__((address_space(3)))
typedef const SLIMT_GLOBAL float *gcf_ptr;
typedef SLIMT_GLOBAL float *gf_ptr;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const SLIMT_GLOBAL f4 *gcf4_ptr;
typedef const SLIMT_LDS float *lcf_ptr;
typedef const SLIMT_LDS f4 *lcf4_ptr;
typedef SLIMT_LDS char *lc_ptr;

struct AttnRow {
  gcf_ptr kl, vl;  // this sentence's cached K / V [S][D]: float(accS) in the f32 form (see unpack24f below)
  gcf_ptr pbk, pbv;  // f32 form: the K / V projections' prepared biases [D] (global memory)
  float uk, uv;      // f32 form: their unquantisation multipliers
  lcf_ptr qrow;    // LDS: q [D]
  lc_ptr arow;     // LDS: int8 output row [D] (A operand of the O projection)
  SLIMT_LDS float *pbuf;  // LDS: 64 floats of per-wave scratch
  SLIMT_LDS float *hsum;  // LDS: 16 floats of per-wave scratch (the heads' probability sums P_h)
  int S, len;
  float alpha, aq_o;
  gf_ptr attn;   // nullable [H][S]
  gf_ptr align;  // nullable [S]: head 0 (update_alignment, Model.cc:84-108)
};

// scaled_dot_product_attention (Modules.cc:24-86) for ONE sentence, all heads,
// by one wave. Inlined: with buffer loads (no per-lane 64-bit addresses) its
// registers no longer collide with the GEMM phases', and a call would save and
// restore 48 callee-saved VGPRs through scratch every layer and step.
// The K/V cache is streamed once per step by exactly one workgroup; weights are
// shared by every workgroup. KV_AUX (a template parameter of the attention code: the
// cache-policy immediate of its buffer loads) = 2 marks the cache loads non-temporal.
// Which is better depends on the load: while the K/V of all decoders in flight fits
// the 256 MB Infinity Cache, every step re-reads it from there and non-temporal loads
// only lose that (1 / 4 / 8 workers: -8 / -12 / -8 %); beyond it they stop the streams
// from displacing the weights (20 workers: +3 %, 64-token sentences +7 %). The engine
// picks the variant per launch (translate_device).

// Cross-attention of one sentence for 32 < S <= 128 (d_head 32). Out of line:
// it is register-hungry (two groups of K or V in flight) and rare enough that
// the callee-saved spills of a call do not matter, while inlined it would cost
// the S <= 32 path its registers.
__device__ __forceinline__ const float *uniform_ptr(const float *p) {
  // function arguments arrive in VGPRs; a buffer descriptor built from them would
  // make every load a waterfall loop. These values are wave-uniform: say so.
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
}

template <int D, int DH, int KV_AUX>
__device__ __noinline__ void attention_row_long(AttnRow r, int lane) {
  constexpr int H = D / DH;
  const int S = __builtin_amdgcn_readfirstlane(r.S), len = __builtin_amdgcn_readfirstlane(r.len);
  const int lenf = len > 0 ? len : S;  // keys fetched
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  {
    // 32 < S <= 128: two heads per pass as above, each lane holding the keys
    // j, j + 32, j + 64, j + 96 of its head. K arrives 32 keys (8 x 16 B per
    // lane) at a time with the next group in flight, V likewise in groups of 32
    // keys; the probabilities of a head go through LDS (pbuf: 2 x 128 floats).
    // Row sums keep the canonical 128-column order: lane L = key % 64 first adds
    // keys L and L + 64, then the 64-lane butterfly -- here (e0 + e2), (e1 + e3)
    // in the lane, the 32-lane butterfly on both, and their sum.
    const int hh = lane >> 5, j = lane & 31;
    const int ng = (S + 31) >> 5;
    const float uk = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r.uk)));
    const float uv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r.uv)));
    const float *pbk = uniform_ptr((const float *)r.pbk), *pbv = uniform_ptr((const float *)r.pbv);
    const rsrc_t rk = make_rsrc(uniform_ptr((const float *)r.kl), (unsigned)(S * D) * 4u);
    const rsrc_t rv = make_rsrc(uniform_ptr((const float *)r.vl), (unsigned)(lenf * D) * 4u);  // padding is
    const int koff = ((hh * (DH / 4) * S + j) * 4) * 4;                                        // not fetched
    const int voff = lane * 4;
#pragma unroll 1
    for (int hp = 0; hp < H / 2; ++hp) {
      const int h = 2 * hp + hh;
      auto load_k = [&](f4(&k4)[8], int g) {  // keys 32 g + j; masked keys: past the descriptor (zeros, no traffic)
        const int kg = (32 * g + j) < lenf ? koff : 0x40000000;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          k4[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                             rk, kg, (((2 * hp * (DH / 4) + i) * S + 32 * g) * 4) * 4, KV_AUX));
      };
      float sc[4];
      f4 ka[8], kb[8];
      load_k(ka, 0);
      // c_h (the hoisted order, unpack24f below): column lane + 64 hp belongs to head 2 hp + (lane >> 5) = h
      const float ch = half_sum(r.qrow[lane + 64 * hp] * pbk[lane + 64 * hp]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f4(&cur)[8] = (g & 1) ? kb : ka;
        f4(&nxt)[8] = (g & 1) ? ka : kb;
        float s = lowest;
        if (g < ng) {
          if (g + 1 < ng) load_k(nxt, g + 1);
          const int key = 32 * g + j;
          s = 0.0f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f4 q4 = *(lcf4_ptr)(r.qrow + h * DH + 4 * i);
            s = __builtin_fmaf(q4.x, cur[i].x, s);
            s = __builtin_fmaf(q4.y, cur[i].y, s);
            s = __builtin_fmaf(q4.z, cur[i].z, s);
            s = __builtin_fmaf(q4.w, cur[i].w, s);
          }
          s = __builtin_fmaf(s, uk, ch);
          if (r.alpha != 1.0f) s = r.alpha * s;
          s = s + (1.0f - (key < len ? 1.0f : 0.0f)) * minus_inf;
          if (key >= S) s = lowest;
        }
        sc[g] = s;
      }
      const float m = half_max(fmaxf(fmaxf(sc[0], sc[2]), fmaxf(sc[1], sc[3])));
#pragma unroll
      for (int g = 0; g < 4; ++g) sc[g] = (32 * g + j) < S ? exp_p(sc[g] - m) : 0.0f;
      const float sum = half_sum(sc[0] + sc[2]) + half_sum(sc[1] + sc[3]);
#pragma unroll
      for (int g = 0; g < 4; ++g) sc[g] = sc[g] / sum;  // keys >= S: exactly 0
      const float P = half_sum(sc[0] + sc[2]) + half_sum(sc[1] + sc[3]);  // P_h in the same canonical order
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key = 32 * g + j;
        const float p = sc[g];
        if (g < ng) {
          if (r.attn && key < S) r.attn[(size_t)h * S + key] = p;
          if (r.align && hp == 0 && hh == 0 && key < len) r.align[key] = p;
          r.pbuf[hh * 128 + key] = p;
        }
      }
      // V columns of this lane: (head parity, d = lane & 31), 32 keys per group
      auto load_v = [&](float(&v)[32], int g) {
#pragma unroll
        for (int jj = 0; jj < 32; ++jj)
          v[jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                rv, voff, ((32 * g + jj) * D + 2 * hp * DH) * 4, KV_AUX));
      };
      float va[32], vb[32];
      load_v(va, 0);
      float o = 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float(&cur)[32] = (g & 1) ? vb : va;
        float(&nxt)[32] = (g & 1) ? va : vb;
        if (g < ng) {
          if (g + 1 < ng) load_v(nxt, g + 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) {  // keys >= S: p == 0 and v == 0 (past the descriptor)
            const f4 p4 = *(lcf4_ptr)(r.pbuf + hh * 128 + 32 * g + 4 * i);
            o = __builtin_fmaf(p4.x, cur[4 * i + 0], o);
            o = __builtin_fmaf(p4.y, cur[4 * i + 1], o);
            o = __builtin_fmaf(p4.z, cur[4 * i + 2], o);
            o = __builtin_fmaf(p4.w, cur[4 * i + 3], o);
          }
        }
      }
      o = __builtin_fmaf(o, uv, pbv[2 * hp * DH + lane] * P);
      r.arow[2 * hp * DH + lane] = (char)quantize1_byte(o, r.aq_o);
    }
  }
}

// LONG: compile the 32 < S <= 128 path in (a call to attention_row_long). The
// S <= 32 instantiation of the kernel leaves it out: the mere call site cost
// the flagship path ~4 us per step (scratch frame, register allocation).
constexpr int kPastDescriptor = 0x40000000;  // lane offset no K/V descriptor reaches

template <int D, int DH, bool LONG, int KV_AUX = 0>
__device__ __forceinline__ void attention_row(AttnRow r, int lane) {
  constexpr int H = D / DH;
  const int S = r.S, len = r.len;
  const int lenf = len > 0 ? len : S;  // keys fetched (an empty sentence masks everything: uniform weights over real V)
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  if (DH == 32 && S <= 32) {
    // Scores: two heads per pass, lane = (head parity, key); the K rows of pass p+1 are
    // requested as soon as pass p's scores are done. The probabilities of all H heads
    // go to LDS (pbuf: [H][32]).
    // Output: V is streamed as WHOLE rows -- one 16-byte load per lane fetches a key's
    // row for all heads (D floats = 64 lanes x 16 B), lane l owning columns 4 l .. 4 l + 3
    // of head l / 8 -- in key order, 8 rows per group with the next group in flight, the
    // first group requested before the first score pass. Per column the sum is still the
    // ascending-key fmaf chain of the canonical order. 32 V loads per sentence and layer
    // instead of 128 four-byte ones: this phase is bound by the number of vector-memory
    // instructions the CU's address unit takes, not by their bytes (measured: with every
    // K/V load hitting two hot sentences the phase kept 11.3 of its 12.2 us).
    // Padding is never fetched: a masked key's probability is exactly 0 whatever its score
    // (exp_p underflows to 0 eighty-six units below the maximum, the mask is -1e8), and
    // fma(0, v, o) == o. Keys >= len therefore read past the descriptors (V: it ends after
    // len rows; K: a lane offset beyond it), return zeros and cost no memory traffic.
    static_assert(DH != 32 || D == 256, "a V row is one 16-byte load per lane");
    const int hh = lane >> 5, j = lane & 31;
    const int jc = j < S ? j : S - 1;
    const float mask = (1.0f - (j < len ? 1.0f : 0.0f)) * minus_inf;
    f4 k4[8];
    f4 vq[3][4];  // V rows in flight: three groups of four rows
    const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * D) * 4u);
    const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(__builtin_amdgcn_readfirstlane(lenf) * D) * 4u);
    const int koff = j < lenf ? ((hh * (DH / 4) * S + jc) * 4) * 4 : kPastDescriptor;  // [head][dh/4][S][4] floats
    const int voff = lane * 16;  // columns 4 lane .. 4 lane + 3 of a row
    auto load_k = [&](int hp) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        k4[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                           rk, koff, ((2 * hp * (DH / 4) + i) * S * 4) * 4, KV_AUX));
    };
    auto load_v = [&](f4(&vv)[4], int g) {  // rows 4 g .. 4 g + 3
#pragma unroll
      for (int i = 0; i < 4; ++i)
        vv[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, ((4 * g + i) * D) * 4, KV_AUX));
    };
    load_k(0);
    load_v(vq[0], 0);  // needed only after the last score pass: a long head start
    __builtin_amdgcn_sched_barrier(0);
    float ck[4];  // the hoisted order (unpack24f below): c_h of this lane's head in pass hp
#pragma unroll
    for (int i = 0; i < 4; ++i) ck[i] = half_sum(r.qrow[lane + 64 * i] * r.pbk[lane + 64 * i]);
#pragma unroll
    for (int hp = 0; hp < H / 2; ++hp) {
      const int h = 2 * hp + hh;
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f4 q4 = *(lcf4_ptr)(r.qrow + h * DH + 4 * i);
        s = __builtin_fmaf(q4.x, k4[i].x, s);
        s = __builtin_fmaf(q4.y, k4[i].y, s);
        s = __builtin_fmaf(q4.z, k4[i].z, s);
        s = __builtin_fmaf(q4.w, k4[i].w, s);
      }
      if (hp + 1 < H / 2) {
        load_k(hp + 1);
      } else {  // the K registers are free: two more groups of V rows
        load_v(vq[1], 1);
        load_v(vq[2], 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      s = __builtin_fmaf(s, r.uk, ck[hp]);
      if (r.alpha != 1.0f) s = r.alpha * s;
      s = s + mask;
      if (j >= S) s = lowest;
      const float m = half_max(s);
      const float e = j < S ? exp_p(s - m) : 0.0f;
      const float sum = half_sum(e);  // canonical order: masks 1..16; the mask-32 step would add +0
      const float p = e / sum;        // keys >= S: exactly 0
      const float ps = half_sum(p);   // P_h
      if (r.attn && j < S) r.attn[(size_t)h * S + j] = p;
      if (r.align && hp == 0 && hh == 0 && j < len) r.align[j] = p;
      r.pbuf[h * 32 + j] = p;
      if (j == 0) r.hsum[h] = ps;
    }
    // lane l: head l / 8, columns 4 l .. 4 l + 3; keys ascending
    const int ph = (lane >> 3) * 32;
    const float P = r.hsum[lane >> 3];
    const f4 pv4 = *(gcf4_ptr)(r.pbv + 4 * lane);
    f4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f4(&cur)[4] = vq[g % 3];
      const f4 p4 = *(lcf4_ptr)(r.pbuf + ph + 4 * g);
      const float pj[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // keys >= S: p == 0, fma(0, v, o) == o
        const f4 v4 = cur[c];
        o.x = __builtin_fmaf(pj[c], v4.x, o.x);
        o.y = __builtin_fmaf(pj[c], v4.y, o.y);
        o.z = __builtin_fmaf(pj[c], v4.z, o.z);
        o.w = __builtin_fmaf(pj[c], v4.w, o.w);
      }
      if (g + 3 < 8) load_v(vq[g % 3], g + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    o.x = __builtin_fmaf(o.x, r.uv, pv4.x * P);
    o.y = __builtin_fmaf(o.y, r.uv, pv4.y * P);
    o.z = __builtin_fmaf(o.z, r.uv, pv4.z * P);
    o.w = __builtin_fmaf(o.w, r.uv, pv4.w * P);
    *(SLIMT_LDS int *)(r.arow + 4 * lane) =
        pack4(quantize1_byte(o.x, r.aq_o), quantize1_byte(o.y, r.aq_o), quantize1_byte(o.z, r.aq_o), quantize1_byte(o.w, r.aq_o));
  } else if (LONG && DH == 32 && S <= 128) {
    attention_row_long<D, DH, KV_AUX>(r, lane);
  } else if (DH == 64 && S <= 32) {
    // d_head 64 ("base"): one head per score pass, lane = key (both wave halves hold the
    // same 32 keys, so the 32-lane reductions serve both); all heads' probabilities go to
    // LDS, then V is streamed as whole rows (D = 512 floats = two 16-byte loads per lane:
    // columns 4 l .. 4 l + 3 of head l / 16 and of head 4 + l / 16), 4 rows per group with
    // the next group in flight: 64 V loads per sentence and layer instead of 256.
    static_assert(DH != 64 || D == 512, "a V row is two 16-byte loads per lane");
    const int j = lane & 31;
    const int jc = j < S ? j : S - 1;
    const float mask = (1.0f - (j < len ? 1.0f : 0.0f)) * minus_inf;
    const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * D) * 4u);
    const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(__builtin_amdgcn_readfirstlane(lenf) * D) * 4u);  // padding is not fetched
    const int koff = j < lenf ? jc * 16 : kPastDescriptor;  // [head][dh/4][S][4] floats
    const int voff = lane * 16;
    f4 vq[3][4];  // V rows in flight: three groups of two rows (two halves each)
    auto load_v = [&](f4(&vv)[4], int g) {  // rows 2 g, 2 g + 1
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        vv[2 * i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, ((2 * g + i) * D) * 4, KV_AUX));
        vv[2 * i + 1] = __builtin_bit_cast(
            f4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, ((2 * g + i) * D + D / 2) * 4, KV_AUX));
      }
    };
#pragma unroll 1
    for (int h = 0; h < H; ++h) {
      f4 k4[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        k4[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                           rk, koff, ((h * (DH / 4) + i) * S * 4) * 4, KV_AUX));
      const float ch = wave_sum(r.qrow[h * DH + lane] * r.pbk[h * DH + lane]);  // c_h (the hoisted order, unpack24f below)
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f4 q4 = *(lcf4_ptr)(r.qrow + h * DH + 4 * i);
        s = __builtin_fmaf(q4.x, k4[i].x, s);
        s = __builtin_fmaf(q4.y, k4[i].y, s);
        s = __builtin_fmaf(q4.z, k4[i].z, s);
        s = __builtin_fmaf(q4.w, k4[i].w, s);
      }
      s = __builtin_fmaf(s, r.uk, ch);
      if (r.alpha != 1.0f) s = r.alpha * s;
      s = s + mask;
      if (j >= S) s = lowest;
      const float m = half_max(s);
      const float e = j < S ? exp_p(s - m) : 0.0f;
      const float sum = half_sum(e);
      const float p = e / sum;  // keys >= S: exactly 0
      const float ps = half_sum(p);  // P_h
      if (lane < 32) {
        if (r.attn && j < S) r.attn[(size_t)h * S + j] = p;
        if (r.align && h == 0 && j < len) r.align[j] = p;
        r.pbuf[h * 32 + j] = p;
        if (lane == 0) r.hsum[h] = ps;
      }
    }
    load_v(vq[0], 0);  // (held across the head loop, a group of V rows would be spilled)
    load_v(vq[1], 1);
    load_v(vq[2], 2);
    __builtin_amdgcn_sched_barrier(0);
    const int ph0 = (lane >> 4) * 32, ph1 = (4 + (lane >> 4)) * 32;
    f4 o0 = {0.0f, 0.0f, 0.0f, 0.0f}, o1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f4(&cur)[4] = vq[g % 3];
#pragma unroll
      for (int c = 0; c < 2; ++c) {  // keys >= S: p == 0, fma(0, v, o) == o
        const float pa = r.pbuf[ph0 + 2 * g + c], pb = r.pbuf[ph1 + 2 * g + c];
        const f4 v0 = cur[2 * c], v1 = cur[2 * c + 1];
        o0.x = __builtin_fmaf(pa, v0.x, o0.x);
        o0.y = __builtin_fmaf(pa, v0.y, o0.y);
        o0.z = __builtin_fmaf(pa, v0.z, o0.z);
        o0.w = __builtin_fmaf(pa, v0.w, o0.w);
        o1.x = __builtin_fmaf(pb, v1.x, o1.x);
        o1.y = __builtin_fmaf(pb, v1.y, o1.y);
        o1.z = __builtin_fmaf(pb, v1.z, o1.z);
        o1.w = __builtin_fmaf(pb, v1.w, o1.w);
      }
      if (g + 3 < 16) load_v(vq[g % 3], g + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const float P0 = r.hsum[lane >> 4], P1 = r.hsum[4 + (lane >> 4)];
      const f4 pv0 = *(gcf4_ptr)(r.pbv + 4 * lane), pv1 = *(gcf4_ptr)(r.pbv + D / 2 + 4 * lane);
      o0.x = __builtin_fmaf(o0.x, r.uv, pv0.x * P0);
      o0.y = __builtin_fmaf(o0.y, r.uv, pv0.y * P0);
      o0.z = __builtin_fmaf(o0.z, r.uv, pv0.z * P0);
      o0.w = __builtin_fmaf(o0.w, r.uv, pv0.w * P0);
      o1.x = __builtin_fmaf(o1.x, r.uv, pv1.x * P1);
      o1.y = __builtin_fmaf(o1.y, r.uv, pv1.y * P1);
      o1.z = __builtin_fmaf(o1.z, r.uv, pv1.z * P1);
      o1.w = __builtin_fmaf(o1.w, r.uv, pv1.w * P1);
    }
    *(SLIMT_LDS int *)(r.arow + 4 * lane) =
        pack4(quantize1_byte(o0.x, r.aq_o), quantize1_byte(o0.y, r.aq_o), quantize1_byte(o0.z, r.aq_o), quantize1_byte(o0.w, r.aq_o));
    *(SLIMT_LDS int *)(r.arow + D / 2 + 4 * lane) =
        pack4(quantize1_byte(o1.x, r.aq_o), quantize1_byte(o1.y, r.aq_o), quantize1_byte(o1.z, r.aq_o), quantize1_byte(o1.w, r.aq_o));
  } else {
    // generic: one head per pass, keys lane and lane + 64
    const int j0 = lane < S ? lane : S - 1;
    const int j1 = (lane + 64) < S ? (lane + 64) : S - 1;
    const int dc = lane < DH ? lane : DH - 1;
    const float mask0 = (1.0f - (lane < len ? 1.0f : 0.0f)) * minus_inf;
    const float mask1 = (1.0f - ((lane + 64) < len ? 1.0f : 0.0f)) * minus_inf;
    for (int h = 0; h < H; ++h) {
      // c_h: one column of the head per lane, the canonical butterfly (lanes past the head hold +0)
      const float ch = wave_sum(lane < DH ? r.qrow[h * DH + dc] * r.pbk[h * DH + dc] : 0.0f);
      float s0 = 0.0f, s1 = 0.0f;
      for (int k = 0; k < DH; ++k) {
        const float qk = r.qrow[h * DH + k];
        gcf_ptr kc = r.kl + ((size_t)(h * (DH / 4) + (k >> 2)) * S) * 4 + (k & 3);
        s0 = __builtin_fmaf(qk, kc[(size_t)j0 * 4], s0);
        s1 = __builtin_fmaf(qk, kc[(size_t)j1 * 4], s1);
      }
      s0 = __builtin_fmaf(s0, r.uk, ch);
      s1 = __builtin_fmaf(s1, r.uk, ch);
      if (r.alpha != 1.0f) {
        s0 = r.alpha * s0;
        s1 = r.alpha * s1;
      }
      s0 = s0 + mask0;
      s1 = s1 + mask1;
      if (lane >= S) s0 = lowest;
      if (lane + 64 >= S) s1 = lowest;
      const float m = wave_max(fmaxf(s0, s1));
      const float e0 = lane < S ? exp_p(s0 - m) : 0.0f;
      const float e1 = (lane + 64) < S ? exp_p(s1 - m) : 0.0f;
      const float sum = wave_sum(e0 + e1);
      const float p0 = e0 / sum, p1 = e1 / sum;
      const float P = wave_sum(p0 + p1);  // P_h
      if (r.attn) {
        gf_ptr ap = r.attn + (size_t)h * S;
        if (lane < S) ap[lane] = p0;
        if (lane + 64 < S) ap[lane + 64] = p1;
      }
      if (r.align && h == 0) {
        if (lane < len) r.align[lane] = p0;
        if (lane + 64 < len) r.align[lane + 64] = p1;
      }
      float o = 0.0f;
      for (int jj = 0; jj < S; ++jj) {
        const float pj = __shfl(jj < 64 ? p0 : p1, jj & 63, 64);
        o = __builtin_fmaf(pj, r.vl[(size_t)jj * D + h * DH + dc], o);
      }
      o = __builtin_fmaf(o, r.uv, r.pbv[h * DH + dc] * P);
      if (lane < DH) r.arow[h * DH + lane] = (char)quantize1_byte(o, r.aq_o);
    }
  }
}

// ---- the packed K/V cache (FusedDecodeArgs::kv24 / kv_fmt): one reader per shape of sentence, one unpack policy per form ----
#include "decode_attention_packed.inl.h"

}  // namespace

// Diagnostic phase stamps (100 MHz wall clock) of workgroup 0 at one chosen
// step; a.stamps == nullptr in normal runs (one uniform branch per phase).
#define SLIMT_STAMP(id)                                                                  \
  do {                                                                                   \
    if (a.stamps && m0 == 0 && tid == 0 && t == a.stamp_step)                            \
      a.stamps[(id)] = wall_clock64();                                                   \
  } while (0)

// Lane-derived LDS / global offsets are recomputed per layer and step from an
// opaque copy of the lane id: hoisted out of the step loop they only turn into
// kernel-lifetime registers that spill (a few VALU ops are cheaper than that).
#define SLIMT_PHASE_LANE                                 \
  int lane = lane0;                                      \
  asm volatile("" : "+v"(lane));                         \
  const int lr = lane & 15, lg = lane >> 4;              \
  (void)lr;                                              \
  (void)lg

// RT = row tiles per workgroup: 1 (16 sentences) or 2 (32 sentences). Every streamed weight
// fragment then feeds RT MFMAs (half the weight bytes per sentence and step at RT = 2), wave w
// owns sentences w and w + 16 in the row-wise phases.
// MID: 1 = sentences of 33..64 tokens over the packed cache (D = 256; attention_packed64 with Form24), 2 = of 65..128
// tokens (attention_packed128 with Form24; the SSRU cells then live in global memory: their 32 KB of LDS hold the
// [H][128] probabilities of every wave's sentence).
// SPW: sentences per workgroup, 16 (a full MFMA row tile), 8 or 4 -- for launches that would leave most of the
// chip idle (one batch of 256 at 16 sentences per workgroup runs on 16 of 256 CUs). The row tile stays 16 rows
// (rows SPW.. are zero operands whose results nobody reads: the matrix pipe is 6 % busy), every wave still
// streams its share of every weight, and only waves 0 .. SPW - 1 own a sentence in the row-wise phases
// (LayerNorm, attention, sampling), which then have a SIMD to themselves or share it with one wave instead of
// three. A sentence's arithmetic does not depend on which rows surround it, so results are identical.
// CL: cluster logits (output layers of 16k columns and more: the full vocabulary). With CL = 1 every workgroup streams the
// WHOLE output layer for its 16 sentences every step -- 8.2 MB at 32,000 columns, 76 of a loaded step's 134 us, sixteen
// times per batch of 256. With CL > 1, CL consecutive tiles (workgroups) form a cluster that shares it: each member takes
// 1 / CL of the column tiles for ALL the cluster's sentences, and two hand-overs through global memory per step carry the
// quantised input rows out (256 bytes per sentence) and each member's best (logit, column) per sentence back; the
// sentence's owner then takes the first maximum over the members' candidates -- columns ascend with the member index, so it
// is the reference's scan (Transformer.cc:287-298) whatever the split. The members wait for each other: the engine uses
// clusters only under the decoder admission (every admitted workgroup gets a CU without waiting for another decoder).
// KVI (packed-cache variants), the forms inlined: 20 = the narrow 20-bit form, the 24-bit form as the rare sentence's out-of-line
// fallback; 24 = the 24-bit form inlined and nothing else (launches whose caches are all 24-bit: K/V cache format 2,
// or a model the engine found mostly too wide for 20 bits -- there the out-of-line call would cost every sentence).
// KVI = 16: the tight 16-bit form inlined instead (RT = 1; kv_fmt == 2; the 20- and 24-bit ones out of line), for batches whose
// encoder was allowed it (engine.cpp, kv_tight_wanted); its centres take 2 D floats of LDS per layer.
// MG: the kernel serves merged launches (kernels.h, MergeOut; a.n_sub > 0). A template parameter, not a runtime test: compiled
// into the ONE set of kernels, the merged paths' scalar loads and selects cost the unmerged launches 1.5-3.5 % (B = 64, base,
// the f32 and the streamed-cache variants: same-box A/B against round 5's library, profiles/r06_merge_template_ab.txt) --
// these kernels run at the edge of their 128 registers. The 16-row tilings for sentences of up to 64 tokens have the merged
// twin; the engine gives a merged launch one of those.
template <bool MG, int KSD, int KSF, int DH, bool LONG, bool NT, int RT = 1, bool KV24 = false, int MID = 0, int SPW = 16, int CL = 1,
          int KVI = 20>
__global__ __launch_bounds__(1024) void decode_fused_kernel(FusedDecodeArgs a) {
  static_assert(!MG || (RT == 1 && CL == 1 && MID <= 1), "merged launches: the 16-row tilings, sentences of up to 64 tokens");
  constexpr bool KV20 = KVI != 24;
  static_assert(KVI == 24 || KVI == 20 || (KVI == 16 && KV24 && (KSD == 4 || KSD == 8) && (RT == 1 || (KSD == 4 && MID == 0)) && CL == 1),
                "16-bit form: sentences of up to 128 tokens at D = 256 (the 32-sentence tiling: up to 32), up to 32 at D = 512");
  static_assert(CL == 1 || (CL <= 4 && RT == 1 && SPW == 16 && MID == 0 && !LONG), "cluster logits: the 16-sentence tilings");
  static_assert(SPW == 16 || ((SPW == 8 || SPW == 4) && RT == 1 && KV24), "fewer sentences per workgroup: the packed-cache, 16-row variants");
  static_assert(!KV24 || (((KSD == 4 && DH == 32) || (KSD == 8 && DH == 64)) && !LONG),
                "the packed K/V cache: D = 256 / d_head 32 or D = 512 / d_head 64, S <= 32 (64 / 128 with MID 1 / 2)");
  static_assert(!MID || (KV24 && KSD == 4 && RT == 1), "33..128-token sentences: the D = 256 packed cache, 16 rows");
  constexpr int PBW = MID == 2 ? 1024 : MID == 1 ? 512 : 256;  // floats of attention scratch per wave ([H][S])
  constexpr int KVC = KSD == 8 ? 4 : 2;  // constant vectors per layer in LDS (decode_attention_packed.inl.h: Form24, attention_row24_64)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 64 * KSD, F = 64 * KSF;
  constexpr int R = 16 * RT;    // rows of the operand tiles per workgroup
  constexpr int RS = SPW < 16 ? SPW : R;  // sentences per workgroup
  constexpr int LDF = D + 4;    // f32 row stride
  // int8 operand rows: + 32 bytes -- a fragment read (ds_read_b128: 16 rows x 16 B per hardware lane group)
  // then covers all 64 banks; with + 16, rows lr and lr + 1 of neighbouring lane groups shared banks
  // (every A-fragment read 2-way conflicted; tools/lds_conflicts.py)
  constexpr int LDA = D + 32;   // K = D
  constexpr int LDA3 = F + 32;  // K = F
  // column tiles per wave where every wave has the same number (else 0: rolled loops)
  constexpr int NT_D = (D / 16) % NW == 0 ? (D / 16) / NW : 0;
  constexpr int NT_F1 = (F / 16) % NW == 0 ? (F / 16) / NW : 0;
  // chunks of weight fragments in flight per wave: with two row tiles a chunk feeds twice the
  // MFMAs and epilogues (the same cover from fewer chunks) and the A fragments take 16 more registers
  constexpr int NB_FFN = RT > 1 ? 2 : SLIMT_NB_FFN;
  constexpr int NB_OUT = RT > 1 ? (KSD > 4 ? 2 : 3) : SLIMT_NB_OUT;
  const int tid = threadIdx.x, lane0 = tid & 63, lane = lane0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, S = a.S, H = D / DH, Ld = a.Ld;
  const int n_sub = MG ? a.n_sub : 0;  // (0 at compile time in the unmerged kernels: every merged path folds away)
  const bool row_wave = SPW == 16 || wave < SPW;  // this wave owns a sentence in the row-wise phases

  // 16 rows x 512 ("base") or 32 rows x 256 do not fit the full layout in 160 KiB: the
  // pre-LN buffer then aliases hs (every pre-LN write reads at most the same element of hs,
  // LayerNorm runs from registers) and the SSRU cells live in global memory (a.cells,
  // [Ld][B][D]; 1 KiB per sentence and layer, read and written once per step).
  constexpr bool LEAN = KSD * RT > 4;
  constexpr bool CELLS_GLOBAL = LEAN || MID == 2;  // SSRU cells in a.cells instead of LDS
  float *xs = reinterpret_cast<float *>(smem);  // layer input rows (post-LN); reused for q
  float *hs = xs + R * LDF;                     // h / o rows (post-LN residual source)
  float *pre = LEAN ? hs : hs + R * LDF;        // pre-LN accumulation
  float *cs = pre + R * LDF;                    // SSRU cells [Ld][R][D] (full layout only)
  char *A1 = reinterpret_cast<char *>(cs + (CELLS_GLOBAL ? 0 : (size_t)Ld * R * D));
  char *A2 = A1 + R * LDA;
  char *A3 = A2 + R * LDA;
  float *red_v = reinterpret_cast<float *>(A3 + R * LDA3);  // [NW][R]
  int *red_i = reinterpret_cast<int *>(red_v + NW * R);
  int *flags = red_i + NW * R;  // [0] = number of finished sentences of this tile
  float *pbufs = reinterpret_cast<float *>(flags + 16);  // [NW][256] attention scratch
  float *kvpb = pbufs + NW * PBW;  // KV24: [Ld][K pb, V pb][D], or at D = 512 [Ld][K pb, K c127, V pb, V c127][D]
  // LayerNorm scale / bias of every layer in LDS ([Ld][rnn, attn, ffn][scale, bias][D]) where it fits:
  // fetched per phase they are vector-memory loads that return IN ORDER behind whatever the wave
  // asked for before -- a weight prefetch in front of a LayerNorm would stall it by its whole transfer
  constexpr bool LN_LDS = KSD == 4 && RT == 1 && MID == 0;  // (MID: the LDS goes to the wider attention scratch)
  float *lnc = kvpb + (KV24 ? Ld * KVC * D : 0);
  const bool ln_lds = LN_LDS && a.ln_in_lds;  // (the launcher: only where the 160 KiB allow it)
  float *kvc127 = lnc + (ln_lds ? Ld * 6 * D : 0);  // KVI == 16: [Ld][K, V][D] float(centre) (attention_packed32 with Form16)

  // Which R sentences? With a ticket counter the grid is over-subscribed and the first
  // workgroups to START claim the tiles; the rest leave at once. A workgroup needs a whole
  // CU, and the hardware binds a workgroup to a shader engine when the kernel is dispatched,
  // not when a CU frees up: with exactly one workgroup per tile, tiles waited for a CU on
  // "their" engine while CUs of other engines sat idle (20 % of the CU time under the
  // 16-worker load, tools/occupancy_trace.py, tools/probes/mix_probe.hip). Tiles are
  // independent, so who runs which changes nothing in the results.
  int tile = blockIdx.x;
  if (a.ticket) {
    if (tid == 0) {
      if (a.home_mask) {  // XCD-affine claim (kernels.h)
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        const bool home = (a.home_mask >> (xcc & 31)) & 1u;
        const unsigned tiles = (unsigned)((B + RS - 1) / RS);
        unsigned long long seen = __hip_atomic_load(a.xstate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int mine = 0x7fffffff;
        for (;;) {  // bounded: every failed exchange is another candidate's arrival (at most xgrid of them)
          const unsigned arrived = (unsigned)(seen >> 32) - a.xarr_base;
          const unsigned claimed = (unsigned)seen - a.xclaim_base;
          const unsigned to_come = a.xgrid - (arrived + 1u);
          const unsigned open_tiles = tiles - (claimed < tiles ? claimed : tiles);
          const bool take = open_tiles > 0 && (home || to_come < open_tiles);
          // the two halves are counters of their own (modulo 2^32 each, like their bases): composed, not added as
          // one 64-bit number, so that the claimed count's wrap-around never carries into the arrivals
          const unsigned long long want =
              ((unsigned long long)((unsigned)(seen >> 32) + 1u) << 32) | (unsigned long long)((unsigned)seen + (take ? 1u : 0u));
          if (__hip_atomic_compare_exchange_strong(a.xstate, &seen, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)) {
            mine = take ? (int)claimed : 0x7fffffff;
            break;
          }
        }
        flags[1] = mine;
      } else {
        flags[1] = (int)(atomicAdd(a.ticket, 1u) - a.ticket_base);
      }
    }
    __syncthreads();
    tile = flags[1];
    if ((unsigned)tile >= (unsigned)((B + RS - 1) / RS)) return;
  }
  const int m0 = tile * RS;
  // merged launch (kernels.h, MergeOut). Aligned (every sub-batch starts at a multiple of the tile: each has its own output
  // layer): this tile lies inside sub-batch sj, whose sentences are the global ones first .. Bv - 1, the global ones up to
  // the next sub-batch are holes, and a tile of nothing but holes leaves. Dense (one output layer for all: sub-batches
  // follow each other without holes, a tile may hold sentences of several): Bv = B. Either way every SENTENCE finds its
  // own sub-batch below (the caller's arrays, its padded length); the workspace (K/V cache, its form bytes, SSRU cells)
  // is indexed by the global sentence.
  const int sj = n_sub ? __builtin_amdgcn_readfirstlane(merge_find(a.sub, n_sub, m0)) : 0;
  const int Bv = n_sub && !a.sub_dense ? a.sub[sj].first + a.sub[sj].n : B;
  if (m0 >= Bv) return;
  if (tid == 0) occ_trace_event(a.trace, 1, 0);
  // my cluster (CL > 1): tiles cl_first .. cl_first + cl_n - 1 (the last cluster of a batch may be short), me = member cl_m
  const int n_tiles_b = (B + RS - 1) / RS;
  const int cl_first = tile / CL * CL;
  const int cl_n = CL == 1 ? 1 : (n_tiles_b - cl_first < CL ? n_tiles_b - cl_first : CL);
  const int cl_m = tile - cl_first;
  if (CL > 1 && tid == 0) {
    flags[2] = 0;  // every member of my cluster has finished all its sentences
    flags[3] = 0;  // a cluster wait of this launch has timed out
  }

  // per-sentence state of rows wave + 16 rr, owned by wave `wave` (uniform within the wave)
  int bq[RT], len[RT];
  int sw[RT];  // this wave's sentences' sub-batches (merged launches); what follows from them is re-read where it is needed --
  // once per step at most -- instead of living in scalar registers across the loop (five more per sentence spilled 11
  // vector registers in the headline's instantiation)
#define SLIMT_SW(rr) sw[rr]
#define SLIMT_SUB_FIRST(rr) (n_sub ? a.sub[SLIMT_SW(rr)].first : 0)
#define SLIMT_SUB_TMAX(rr) (n_sub ? a.sub[SLIMT_SW(rr)].Tmax : a.Tmax)
#define SLIMT_SUB_S(rr) (n_sub ? a.sub[SLIMT_SW(rr)].S : S)
  bool live[RT], finished[RT];
  uint32_t n_out[RT];
  // Sentences of 33..128 tokens in a 4-sentence workgroup: the waves that own no sentence share the score passes of the
  // sentence in slot wave % 4 with its owner (heads wave / 4 and + 4; attention_packed128<PART>): they know its index, length
  // and cache forms, and nothing else of it. Four sentences per workgroup is what the engine picks while CUs would idle
  // (adaptive rows): there the shared passes are worth 9 % (one to four contexts of 64 such sentences: 13.2 -> 12.0 ms per
  // batch). At 8 per workgroup -- what it picks under the 20-worker load, where these sentences wait for their K/V bytes, not
  // for their instructions -- sharing cost 2 %: not shared there (profiles/r06_shared_score_passes.txt).
  constexpr bool SHARE = !MG && MID >= 1 && SPW == 4;
  const bool sharer = SHARE && wave >= SPW;
#pragma unroll
  for (int rr = 0; rr < RT; ++rr) {
    bq[rr] = m0 + 16 * rr + (sharer ? (wave & (SPW - 1)) : wave);
    live[rr] = row_wave && bq[rr] < Bv;
    // this sentence's sub-batch: its first global sentence, the row length of its outputs (its own padded length's
    // limit, Model.cc:159-161), the width of its alignment rows, the steps it runs at most
    sw[rr] = n_sub && live[rr] ? __builtin_amdgcn_readfirstlane(merge_find(a.sub, n_sub, bq[rr])) : 0;
    len[rr] = live[rr] || (sharer && bq[rr] < Bv)
                  ? checked_length(n_sub ? a.sub[sw[rr]].lengths[bq[rr] - SLIMT_SUB_FIRST(rr)] : a.lengths[bq[rr]], S)
                  : 0;
    finished[rr] = !live[rr];
    n_out[rr] = 0;
  }
  const int valid_rows = (Bv - m0) < RS ? (Bv - m0) : RS;
  // the form of this wave's sentences' caches: bit l = 24-bit, bit 8 + l = the tight 16-bit form, neither = 20-bit
  // (kernels.h, kv_fmt: 1 / 2 / 0)
  unsigned kv_wide[RT];
#pragma unroll
  for (int rr = 0; rr < RT; ++rr) {
    kv_wide[rr] = 0xffu;
    if constexpr (KV24 && KV20) {
      if (a.kv_fmt && (live[rr] || (sharer && bq[rr] < Bv))) {
        unsigned w = 0;
        for (int l = 0; l < Ld; ++l) {
          const unsigned f = a.kv_fmt[(size_t)l * B + bq[rr]];
          w |= (unsigned)(f == 1) << l | (unsigned)(f == 2) << (8 + l);
        }
        kv_wide[rr] = __builtin_amdgcn_readfirstlane(w);
      }
    }
  }

  // start_states, Transformer.cc:78-85
  if constexpr (CELLS_GLOBAL) {
    for (int l = 0; l < Ld; ++l)
      for (int i = tid; i < valid_rows * D; i += 1024) a.cells[((size_t)l * B + m0) * D + i] = 0.0f;
  } else {
    for (int i = tid; i < Ld * R * D; i += 1024) cs[i] = 0.0f;
  }
  if (tid == 0) flags[0] = 0;
  if constexpr (SPW < 16) {  // operand rows no wave owns: defined (zero) for the whole loop
    for (int i = tid; i < (2 * R * LDA + R * LDA3) / 4; i += 1024) reinterpret_cast<int *>(A1)[i] = 0;
  }
  if (ln_lds) {
    for (int i = tid; i < Ld * 6 * D; i += 1024) {
      const FusedLayerW &Lw = a.L[i / (6 * D)];
      const int v = (i / D) % 6, d = i % D;
      const float *src = v == 0 ? Lw.rnn_ln_s : v == 1 ? Lw.rnn_ln_b : v == 2 ? Lw.attn_ln_s : v == 3 ? Lw.attn_ln_b
                         : v == 4 ? Lw.ffn_ln_s : Lw.ffn_ln_b;
      lnc[i] = src[d];
    }
  }
  if constexpr (KVI == 16) {
    if constexpr (RT == 1) {
      for (int i = tid; i < Ld * 2 * D; i += 1024) kvc127[i] = (float)a.kv_centre[i / (2 * D)][(i / D) & 1][i % D];
    } else {  // (the K centres only: attention_packed32 with Form16's V pass reads its own from global memory)
      for (int i = tid; i < Ld * D; i += 1024) kvc127[i] = (float)a.kv_centre[i / D][0][i % D];
    }
  }
  if constexpr (KV24) {
    if constexpr (KVC == 2) {
      for (int i = tid; i < Ld * 2 * D; i += 1024) kvpb[i] = a.kv_pb[i / (2 * D)][(i / D) & 1][i % D];
    } else {
      for (int i = tid; i < Ld * 4 * D; i += 1024) {
        const int l = i / (4 * D), v = (i / D) & 3, d = i % D;
        kvpb[i] = (v & 1) ? (float)(__mul24(127, a.kv_cs[l][v >> 1][d]) * 256) : a.kv_pb[l][v >> 1][d];
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < RT; ++rr) {
    if (live[rr]) {  // outputs past a sentence's length read as zero (no memset launches)
      uint32_t *oi = n_sub ? a.sub[sw[rr]].out_ids : a.out_ids;
      float *al = n_sub ? a.sub[sw[rr]].align : a.align;
      const bool staged = (n_sub ? a.sub[sw[rr]].align_out : a.align_out) != nullptr;
      const int Tr = SLIMT_SUB_TMAX(rr), Sr = SLIMT_SUB_S(rr), fr = SLIMT_SUB_FIRST(rr);
      for (int i = lane; i < Tr; i += 64) oi[(size_t)(bq[rr] - fr) * Tr + i] = 0;
      if (al && !staged)
        for (int i = lane; i < Tr * Sr; i += 64) al[(size_t)(bq[rr] - fr) * Tr * Sr + i] = 0.0f;
    }
    // step-0 embedding: zeros * sqrt(D) + pos(0)  (Transformer.cc:138-144,160)
#pragma unroll
    for (int i = 0; i < KSD; ++i) {
      const float z = 0.0f * a.emb.sqrt_d;
      xs[(16 * rr + wave) * LDF + lane + 64 * i] = live[rr] ? z + a.emb.pos[lane + 64 * i] : 0.0f;
    }
  }
  __syncthreads();

  // the output layer's column count may live on the device (a shortlist generated there)
  PreparedWeight outw = a.out;
  if (n_sub) {  // this sub-batch's packed output layer: job strides behind the first (same K, multipliers, a_quant)
    const int job = a.sub[sj].job;
    outw.Wp = reinterpret_cast<const char *>(a.out.Wp) + (size_t)job * a.out_stride_wp;
    outw.colsum = reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.out.colsum) + (size_t)job * a.out_stride_cs);
    outw.cp4 = outw.colsum + epi_pair_offset_ints(a.sub[sj].N);
    outw.pb = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.out.pb) + (size_t)job * a.out_stride_pb);
    outw.N = a.out_n_dev ? (int)a.out_n_dev[job] : a.sub[sj].N;  // (a shortlist generated on the device: job j's count)
    outw.n_tiles = (outw.N + 15) / 16;
  } else if (a.out_n_dev) {
    outw.N = (int)*a.out_n_dev;
    outw.n_tiles = (outw.N + 15) / 16;
  }
  // logit[0] NaN for every row (a NaN in column 0's prepared bias or in the multiplier): the reference's scan starts
  // from logits[0] and only moves on `value > max`, so it stays at class 0 (Transformer.cc:287-298); the arg-max
  // below skips NaNs, so the rule is applied where the token is taken
  const bool nan0 = outw.pb[0] != outw.pb[0] || a.out.u != a.out.u;
  // (a dense tile runs while any of its sentences does: each ends at its own limit below; an aligned one has one limit)
  const int max_steps = n_sub && !a.sub_dense ? a.sub[sj].max_steps : a.max_steps;
  bool all_done = false;
  for (int t = 0; t < max_steps; ++t) {
    SLIMT_STAMP(0);
    if (a.stamps && m0 == 0 && tid == 0 && t == a.stamp_step) a.stamps[60] = clock64();
    for (int l = 0; l < Ld; ++l) {
      SLIMT_PHASE_LANE;
      const FusedLayerW &L = a.L[l];
      float *cl = CELLS_GLOBAL ? a.cells + ((size_t)l * B + m0) * D : cs + (size_t)l * R * D;
      const int sb = 1 + 10 * l;
      // ---- SSRU (Modules.cc:190-235) ------------------------------------
      // quantise x twice (Wf / W have their own multipliers)
      if (row_wave) {
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) {
          const int row = 16 * rr + wave;
#pragma unroll
          for (int i = 0; i < KSD; ++i) {
            const float v = xs[row * LDF + lane + 64 * i];
            A1[row * LDA + lane + 64 * i] = (char)quantize1_byte(v, L.rnn_f.a_quant);
            A2[row * LDA + lane + 64 * i] = (char)quantize1_byte(v, L.rnn_w.a_quant);
          }
        }
      }
      lds_barrier();
      // every sentence of this tile has emitted EOS (counted in the previous step's sampling
      // phase; checked here, behind the first barrier that follows it anyway)
      // (a cluster leaves together: its members need each other's share of the output layer until the last sentence ends)
      if (l == 0 && (CL > 1 ? flags[2] != 0 : flags[0] >= valid_rows)) {
        all_done = true;
        break;
      }
      SLIMT_STAMP(sb + 0);
      for (int tile = wave; tile < D / 16; tile += NW) {
        const rsrc_t rf = make_rsrc(L.rnn_f.Wp, (D / 16) * KSD * 1024u);
        const rsrc_t rw = make_rsrc(L.rnn_w.Wp, (D / 16) * KSD * 1024u);
        const rsrc_t rfc = make_rsrc(L.rnn_f.colsum, D * 4u), rfp = make_rsrc(L.rnn_f.pb, D * 4u);
        const rsrc_t rwc = make_rsrc(L.rnn_w.colsum, D * 4u), rwp = make_rsrc(L.rnn_w.pb, D * 4u);
        v4i bf[KSD], bw[KSD];
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
          bf[ks] = load_frag(rf, lane * 16, (tile * KSD + ks) * 1024);
          bw[ks] = load_frag(rw, lane * 16, (tile * KSD + ks) * 1024);
        }
        const int csf = __builtin_amdgcn_raw_buffer_load_b32(rfc, lr * 4, tile * 64, 0);
        const int csw = __builtin_amdgcn_raw_buffer_load_b32(rwc, lr * 4, tile * 64, 0);
        const float pbf = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfp, lr * 4, tile * 64, 0));
        const float pbw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rwp, lr * 4, tile * 64, 0));
        const int col = tile * 16 + lr;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          v4i accf = {0, 0, 0, 0}, accw = {0, 0, 0, 0};
#pragma unroll
          for (int ks = 0; ks < KSD; ++ks) {
            const v4i af = *reinterpret_cast<const v4i *>(A1 + (16 * rt + lr) * LDA + ks * 64 + lg * 16);
            const v4i aw = *reinterpret_cast<const v4i *>(A2 + (16 * rt + lr) * LDA + ks * 64 + lg * 16);
            accf = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[ks], accf, 0, 0, 0);
            accw = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw, bw[ks], accw, 0, 0, 0);
          }
          const Dequant4 f4v = dequant4(accf, csf, L.rnn_f.u, pbf), w4v = dequant4(accw, csw, L.rnn_w.u, pbw);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = 16 * rt + lg * 4 + r;
            const float f = f4v.v[r];
            const float wx = w4v.v[r];
            const bool cell_ok = !CELLS_GLOBAL || rl < valid_rows;  // global cells: rows of this batch only
            const float c = cell_ok ? cl[rl * D + col] : 0.0f;
            const float sg = sigmoid_p_select(f);  // highway(c, Wx, f), TensorOps.cc:674-678
            const float t1 = sg * c;
            const float t2 = (1.0f - sg) * wx;
            const float cn = t1 + t2;
            if (cell_ok) cl[rl * D + col] = cn;
            const float y = cn > 0.0f ? cn : 0.0f;
            pre[rl * LDF + col] = xs[rl * LDF + col] + y;  // Modules.cc:230
          }
        }
      }
      lds_barrier();
      SLIMT_STAMP(sb + 1);
      // h = LN(x + relu(c')), quantised for the Q projection (whose weight fragments are
      // requested now: their round trip runs under the LayerNorm and the barrier)
      Frags fq[1];
      stream_prologue<KSD, 1, NT_D>(L.q, wave, lane, fq);
      if (row_wave) {
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) {
          const int row = 16 * rr + wave;
          if (ln_lds)
            ln_row<KSD>(pre + row * LDF, lnc + (6 * l + 0) * D, lnc + (6 * l + 1) * D, a.eps, hs + row * LDF, A1 + row * LDA, L.q.a_quant, lane);
          else
            ln_row<KSD>(pre + row * LDF, L.rnn_ln_s, L.rnn_ln_b, a.eps, hs + row * LDF, A1 + row * LDA, L.q.a_quant, lane);
        }
      }
      lds_barrier();
      SLIMT_STAMP(sb + 2);
      // ---- cross-attention (Modules.cc:287-319) --------------------------
      // Q projection -> xs (x is dead until the end of the layer)
      stream_gemm_from<KSD, 1, NT_D, false, RT>(A1, LDA, L.q, wave, lane, fq,
                                           [&](int tile, int rt, const v4i &acc, int cq, float pb) {
                                             const int col = tile * 16 + lr;
                                             const Dequant4 q4v = dequant4(acc, cq, L.q.u, pb);
#pragma unroll
                                             for (int r = 0; r < 4; ++r)
                                               xs[(16 * rt + lg * 4 + r) * LDF + col] = q4v.v[r];
                                           });
      lds_barrier();
      SLIMT_STAMP(sb + 3);
      // SDPA over the cached K/V of this wave's sentence(s); output quantised into A1
      if constexpr (SHARE) {
        // 33..128 tokens, 4 sentences per workgroup: a sentence's score passes are half of its attention and 12 waves own no
        // sentence -- the four waves of slot wave % 4 take two heads each (PART 1), a barrier, the owner runs
        // the context pass (PART 2; per column a chain over all keys in order: nothing to share). The form the kernel is
        // built around only: a sentence-layer in a wider form is its owner's alone, through the fallback call, as before.
        const int row = wave & (SPW - 1);
        const int b = bq[0];
        const bool mine = live[0], part_of = mine || (sharer && b < Bv);
        AttnRow ar;
        bool shared = false;
        const bool kv_streams = 8 * l + (b & 7) >= a.kv_temporal_eighths;
        const lcf_ptr c0 = (lcf_ptr)(kvpb + (KVC * l) * D), c1 = (lcf_ptr)(kvpb + (KVC * l + 1) * D);
        const Form24 f24 = {a.kv_u256[l][0], a.kv_u256[l][1]};
        const Form20 f20 = {a.kv_u4096[l][0], a.kv_u4096[l][1]};
        const Form16<CentreLds, CentreLds> f16 = {a.kv_u[l][0], a.kv_u[l][1], {(lcf_ptr)(kvc127 + (2 * l) * D)}, {(lcf_ptr)(kvc127 + (2 * l + 1) * D)}};
#define SLIMT_ATTN(FN, FORM, PART)                                       \
  do {                                                                   \
    if (NT && kv_streams)                                                \
      FN<MID, SLIMT_KV_AUX_NT, decltype(FORM), PART, SPW>(ar, lane, c0, c1, FORM);   \
    else                                                                 \
      FN<MID, SLIMT_KV_AUX_KEEP, decltype(FORM), PART, SPW>(ar, lane, c0, c1, FORM); \
  } while (0)
#define SLIMT_ATTN_COLD(FORM)                                            \
  do {                                                                   \
    if (NT && kv_streams)                                                \
      attention_packed_cold<MID, SLIMT_KV_AUX_NT>(ar, lane, c0, c1, FORM); \
    else                                                                 \
      attention_packed_cold<MID, SLIMT_KV_AUX_KEEP>(ar, lane, c0, c1, FORM); \
  } while (0)
        const unsigned forms = kv_wide[0];
        const bool wide = KV20 && ((forms >> l) & 1u);
        const bool tight = KVI == 16 && ((forms >> (8 + l)) & 1u);
        if (part_of) {
          ar.kl = (gcf_ptr)((const SLIMT_GLOBAL char *)(a.kv + (size_t)(2 * l) * B * S * D) + (size_t)EXP_SENT(b) * S * D * 3);
          ar.vl = (gcf_ptr)((const SLIMT_GLOBAL char *)(a.kv + (size_t)(2 * l + 1) * B * S * D) + (size_t)EXP_SENT(b) * ((S + 3) & ~3) * D * 3);
          ar.qrow = (lcf_ptr)(xs + row * LDF);
          ar.arow = (lc_ptr)(A1 + row * LDA);
          ar.pbuf = (SLIMT_LDS float *)(pbufs + row * PBW);
          ar.hsum = (SLIMT_LDS float *)(red_v + row * R);
          ar.pbk = (gcf_ptr)a.kv_pb[l][0];
          ar.pbv = (gcf_ptr)a.kv_pb[l][1];
          ar.uk = a.kv_u[l][0];
          ar.uv = a.kv_u[l][1];
          ar.S = S;
          ar.len = len[0];
          ar.alpha = a.alpha;
          ar.aq_o = L.o.a_quant;
          ar.attn = (a.attn && (l + 1 == Ld)) ? (gf_ptr)(a.attn + (size_t)b * H * S) : (gf_ptr) nullptr;
          const bool want_align = mine && a.align && (l + 1 == Ld) && !finished[0] && ((int)n_out[0] < a.Tmax);
          ar.align = want_align ? (gf_ptr)(a.align + ((size_t)b * a.Tmax + n_out[0]) * S) : (gf_ptr) nullptr;
          shared = KVI == 24 ? true : KVI == 16 ? tight : !wide;  // the inlined form
          if (shared) {
            if constexpr (KVI == 24)
              SLIMT_ATTN(attention_packed, f24, 1);
            else if constexpr (KVI == 16)
              SLIMT_ATTN(attention_packed, f16, 1);
            else
              SLIMT_ATTN(attention_packed, f20, 1);
          } else if (mine) {
            if (!wide)
              SLIMT_ATTN_COLD(f20);
            else
              SLIMT_ATTN_COLD(f24);
          }
        } else if (row_wave) {
#pragma unroll
          for (int i = 0; i < KSD; ++i) A1[row * LDA + lane + 64 * i] = 0;
        }
        lds_barrier();  // every head's probabilities and P_h are in the sentence's scratch
        if (mine && shared) {
          if constexpr (KVI == 24)
            SLIMT_ATTN(attention_packed, f24, 2);
          else if constexpr (KVI == 16)
            SLIMT_ATTN(attention_packed, f16, 2);
          else
            SLIMT_ATTN(attention_packed, f20, 2);
        }
#undef SLIMT_ATTN
#undef SLIMT_ATTN_COLD
      } else {
#pragma unroll 1
      for (int rr = 0; rr < RT; ++rr) {
        const int row = 16 * rr + wave;
        const bool lv = rr ? live[RT - 1] : live[0];
        if (lv) {
          const int b = rr ? bq[RT - 1] : bq[0];
          const bool fin = rr ? finished[RT - 1] : finished[0];
          const int no = rr ? (int)n_out[RT - 1] : (int)n_out[0];
          // this sentence's cache of this layer: streamed past the caches (non-temporal), or kept
          // (FusedDecodeArgs::kv_temporal_eighths: layers first, then sentences by index mod 8)
          const bool kv_streams = 8 * l + (b & 7) >= a.kv_temporal_eighths;
          AttnRow ar;
          if constexpr (KV24) {  // same planes, 3 bytes per value
            ar.kl = (gcf_ptr)((const SLIMT_GLOBAL char *)(a.kv + (size_t)(2 * l) * B * S * D) + (size_t)EXP_SENT(b) * S * D * 3);
            ar.vl = (gcf_ptr)((const SLIMT_GLOBAL char *)(a.kv + (size_t)(2 * l + 1) * B * S * D) +
                              (size_t)EXP_SENT(b) * ((S + 3) & ~3) * D * 3);
          } else {
            ar.kl = (gcf_ptr)(a.kv + ((size_t)(2 * l) * B + EXP_SENT(b)) * S * D);
            ar.vl = (gcf_ptr)(a.kv + ((size_t)(2 * l + 1) * B + EXP_SENT(b)) * S * D);
          }
          ar.qrow = (lcf_ptr)(xs + row * LDF);
          ar.arow = (lc_ptr)(A1 + row * LDA);
          ar.pbuf = (SLIMT_LDS float *)(pbufs + wave * PBW);
          ar.hsum = (SLIMT_LDS float *)(red_v + wave * R);  // (the arg-max scratch: idle until the output layer)
          ar.pbk = (gcf_ptr)a.kv_pb[l][0];
          ar.pbv = (gcf_ptr)a.kv_pb[l][1];
          ar.uk = a.kv_u[l][0];
          ar.uv = a.kv_u[l][1];
          ar.S = S;
          ar.len = rr ? len[RT - 1] : len[0];
          ar.alpha = a.alpha;
          ar.aq_o = L.o.a_quant;
          ar.attn = (a.attn && (l + 1 == Ld)) ? (gf_ptr)(a.attn + (size_t)b * H * S) : (gf_ptr) nullptr;
          const int swr = rr ? sw[RT - 1] : sw[0];
          const int fwr = n_sub ? a.sub[swr].first : 0, Tr = n_sub ? a.sub[swr].Tmax : a.Tmax, Sr = n_sub ? a.sub[swr].S : S;
          float *al = n_sub ? a.sub[swr].align : a.align;
          const bool want_align = al && (l + 1 == Ld) && !fin && (no < Tr);
          ar.align = want_align ? (gf_ptr)(al + ((size_t)(b - fwr) * Tr + no) * Sr) : (gf_ptr) nullptr;
          if constexpr (KV24) {
            // the packed cache (decode_attention_packed.inl.h): one reader per shape of sentence, one unpack policy per form.
            // This sentence-layer's form is uniform in the wave (kv_wide); the form the kernel was built around (KVI) is
            // inlined, a wider one is the rare sentence-layer's out-of-line fallback.
            constexpr int SHAPE = KVC == 4 ? 3 : MID;  // 0: up to 32 tokens, 1: 33..64, 2: 65..128 (D = 256); 3: D = 512
            const lcf_ptr c0 = (lcf_ptr)(kvpb + (KVC * l) * D), c1 = (lcf_ptr)(kvpb + (KVC * l + 1) * D);
            const Form24 f24 = {a.kv_u256[l][0], a.kv_u256[l][1]};
            const Form20 f20 = {a.kv_u4096[l][0], a.kv_u4096[l][1]};
#define SLIMT_ATTN(FN, FORM)                                                 \
  do {                                                                       \
    if (NT && kv_streams)                                                    \
      FN<SHAPE, SLIMT_KV_AUX_NT>(ar, lane, c0, c1, FORM);                    \
    else                                                                     \
      FN<SHAPE, SLIMT_KV_AUX_KEEP>(ar, lane, c0, c1, FORM);                  \
  } while (0)
            if constexpr (!KV20) {
              SLIMT_ATTN(attention_packed, f24);
            } else {
              const unsigned forms = rr ? kv_wide[RT - 1] : kv_wide[0];
              const bool wide = (forms >> l) & 1u;
              const bool tight = KVI == 16 && ((forms >> (8 + l)) & 1u);
              if (tight) {
                if constexpr (KVI == 16 && RT == 1) {
                  const Form16<CentreLds, CentreLds> f16 = {ar.uk, ar.uv, {(lcf_ptr)(kvc127 + (2 * l) * D)}, {(lcf_ptr)(kvc127 + (2 * l + 1) * D)}};
                  SLIMT_ATTN(attention_packed, f16);
                } else if constexpr (KVI == 16) {  // 32 sentences: LDS holds the K centres only ([Ld][D])
                  const Form16<CentreLds, CentreGlobal> f16 = {ar.uk, ar.uv, {(lcf_ptr)(kvc127 + l * D)}, {a.kv_centre[l][1]}};
                  SLIMT_ATTN(attention_packed, f16);
                }  // (the other kernels never meet the form: the launcher refuses a tight batch, the engine never sends one)
              } else if (!wide) {
                if constexpr (KVI == 16)  // (the rare sentence-layer past the calibrated centres' int16)
                  SLIMT_ATTN(attention_packed_cold, f20);
                else
                  SLIMT_ATTN(attention_packed, f20);
              } else {
                SLIMT_ATTN(attention_packed_cold, f24);
              }
            }
#undef SLIMT_ATTN
          } else if (NT && kv_streams)
            attention_row<D, DH, LONG, 2>(ar, lane);
          else
            attention_row<D, DH, LONG, 0>(ar, lane);
        } else if (row_wave) {
#pragma unroll
          for (int i = 0; i < KSD; ++i) A1[row * LDA + lane + 64 * i] = 0;
        }
      }
      }
      Frags fo[1];  // this wave's sentences are done: O's fragments travel under the barrier wait
      stream_prologue<KSD, 1, NT_D>(L.o, wave, lane, fo);
      lds_barrier();
      SLIMT_STAMP(sb + 4);
      // O projection + residual h (Modules.cc:308-314)
      stream_gemm_from<KSD, 1, NT_D, false, RT>(A1, LDA, L.o, wave, lane, fo,
                                           [&](int tile, int rt, const v4i &acc, int co, float pb) {
                                             const int col = tile * 16 + lr;
                                             const Dequant4 o4v = dequant4(acc, co, L.o.u, pb);
#pragma unroll
                                             for (int r = 0; r < 4; ++r) {
                                               const int rl = 16 * rt + lg * 4 + r;
                                               pre[rl * LDF + col] = o4v.v[r] + hs[rl * LDF + col];
                                             }
                                           });
      lds_barrier();
      SLIMT_STAMP(sb + 5);
      // FFN1's first chunks: ahead of the LayerNorm where its constants live in LDS (the LayerNorm
      // then waits for no vector-memory load), else ahead of the barrier behind it
      Frags f1[NB_FFN];
      if constexpr (LN_LDS) stream_prologue<KSD, NB_FFN, NT_F1>(L.ffn1, wave, lane, f1);
      if (row_wave) {
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) {
          const int row = 16 * rr + wave;
          if (ln_lds)
            ln_row<KSD>(pre + row * LDF, lnc + (6 * l + 2) * D, lnc + (6 * l + 3) * D, a.eps, hs + row * LDF, A1 + row * LDA, L.ffn1.a_quant, lane);
          else
            ln_row<KSD>(pre + row * LDF, L.attn_ln_s, L.attn_ln_b, a.eps, hs + row * LDF, A1 + row * LDA, L.ffn1.a_quant, lane);
        }
      }
      if constexpr (!LN_LDS) stream_prologue<KSD, NB_FFN, NT_F1>(L.ffn1, wave, lane, f1);
      lds_barrier();
      SLIMT_STAMP(sb + 6);
      // ---- FFN (Modules.cc:251-257) ----------------------------------------
      stream_gemm_from<KSD, NB_FFN, NT_F1, false, RT>(
          A1, LDA, L.ffn1, wave, lane, f1, [&](int tile, int rt, const v4i &acc, int c1, float pb) {
            const int col = tile * 16 + lr;
            const Dequant4 h4v = dequant4(acc, c1, L.ffn1.u, pb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // relu, then PrepareA: for aq > 0, clamp(rint(max(v, 0) aq), -127, 127) == rint(clamp(v aq, 0, 127)) for every
              // float v (the product keeps the sign; a NaN falls to the lower bound of either form) -- the relu rides in the
              // clamp, the rounding in the magic add (quantize1_byte)
              const float v = h4v.v[r];
              const float tq = __builtin_amdgcn_fmed3f(v * L.ffn2.a_quant, 0.0f, 127.0f);
              A3[(16 * rt + lg * 4 + r) * LDA3 + col] = (char)__float_as_int(tq + 12582912.0f);
            }
          });
      Frags f2[NB_FFN];  // FFN2's first chunks: requested as this wave's FFN1 tiles are done
      stream_prologue<KSF, NB_FFN, NT_D>(L.ffn2, wave, lane, f2);
      lds_barrier();
      SLIMT_STAMP(sb + 7);
      stream_gemm_from<KSF, NB_FFN, NT_D, false, RT>(
          A3, LDA3, L.ffn2, wave, lane, f2, [&](int tile, int rt, const v4i &acc, int c2, float pb) {
            const int col = tile * 16 + lr;
            const Dequant4 y4v = dequant4(acc, c2, L.ffn2.u, pb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int rl = 16 * rt + lg * 4 + r;
              pre[rl * LDF + col] = y4v.v[r] + hs[rl * LDF + col];
            }
          });
      lds_barrier();
      SLIMT_STAMP(sb + 8);
      // next layer's input; after the last layer: quantised for the logits
      if (row_wave) {
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) {
          const int row = 16 * rr + wave;
          if (ln_lds)
            ln_row<KSD>(pre + row * LDF, lnc + (6 * l + 4) * D, lnc + (6 * l + 5) * D, a.eps, xs + row * LDF,
                        (l + 1 == Ld) ? A1 + row * LDA : nullptr, a.out.a_quant, lane);
          else
            ln_row<KSD>(pre + row * LDF, L.ffn_ln_s, L.ffn_ln_b, a.eps, xs + row * LDF,
                        (l + 1 == Ld) ? A1 + row * LDA : nullptr, a.out.a_quant, lane);
        }
      }
      // Between layers no barrier: the next phase (the SSRU's quantisation) reads and writes
      // only this wave's own rows; the barrier after it covers both. (After the last layer: below.)
      SLIMT_STAMP(sb + 9);
    }
    if (all_done) break;
    // ---- output layer + greedy sample (Transformer.cc:176-182,279-339) ----
    SLIMT_PHASE_LANE;
    int cl_ix[RT];  // CL > 1: the sampled column of this wave's sentence, from the cluster's candidates
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) cl_ix[rr] = 0x7fffffff;
    if constexpr (CL > 1) {
      constexpr int CR = 16 * CL;         // sentences (rows) of a full cluster
      constexpr int kCoh = 17;            // sc0 | sc1: exchanged data is written through and read past the caches
      char *CA = reinterpret_cast<char *>(xs);  // [CR][LDA] int8: the cluster's input rows (x / h / pre-LN rows are dead here)
      float *cred_v = reinterpret_cast<float *>(CA + CR * LDA);  // [NW][CR] every wave's candidates
      int *cred_i = reinterpret_cast<int *>(cred_v + NW * CR);
      static_assert((size_t)CR * LDA + 2 * NW * CR * 4 <= (LEAN ? 2 : 3) * (size_t)R * LDF * 4, "the cluster's rows + candidates fit the f32 row buffers");
      const rsrc_t ract = make_rsrc(a.cl_act, (unsigned)n_tiles_b * 16u * (unsigned)D);
      const rsrc_t rpart = make_rsrc(a.cl_part, (unsigned)n_tiles_b * (unsigned)(CR + 1) * 8u);
      // a cluster barrier: every wave's stores are out, one release, one arrival; then a bounded wait for the others'
      auto cluster_sync = [&](unsigned target) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          unsigned *ctr = a.cl_sync + tile / CL;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned spin = 0;
          // (flags[3]: a wait of this launch has run out already -- the batch has failed, dev_error says so; the waits
          // behind it do not spin their two seconds again: ADVICE r05)
          while (!flags[3] && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spin > (1u << 22)) {  // ~2 s: a member never arrived (it cannot under the admission the engine asks for)
              if (a.dev_error) __hip_atomic_store(a.dev_error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              flags[3] = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(4);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
      };
      lds_barrier();  // the last LayerNorm's quantised rows are in A1
      SLIMT_STAMP(43);
      // (1) my 16 rows out
      for (int i = tid; i < 16 * (D / 16); i += 1024) {
        const int row = i / (D / 16), ch = i % (D / 16);
        const v4i v = *reinterpret_cast<const v4i *>(A1 + row * LDA + ch * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, ract, row * D + ch * 16, tile * 16 * D, kCoh);
      }
      SLIMT_STAMP(56);
      cluster_sync((unsigned)(2 * t + 1) * (unsigned)cl_n);
      SLIMT_STAMP(57);
      // (2) the cluster's rows in (rows of absent members: zeros, their results are never read)
      for (int i = tid; i < CR * (D / 16); i += 1024) {
        const int row = i / (D / 16), ch = i % (D / 16);
        v4i v = {0, 0, 0, 0};
        if (row < 16 * cl_n) v = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(ract, row * D + ch * 16, cl_first * 16 * D, kCoh));
        *reinterpret_cast<v4i *>(CA + row * LDA + ch * 16) = v;
      }
      // (3) my share of the column tiles against all of them. The weights are the MFMA's A operand: an accumulator lane
      // holds FOUR consecutive columns (16 tile + 4 lg ..) of ONE sentence (16 st + lr), so the running maximum is one
      // (value, column) pair per sentence tile; a lane's columns only grow, strict > keeps its first maximum.
      const int tpm = (outw.n_tiles + cl_n - 1) / cl_n;
      const int tile0 = cl_m * tpm, tile1 = tile0 + tpm < outw.n_tiles ? tile0 + tpm : outw.n_tiles;
      const rsrc_t rw = make_rsrc(outw.Wp, (unsigned)outw.n_tiles * KSD * 1024u);
      const rsrc_t rc = make_rsrc(outw.colsum, (unsigned)outw.n_tiles * 64u), rp = make_rsrc(outw.pb, (unsigned)outw.n_tiles * 64u);
      struct WTile {
        v4i f[KSD];
        v4i cs;
        f4 pb;
      };
      constexpr int NBC = 3;
      WTile wt[NBC];
      auto loadw = [&](WTile &x, int tl) {  // (past the last tile of the layer: zeros)
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) x.f[ks] = load_frag(rw, lane * 16, (tl * KSD + ks) * 1024);
        x.cs = load_frag(rc, lg * 16, tl * 64);
        x.pb = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rp, lg * 16, tl * 64, 0));
      };
#pragma unroll
      for (int k = 0; k < NBC; ++k) {
        loadw(wt[k], tile0 + wave + NW * k);
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();  // CA is complete
      SLIMT_STAMP(58);
      float cbv[CL];
      int cbi[CL];
#pragma unroll
      for (int st = 0; st < CL; ++st) {
        cbv[st] = -3.402823466e+38f;
        cbi[st] = 0x7fffffff;
      }
      const int mine = tile1 > tile0 + wave ? (tile1 - tile0 - wave + NW - 1) / NW : 0;
      const int rounds = (mine + NBC - 1) / NBC;
      for (int c = 0; c < rounds * NBC; c += NBC) {
#pragma unroll
        for (int k = 0; k < NBC; ++k) {
          const int tl = tile0 + wave + NW * (c + k);
          const bool in_slice = tl < tile1;
          const WTile &x = wt[k];
          const int col0 = tl * 16 + lg * 4;
          // The phase is bound by the issue slots of this epilogue (500 logits per lane and step whatever the split), so
          // per logit: the shift 127 colsum STARTS the accumulator (integer arithmetic: the same sum as adding it after),
          // conversion, half a packed multiply, half a packed add, compare, two selects. Columns outside my share or past
          // the layer's last one get a NaN bias instead of a predicate: their logit is NaN and never "greater".
          const v4i sh = {__mul24(127, x.cs[0]), __mul24(127, x.cs[1]), __mul24(127, x.cs[2]), __mul24(127, x.cs[3])};
          const int left = in_slice ? outw.N - col0 : 0;  // columns of mine in this lane's four
          const float qnan = __int_as_float(0x7fc00000);
          const f2 pb01 = {left > 0 ? x.pb.x : qnan, left > 1 ? x.pb.y : qnan}, pb23 = {left > 2 ? x.pb.z : qnan, left > 3 ? x.pb.w : qnan};
          const f2 uu = {a.out.u, a.out.u};
          // the CL sentence tiles' accumulator chains run INTERLEAVED (k-step outside, sentence tile inside): one tile after
          // the other, a wave sat out an LDS round trip + four dependent MFMAs per sentence tile (a member's quarter took
          // 27-41 us for 2 MB of weights and 500 logits per lane)
          v4i accs[CL];
#pragma unroll
          for (int st = 0; st < CL; ++st) accs[st] = sh;
#pragma unroll
          for (int ks = 0; ks < KSD; ++ks) {
            v4i af[CL];
#pragma unroll
            for (int st = 0; st < CL; ++st) af[st] = *reinterpret_cast<const v4i *>(CA + (16 * st + lr) * LDA + ks * 64 + lg * 16);
#pragma unroll
            for (int st = 0; st < CL; ++st) accs[st] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x.f[ks], af[st], accs[st], 0, 0, 0);
          }
#pragma unroll
          for (int st = 0; st < CL; ++st) {
            const v4i acc = accs[st];
            // y = float(acc + 127 colsum) * u + pb (Intgemm.inl.cc:146-153): dequant4's operations
            f2 lo = {(float)acc[0], (float)acc[1]}, hi = {(float)acc[2], (float)acc[3]};
            lo = lo * uu;
            hi = hi * uu;
            lo = lo + pb01;
            hi = hi + pb23;
            const float vv[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool better = vv[i] > cbv[st];
              cbv[st] = better ? vv[i] : cbv[st];
              cbi[st] = better ? col0 + i : cbi[st];
            }
          }
          loadw(wt[k], tl + NW * NBC);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      SLIMT_STAMP(44);
      // this wave's best per sentence: over the four lane groups (larger value, then smaller column)
      auto lane_step = [&](float &v, int &ix, int m) {
        const float ov = __shfl_xor(v, m, 64);
        const int o = __shfl_xor(ix, m, 64);
        const bool take = ov > v || (ov == v && o < ix);
        v = take ? ov : v;
        ix = take ? o : ix;
      };
#pragma unroll
      for (int st = 0; st < CL; ++st) {
        lane_step(cbv[st], cbi[st], 16);
        lane_step(cbv[st], cbi[st], 32);
        if (lg == 0) {
          cred_v[wave * CR + 16 * st + lr] = cbv[st];
          cred_i[wave * CR + 16 * st + lr] = cbi[st];
        }
      }
      SLIMT_STAMP(45);
      lds_barrier();
      // (4) my share's best per sentence, over the 16 waves, out to the sentence's owner
      for (int srow = wave; srow < CR; srow += NW) {
        float v = lane < NW ? cred_v[lane * CR + srow] : -3.402823466e+38f;
        int ix = lane < NW ? cred_i[lane * CR + srow] : 0x7fffffff;
        row16_argmax(v, ix);
        if (lane == 0) {
          const v2i rec = {__float_as_int(v), ix};
          __builtin_amdgcn_raw_buffer_store_b64(rec, rpart, srow * 8, tile * (CR + 1) * 8, kCoh);
        }
      }
      if (tid == 0) {  // ... and whether every sentence of mine has ended (as of the step before)
        const v2i rec = {flags[0] >= valid_rows ? 1 : 0, 0};
        __builtin_amdgcn_raw_buffer_store_b64(rec, rpart, CR * 8, tile * (CR + 1) * 8, kCoh);
      }
      SLIMT_STAMP(59);
      cluster_sync((unsigned)(2 * t + 2) * (unsigned)cl_n);
      SLIMT_STAMP(41);
      // (5) the owner: first maximum over the members' candidates (member = lane; columns ascend with it)
      if (row_wave) {
        float v = -3.402823466e+38f;
        int ix = 0x7fffffff;
        int done = 1;
        if (lane < cl_n) {
          const int member = (cl_first + lane) * (CR + 1) * 8;  // (lane-dependent: in the vector offset)
          const v2i rec = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rpart, member + (16 * cl_m + wave) * 8, 0, kCoh));
          const int rv = rec.x;  // (a copy first: bit casts of vector elements read element 0 with this compiler)
          v = __int_as_float(rv);
          ix = rec.y;
          const v2i drec = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rpart, member + CR * 8, 0, kCoh));
          done = drec.x;
        }
        row16_argmax(v, ix);
        cl_ix[0] = __builtin_amdgcn_readfirstlane(ix);
        const bool all = __builtin_amdgcn_ballot_w64(done != 0) == ~0ull;
        if (wave == 0 && lane == 0) flags[2] = all ? 1 : 0;
      }
    } else {
    Frags fl[NB_OUT];  // requested before the barrier that ends the last LayerNorm
    stream_prologue<KSD, NB_OUT, 0, (KSD >= 4)>(outw, wave, lane, fl);
    lds_barrier();
    SLIMT_STAMP(43);
    float bv[RT][4];
    int bi[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bv[rt][r] = -3.402823466e+38f;
        bi[rt][r] = 0x7fffffff;
      }
    stream_gemm_from<KSD, NB_OUT, 0, (KSD >= 4), RT>(
        A1, LDA, outw, wave, lane, fl, [&](int tile, int rt, const v4i &acc, int co, float pb) {
          const int col = tile * 16 + lr;
          const bool in_range = col < outw.N;  // no branch: the streaming loop stays one block
          const Dequant4 l4v = dequant4(acc, co, a.out.u, pb);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = l4v.v[r];
#ifdef SLIMT_EXP_LOGITS_MAXONLY  // timing only (wrong tokens): the epilogue without its index bookkeeping -- what a two-stage arg-max could save
            bv[rt][r] = fmaxf(bv[rt][r], v);
            bi[rt][r] = col;
            (void)in_range;
            continue;
#endif
            // a lane's columns only grow, so strict > keeps its first maximum
            const bool better = in_range && v > bv[rt][r];
            bv[rt][r] = better ? v : bv[rt][r];
            bi[rt][r] = better ? col : bi[rt][r];
          }
        });
    SLIMT_STAMP(44);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        row16_argmax(bv[rt][r], bi[rt][r]);
        if (lr == 0) {
          red_v[wave * R + 16 * rt + lg * 4 + r] = bv[rt][r];
          red_i[wave * R + 16 * rt + lg * 4 + r] = bi[rt][r];
        }
      }
    SLIMT_STAMP(45);
    lds_barrier();
    SLIMT_STAMP(41);
    }
    // wave w finishes sentences w (+ 16): reduce over the 16 waves' candidates
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) {
      if (!row_wave) break;
      const int row = 16 * rr + wave;
      uint32_t tok = 0;
      {
        int ix;
        if constexpr (CL > 1) {
          ix = cl_ix[rr];
        } else {
          float v = lane < NW ? red_v[lane * R + row] : -3.402823466e+38f;
          ix = lane < NW ? red_i[lane * R + row] : 0x7fffffff;
          row16_argmax(v, ix);
          ix = __builtin_amdgcn_readfirstlane(ix);
        }
        // no column beat the start value (every logit NaN or -inf): class 0, where the reference's scan
        // starts and stays (Transformer.cc:287-298) -- and never an index past the shortlist
        ix = (ix == 0x7fffffff || nan0) ? 0 : ix;
        const uint32_t *sl = n_sub ? a.sub[sj].shortlist : a.shortlist;  // (one output layer per tile: the tile's sub-batch's list)
        if (live[rr]) tok = sl ? sl[ix] : (uint32_t)ix;
      }
      if (live[rr] && !finished[rr]) {  // record(), Model.cc:127-137
        const int Tr = SLIMT_SUB_TMAX(rr);
        if (lane == 0 && (int)n_out[rr] < Tr)
          (n_sub ? a.sub[SLIMT_SW(rr)].out_ids : a.out_ids)[(size_t)(bq[rr] - SLIMT_SUB_FIRST(rr)) * Tr + n_out[rr]] = tok;
        n_out[rr] += 1;
        // (merged launches: a sentence also ends at its own sub-batch's step limit -- a tile may go on for the others)
        if (tok == a.eos || (n_sub && (int)n_out[rr] >= a.sub[SLIMT_SW(rr)].max_steps)) {
          finished[rr] = true;
          if (lane == 0) atomicAdd(&flags[0], 1);
        }
      }
      if (t + 1 < max_steps) {
        // next target embedding (Transformer.cc:146-160): position is always 0
#pragma unroll
        for (int i = 0; i < KSD; ++i) {
          float v = 0.0f;
          if (live[rr]) {
            const float e = (float)a.emb.wemb[(size_t)embed_row(a.emb, tok) * D + lane + 64 * i] * a.emb.inv_mult;
            const float sc = e * a.emb.sqrt_d;
            v = sc + a.emb.pos[lane + 64 * i];
          }
          xs[row * LDF + lane + 64 * i] = v;
        }
      }
    }
    // no barrier: the next step starts with this wave quantising its own rows
    SLIMT_STAMP(42);
    if (a.stamps && m0 == 0 && tid == 0 && t == a.stamp_step) a.stamps[61] = clock64();
  }
#pragma unroll
  for (int rr = 0; rr < RT; ++rr)
    if (live[rr] && lane == 0) (n_sub ? a.sub[sw[rr]].out_len : a.out_len)[bq[rr] - SLIMT_SUB_FIRST(rr)] = n_out[rr];
  if (n_sub || (a.align && a.align_out)) {  // (the uniform test first: without it the f32-cache 32-sentence kernel spilled 75 registers for 37)
    // staged alignment rows -> their destination (kernels.h, align_out). This wave wrote the rows it
    // reads (rows 0 .. n_out - 1, columns 0 .. len - 1); everything else is zero.
#pragma unroll 1
    for (int rr = 0; rr < RT; ++rr) {
      if (!live[rr]) continue;
      const float *al_src = n_sub ? a.sub[sw[rr]].align : a.align;
      float *al_dst = n_sub ? a.sub[sw[rr]].align_out : a.align_out;
      if (!al_src || !al_dst) continue;
      const int Tr = SLIMT_SUB_TMAX(rr), Sr = SLIMT_SUB_S(rr);
      const size_t base = (size_t)(bq[rr] - SLIMT_SUB_FIRST(rr)) * Tr * Sr;
      const int rows_set = (int)n_out[rr] < Tr ? (int)n_out[rr] : Tr;
      const int n = Tr * Sr, ln = len[rr];
      if (((base | (size_t)Sr) & 3) == 0 && (reinterpret_cast<size_t>(al_dst) & 15) == 0 &&
          (reinterpret_cast<size_t>(al_src) & 15) == 0) {  // 16-byte pieces: four columns of one row
        const f4 *src = reinterpret_cast<const f4 *>(al_src + base);
        f4 *dst = reinterpret_cast<f4 *>(al_dst + base);
        for (int i = lane; i < n / 4; i += 64) {
          const int row = (4 * i) / Sr, col = (4 * i) % Sr;
          f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
          if (row < rows_set && col < ln) {
            v = src[i];
            if (col + 1 >= ln) v.y = 0.0f;
            if (col + 2 >= ln) v.z = 0.0f;
            if (col + 3 >= ln) v.w = 0.0f;
          }
          dst[i] = v;
        }
      } else {
        for (int i = lane; i < n; i += 64) {
          const int row = i / Sr, col = i % Sr;
          al_dst[base + i] = (row < rows_set && col < ln) ? al_src[base + i] : 0.0f;
        }
      }
    }
  }
  if (tid == 0) occ_trace_event(a.trace, 1, 1);
}

// sentences per workgroup the fused decoder uses: 32 (two row tiles) on request (decode mode 3, or the
// engine's choice for output layers of more than 16k columns) and only for the D = 256 shapes with
// short sentences; 8 or 4 (decode_fused_kernel, SPW) on request where the packed K/V cache variants exist
// (the engine asks for them when a launch at 16 would leave most CUs idle); else 16.
int fused_decode_rows(int D, int F, int H, int Ld, int S, int B, int forced, bool kv24) {
  (void)B;
  const bool ok32 = D == 256 && F == 1536 && D / H == 32 && Ld <= 4 && S <= 32;
  if (forced == 32 && ok32) return 32;
  const bool small = kv24 && Ld <= 4 &&
                     ((D == 256 && F == 1536 && D / H == 32 && S <= 128) || (D == 512 && F == 2048 && D / H == 64 && S <= 32));
  if ((forced == 8 || forced == 4) && small) return forced;
  return 16;
}

// workgroups launched for B sentences: with tickets 2 x the tiles (up to 16 sentences each) / 4 x (32),
// i.e. one candidate per shader engine of every XCD for a batch of 256
int fused_decode_grid(int B, bool tickets, int rows) {
  const int tiles = (B + rows - 1) / rows;
  return tickets ? tiles * (rows <= 16 ? 2 : 4) : tiles;
}

// ln_in_lds (out, nullable): whether the LayerNorm constants of all layers (LN_LDS in the kernel: the
// D = 256, 16-row, non-MID variants) still fit the 160 KiB; the returned size includes them then.
// tight: the kernels with the 16-bit cache form inlined (KVI = 16) keep its column terms [Ld][K, V][D] behind everything else.
size_t fused_decode_lds_bytes(int D, int F, int Ld, int rows, bool kv24 = false, int mid = 0,
                              bool *ln_in_lds = nullptr, bool tight = false) {
  // D * rows > 256 * 16: two f32 row buffers, SSRU cells in global memory (see the kernel)
  const size_t R = (size_t)rows;
  const bool lean = (size_t)D * R > 256 * 16;
  const size_t cells = (lean || mid == 2) ? 0 : (size_t)Ld * R * D * 4;
  const size_t f32rows = (lean ? 2 : 3) * R * (D + 4) * 4 + cells;
  const size_t base = f32rows + 2 * R * (size_t)(D + 32) + R * (size_t)(F + 32) + 2 * NW * R * 4 + 64 +
                      NW * (mid == 2 ? 1024 : mid == 1 ? 512 : 256) * 4 + (kv24 ? (size_t)Ld * (D == 512 ? 4 : 2) * D * 4 : 0) +
                      (tight ? (size_t)Ld * (rows > 16 ? 1 : 2) * D * 4 : 0);  // (32 sentences: the K centres only)
  const size_t ln = (D == 256 && rows == 16 && !mid) ? (size_t)Ld * 6 * D * 4 : 0;
  const bool fits = ln > 0 && base + ln <= 160 * 1024;
  if (ln_in_lds) *ln_in_lds = fits;
  return fits ? base + ln : base;
}

// the tight (16-bit) cache form: the short-sentence tilings of 16 / 8 / 4 sentences, D = 256 / F = 1536 and D = 512 / F = 2048
// (decode_fused_kernel<..., KVI = 16>)
bool fused_decode_tight_supported(int D, int F, int H, int Ld) {
  if (Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  return ((D == 256 && F == 1536 && D / H == 32) || (D == 512 && F == 2048 && D / H == 64)) &&
         fused_decode_lds_bytes(D, F, Ld, 16, true, 0, nullptr, true) <= 160 * 1024;
}

// ... in the 32-sentence tiling (D = 256, S <= 32; decode_fused_kernel<..., RT = 2, ..., KVI = 16>)
bool fused_decode_tight_rows32_supported(int D, int F, int H, int Ld) {
  if (Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  return D == 256 && F == 1536 && D / H == 32 && fused_decode_lds_bytes(D, F, Ld, 32, true, 0, nullptr, true) <= 160 * 1024;
}

// ... and for sentences of 33..128 tokens (D = 256; decode_fused_kernel<..., MID = 1 / 2, ..., KVI = 16>)
bool fused_decode_tight_mid_supported(int D, int F, int H, int Ld, int mid) {  // mid: 1 = 33..64 tokens, 2 = 65..128
  if (Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  return D == 256 && F == 1536 && D / H == 32 && fused_decode_lds_bytes(D, F, Ld, 16, true, mid, nullptr, true) <= 160 * 1024;
}

// sentences of 33..64 tokens over the packed cache (decode_fused_kernel<..., MID>)
bool fused_decode_mid_supported(int D, int F, int H, int Ld) {
  if (Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  return D == 256 && F == 1536 && D / H == 32 && fused_decode_lds_bytes(D, F, Ld, 16, true, 1) <= 160 * 1024;
}

// sentences of 65..128 tokens over the packed cache (decode_fused_kernel<..., MID = 2>)
bool fused_decode_long24_supported(int D, int F, int H, int Ld) {
  if (Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  return D == 256 && F == 1536 && D / H == 32 && fused_decode_lds_bytes(D, F, Ld, 16, true, 2) <= 160 * 1024;
}

bool fused_decode_supported(int D, int F, int H, int Ld) {
  if (Ld < 1 || Ld > 4 || H <= 0 || D % H) return false;
  const int dh = D / H;
  const bool shape = (D == 64 && F == 128 && dh == 16) || (D == 128 && F == 256 && dh == 16) ||
                     (D == 256 && F == 1536 && dh == 32) || (D == 512 && F == 2048 && dh == 64);
  return shape && fused_decode_lds_bytes(D, F, Ld, 16) <= 160 * 1024;
}

// the long-sentence instantiation exists for d_head 32 only (attention_row_long), the
// non-temporal K/V variant for d_head 32 and 64 (the buffer-load paths of attention_row)
template <bool MG, int KSD, int KSF, int DH>
static auto decode_fused_pick(bool long_sentences, bool nt) -> void (*)(FusedDecodeArgs) {
  if constexpr (DH == 32) {
    if (long_sentences) return nt ? decode_fused_kernel<MG, KSD, KSF, DH, true, true> : decode_fused_kernel<MG, KSD, KSF, DH, true, false>;
  }
  if constexpr (DH >= 32) {
    if (nt) return decode_fused_kernel<MG, KSD, KSF, DH, false, true>;
  }
  (void)long_sentences;
  (void)nt;
  return decode_fused_kernel<MG, KSD, KSF, DH, false, false>;
}

template <bool MG>
static hipError_t launch_decode_fused_t(const FusedDecodeArgs &a_in, int D, int F, int H, hipStream_t st) {
  FusedDecodeArgs a = a_in;
  if (!fused_decode_supported(D, F, H, a.Ld)) return hipErrorInvalidValue;
  const bool kv24 = a.kv24;
  const int rows = fused_decode_rows(D, F, H, a.Ld, a.S, a.B, a.rows_per_wg, kv24);
  const dim3 grid(a.home_mask ? (int)a.xgrid : fused_decode_grid(a.B, a.ticket != nullptr, rows));
  const int mid = (kv24 && D == 256 && a.S > 32) ? (a.S > 64 ? 2 : 1) : 0;  // 33..64 / 65..128-token sentences
  if (kv24 && !(((D == 256 && D / H == 32) || (D == 512 && D / H == 64)) && a.S <= (D == 256 ? 128 : 32))) return hipErrorInvalidValue;
  if (a.home_mask && rows != 16) return hipErrorInvalidValue;  // the XCD-affine claim counts 16-sentence tiles
  // a batch with sentence-layers in the tight form: only the kernels with its reader (engine.cpp, kv_tight_wanted)
  if (a.kv_tight && !(kv24 && a.kv_fmt && a.cluster <= 1 &&
                      (rows == 32 ? a.S <= 32 && fused_decode_tight_rows32_supported(D, F, H, a.Ld)
                       : a.S <= 32 ? fused_decode_tight_supported(D, F, H, a.Ld)
                                   : a.S <= 128 && fused_decode_tight_mid_supported(D, F, H, a.Ld, mid))))
    return hipErrorInvalidValue;
  auto go = [&](void (*k)(FusedDecodeArgs), size_t lds) -> hipError_t {
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, st, a);
    return hipGetLastError();
  };
  // the variants over the packed cache: <KSD, KSF, DH, MID> x non-temporal K/V loads x sentences per workgroup
#define SLIMT_KV24_PICK(KSD_, KSF_, DH_, MID_)                                                                  \
  (rows == 4   ? (a.kv_nt ? decode_fused_kernel<MG, KSD_, KSF_, DH_, false, true, 1, true, MID_, 4>                 \
                          : decode_fused_kernel<MG, KSD_, KSF_, DH_, false, false, 1, true, MID_, 4>)               \
   : rows == 8 ? (a.kv_nt ? decode_fused_kernel<MG, KSD_, KSF_, DH_, false, true, 1, true, MID_, 8>                 \
                          : decode_fused_kernel<MG, KSD_, KSF_, DH_, false, false, 1, true, MID_, 8>)               \
               : (a.kv_nt ? decode_fused_kernel<MG, KSD_, KSF_, DH_, false, true, 1, true, MID_>                    \
                          : decode_fused_kernel<MG, KSD_, KSF_, DH_, false, false, 1, true, MID_>))
  // every cache of this launch in the 24-bit form (a.kv_fmt == nullptr): the 16-sentence tilings have an instantiation
  // with that form inlined (KVI = 24); the 8- / 4-sentence ones reach it through the fallback call
#define SLIMT_KV24_ONLY(KSD_, KSF_, DH_, MID_)                                                        \
  (a.kv_nt ? decode_fused_kernel<MG, KSD_, KSF_, DH_, false, true, 1, true, MID_, 16, 1, 24>          \
           : decode_fused_kernel<MG, KSD_, KSF_, DH_, false, false, 1, true, MID_, 16, 1, 24>)
  const bool only24 = kv24 && !a.kv_fmt && rows == 16 && a.cluster <= 1;
  if constexpr (MG) {  // (the merged twins: 16-row tilings, sentences of up to 64 tokens, no clusters)
    if (rows > 16 || mid == 2 || a.cluster > 1) return hipErrorInvalidValue;
  }
  if (mid) {
    if (rows > 16 || F != 1536) return hipErrorInvalidValue;
    const size_t ldsm = fused_decode_lds_bytes(D, F, a.Ld, 16, true, mid, nullptr, a.kv_tight);
    if (ldsm > 160 * 1024) return hipErrorInvalidValue;
    if (a.kv_tight) {
#define SLIMT_KV16_PICK(MID_, SPW_)                                                                \
  (a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 1, true, MID_, SPW_, 1, 16>               \
           : decode_fused_kernel<MG, 4, 24, 32, false, false, 1, true, MID_, SPW_, 1, 16>)
      if (mid == 1) return go(rows == 4 ? SLIMT_KV16_PICK(1, 4) : rows == 8 ? SLIMT_KV16_PICK(1, 8) : SLIMT_KV16_PICK(1, 16), ldsm);
      if constexpr (!MG) return go(rows == 4 ? SLIMT_KV16_PICK(2, 4) : rows == 8 ? SLIMT_KV16_PICK(2, 8) : SLIMT_KV16_PICK(2, 16), ldsm);
#undef SLIMT_KV16_PICK
    }
    if constexpr (!MG) {
      if (only24 && mid == 2) return go(SLIMT_KV24_ONLY(4, 24, 32, 2), ldsm);
      if (mid == 2) return go(SLIMT_KV24_PICK(4, 24, 32, 2), ldsm);
    }
    if (only24) return go(SLIMT_KV24_ONLY(4, 24, 32, 1), ldsm);
    return go(SLIMT_KV24_PICK(4, 24, 32, 1), ldsm);
  }
  const size_t lds = fused_decode_lds_bytes(D, F, a.Ld, rows <= 16 ? 16 : rows, kv24, 0, &a.ln_in_lds, a.kv_tight);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (kv24 && D == 512) {
    if (F != 2048) return hipErrorInvalidValue;
    if (a.kv_tight) {
#define SLIMT_KV16_PICK(SPW_)                                                                      \
  (a.kv_nt ? decode_fused_kernel<MG, 8, 32, 64, false, true, 1, true, 0, SPW_, 1, 16>                  \
           : decode_fused_kernel<MG, 8, 32, 64, false, false, 1, true, 0, SPW_, 1, 16>)
      return go(rows == 4 ? SLIMT_KV16_PICK(4) : rows == 8 ? SLIMT_KV16_PICK(8) : SLIMT_KV16_PICK(16), lds);
#undef SLIMT_KV16_PICK
    }
    if (only24) return go(SLIMT_KV24_ONLY(8, 32, 64, 0), lds);
    return go(SLIMT_KV24_PICK(8, 32, 64, 0), lds);
  }
  if constexpr (!MG) {
  if (a.cluster > 1) {  // cluster logits: the 16-sentence tiling of the D = 256 packed-cache shape
    if (!(kv24 && D == 256 && F == 1536 && rows == 16 && a.cluster == 4 && a.cl_act && a.cl_part && a.cl_sync)) return hipErrorInvalidValue;
    return go(a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 1, true, 0, 16, 4>
                      : decode_fused_kernel<MG, 4, 24, 32, false, false, 1, true, 0, 16, 4>, lds);
  }
  }
  if (only24) return go(SLIMT_KV24_ONLY(4, 24, 32, 0), lds);
  if (kv24 && rows <= 16 && a.kv_tight) {  // sentences may be in the tight 16-bit form: the kernels with it (and the 20-bit one) inlined
#define SLIMT_KV16_PICK(SPW_)                                                                      \
  (a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 1, true, 0, SPW_, 1, 16>                  \
           : decode_fused_kernel<MG, 4, 24, 32, false, false, 1, true, 0, SPW_, 1, 16>)
    return go(rows == 4 ? SLIMT_KV16_PICK(4) : rows == 8 ? SLIMT_KV16_PICK(8) : SLIMT_KV16_PICK(16), lds);
#undef SLIMT_KV16_PICK
  }
  if (kv24 && rows <= 16) return go(SLIMT_KV24_PICK(4, 24, 32, 0), lds);
#undef SLIMT_KV24_PICK
#undef SLIMT_KV24_ONLY
  if constexpr (!MG) {
  if (rows == 32) {
    if (kv24 && a.kv_tight)
      return go(a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 2, true, 0, 16, 1, 16>
                        : decode_fused_kernel<MG, 4, 24, 32, false, false, 2, true, 0, 16, 1, 16>, lds);
    if (kv24 && !a.kv_fmt)  // every cache in the 24-bit form: that form inlined (KVI = 24), as for the 16-sentence tilings
      return go(a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 2, true, 0, 16, 1, 24>
                        : decode_fused_kernel<MG, 4, 24, 32, false, false, 2, true, 0, 16, 1, 24>, lds);
    auto k = kv24 ? (a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 2, true> : decode_fused_kernel<MG, 4, 24, 32, false, false, 2, true>)
                  : (a.kv_nt ? decode_fused_kernel<MG, 4, 24, 32, false, true, 2> : decode_fused_kernel<MG, 4, 24, 32, false, false, 2>);
    return go(k, lds);
  }
  }
#define SLIMT_FUSED_CASE(KSD_, KSF_, DH_)                                                   \
  if (D == 64 * KSD_ && F == 64 * KSF_ && D / H == DH_) {                                    \
    auto k = decode_fused_pick<MG, KSD_, KSF_, DH_>(a.S > 32, a.kv_nt);                           \
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(k), (int)lds); \
    if (e != hipSuccess) return e;                                                           \
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, st, a);                                     \
    return hipGetLastError();                                                                \
  }
  SLIMT_FUSED_CASE(1, 2, 16) SLIMT_FUSED_CASE(2, 4, 16) SLIMT_FUSED_CASE(4, 24, 32)
  SLIMT_FUSED_CASE(8, 32, 64)
#undef SLIMT_FUSED_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_decode_fused(const FusedDecodeArgs &a, int D, int F, int H, hipStream_t st) {
  return a.n_sub > 0 ? launch_decode_fused_t<true>(a, D, F, H, st) : launch_decode_fused_t<false>(a, D, F, H, st);
}

}  // namespace slimt_hip
