// Device-side numerics shared by every kernel (gfx950 only).
//
// The float contract ("portable order", stated in DESIGN.md; the test oracle
// restates it independently on the CPU):
//  * every float expression is evaluated literally (this TU is compiled with
//    -ffp-contract=off; fused multiply-adds are written as __builtin_fmaf);
//  * exp is the fixed fmaf polynomial below, never the ocml/libm one;
//  * a row sum is "lane = i % 64 adds its elements in ascending i, then a xor
//    butterfly with masks 1,2,4,8,16,32" -- the natural wave64 reduction;
//  * division and sqrt are the correctly rounded IEEE ones
//    (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).
// With that the GPU reproduces the CPU restatement of the reference bit for
// bit, and differs from the reference's own scalar libm path by float
// rounding only (<= ~1e-6 relative, tests/test_oracle.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace slimt_hip {

// a sentence length as the kernels use it: device-resident arrays cannot be checked by the host, and a length past the
// padded width S would size cache descriptors past the sentence's block
__device__ __forceinline__ int checked_length(uint32_t len, int S) { return len < (uint32_t)S ? (int)len : S; }

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- the narrow (20-bit) form of the packed K/V cache (kernels.h, FusedDecodeArgs::kv_fmt) ----------------------
// Eight accumulators x[0..7], each in [-2^19, 2^19): x = 16 hi + lo with hi = x >> 4 (a signed 16-bit integer) and
// lo = x & 15. The eight hi halves are one 16-byte quad (little endian, value c in the low / high half of dword c / 2),
// the eight lo nibbles one dword (value c at bits 4 c .. 4 c + 3). The decoder rebuilds x << 12 with ONE v_perm per
// value: bytes {hi's two, lo << 4, 0} (decode_fused.hip, unpack20).
struct Packed20 {
  v4i hi;
  int lo;
};
__device__ __forceinline__ Packed20 pack20(const v4i &a, const v4i &b) {
  const int x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  Packed20 o;
  int h[4];
  o.lo = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    h[c] = ((x[2 * c] >> 4) & 0xffff) | ((x[2 * c + 1] << 12) & (int)0xffff0000);
    o.lo |= ((x[2 * c] & 15) << (8 * c)) | ((x[2 * c + 1] & 15) << (8 * c + 4));
  }
  o.hi = v4i{h[0], h[1], h[2], h[3]};
  return o;
}
// The tight form (kernels.h, FusedDecodeArgs::kv_tight): eight SIGNED accumulators, each in [-2^15, 2^15), as one quad of
// int16 (value c in the low / high half of dword c / 2).
__device__ __forceinline__ v4i pack16(const v4i &a, const v4i &b) {
  return v4i{(a.x & 0xffff) | (a.y << 16), (a.z & 0xffff) | (a.w << 16), (b.x & 0xffff) | (b.y << 16), (b.z & 0xffff) | (b.w << 16)};
}
// does x fit the narrow form?
__device__ __forceinline__ bool fits20(int x) { return (unsigned)(x + (1 << 19)) < (1u << 20); }
constexpr int kKvNarrowMin = -(1 << 19), kKvNarrowMax = (1 << 19) - 1;

__device__ __forceinline__ float exp_p(float x) {
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = __builtin_fmaf(p, r2, r) + 1.0f;
  int ni = (int)n;
  return y * __int_as_float((ni + 127) << 23);
}

// exp_p without its two early exits: the polynomial runs for every input and the special cases are selected
// afterwards -- the same bits for every float (below -86 the discarded product may be anything; a NaN takes the
// polynomial path in both forms), no exec-mask branches around a dozen instructions.
__device__ __forceinline__ float exp_p_select(float x) {
  const float xc = x > 88.0f ? 88.0f : x;
  float n = __builtin_rintf(xc * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693359375f, xc);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = __builtin_fmaf(p, r2, r) + 1.0f;
  const int ni = (int)n;
  const float v = y * __int_as_float((ni + 127) << 23);
  return x < -86.0f ? 0.0f : v;
}

// TensorOps.cc:33-36
__device__ __forceinline__ float sigmoid_p(float x) {
  if (x > 0) return 1.0f / (1.0f + exp_p(-x));
  float e = exp_p(x);
  return e / (1.0f + e);
}
// The same without divergent branches: both arms are exp of the non-positive argument over 1 + that, only the numerator
// differs -- one exponential and one division per value whatever the signs in the wave (same bits: same operations on
// the same operands; a NaN compares false and takes the second arm in both forms).
__device__ __forceinline__ float sigmoid_p_select(float x) {
  const bool pos = x > 0;
  const float t = exp_p_select(pos ? -x : x);
  return (pos ? 1.0f : t) / (1.0f + t);
}

// Butterfly partners. The canonical reduction is the xor butterfly with masks
// ascending (1, 2, 4, ..., 32): at step M every lane combines its value with
// that of lane (l ^ M). Only WHICH VALUE arrives matters for the bits, not the
// mechanism, so the cheap VALU data paths are used (no LDS crossbar trip):
//   M = 1, 2   DPP quad_perm;
//   M = 4, 8   DPP row_half_mirror / row_mirror: inside a butterfly the groups
//              of M lanes are already uniform, so lane (7 - l) / (15 - l) holds
//              the same bits as lane (l ^ 4) / (l ^ 8);
//   M = 16, 32 gfx950 v_permlane16_swap / v_permlane32_swap.
// butterfly_pair<M> is therefore only valid as step M of a butterfly whose
// lower steps have been applied (with a commutative op); wave_sum and friends
// are the only users.
typedef unsigned slimt_u2 __attribute__((ext_vector_type(2)));
// (a, b) = (own value, partner's value) in some order -- enough for a
// commutative op, and it saves the select after a permlane swap.
template <int M>
__device__ __forceinline__ void butterfly_pair(float v, float &a, float &b) {
  const int i = __float_as_int(v);
  a = v;
#ifdef SLIMT_REDUCE_BPERMUTE
  (void)i;
  b = __shfl_xor(v, M, 64);
  return;
#endif
  if constexpr (M == 1) {
    b = __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, false));  // [1,0,3,2]
  } else if constexpr (M == 2) {
    b = __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, false));  // [2,3,0,1]
  } else if constexpr (M == 4) {
    b = __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x141, 0xf, 0xf, false));  // row_half_mirror
  } else if constexpr (M == 8) {
    b = __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x140, 0xf, 0xf, false));  // row_mirror
  } else if constexpr (M == 16) {
    // even rows: (.x, .y) = (own, upper neighbour row); odd rows: (lower neighbour row, own)
    const slimt_u2 r = __builtin_amdgcn_permlane16_swap(i, i, false, false);
    a = __int_as_float(r.x);
    b = __int_as_float(r.y);
  } else {
    const slimt_u2 r = __builtin_amdgcn_permlane32_swap(i, i, false, false);
    a = __int_as_float(r.x);
    b = __int_as_float(r.y);
  }
}

template <int M>
__device__ __forceinline__ float bf_add(float v) {
  float a, b;
  butterfly_pair<M>(v, a, b);
  return a + b;
}
template <int M>
__device__ __forceinline__ float bf_max(float v) {
  float a, b;
  butterfly_pair<M>(v, a, b);
  return fmaxf(a, b);
}

// argmax butterfly over the 16 lanes of a DPP row (masks 1, 2, 4, 8): larger
// value wins, equal values -> smaller index (a total order, so both partners
// agree and the mirror trick above stays valid). All 16 lanes end equal.
template <int M>
__device__ __forceinline__ void argmax_step(float &v, int &ix) {
  float a, ov;
  butterfly_pair<M>(v, a, ov);
  float ia, oi;
  butterfly_pair<M>(__int_as_float(ix), ia, oi);
  const int o = __float_as_int(oi);
  const bool take = ov > v || (ov == v && o < ix);
  v = take ? ov : v;
  ix = take ? o : ix;
}
__device__ __forceinline__ void row16_argmax(float &v, int &ix) {
  argmax_step<1>(v, ix);
  argmax_step<2>(v, ix);
  argmax_step<4>(v, ix);
  argmax_step<8>(v, ix);
}

// xor butterfly over the 64 lanes, masks ascending; all lanes end equal.
__device__ __forceinline__ float wave_sum(float v) {
  return bf_add<32>(bf_add<16>(bf_add<8>(bf_add<4>(bf_add<2>(bf_add<1>(v))))));
}
__device__ __forceinline__ float wave_max(float v) {
  return bf_max<32>(bf_max<16>(bf_max<8>(bf_max<4>(bf_max<2>(bf_max<1>(v))))));
}
// the same butterflies restricted to each 32-lane half (masks 1..16)
__device__ __forceinline__ float half_sum(float v) {
  return bf_add<16>(bf_add<8>(bf_add<4>(bf_add<2>(bf_add<1>(v)))));
}
__device__ __forceinline__ float half_max(float v) {
  return bf_max<16>(bf_max<8>(bf_max<4>(bf_max<2>(bf_max<1>(v)))));
}

// intgemm PrepareA: round-to-nearest-even, clamp to [-127, 127]
// (Intgemm.inl.cc:29-34). max/min return the non-NaN operand,
// so NaN quantises to -127 -- what intgemm's cvtps_epi32 (INT_MIN) + saturating
// packs + max_epi8(-127) produce as well.
// Workgroup barrier for phases that hand their data over through LDS only. __syncthreads()
// is a full workgroup fence: the compiler puts s_waitcnt vmcnt(0) in front of s_barrier, which
// also waits for every global load in flight -- a weight prefetch issued before the barrier
// then costs its whole round trip AT the barrier. This one orders LDS traffic only
// (s_waitcnt lgkmcnt(0); s_barrier) and leaves global loads and stores in flight. Not for data
// that changes hands through global memory.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int quantize1(float x, float a_quant) {
  float v = __builtin_rintf(x * a_quant);
  v = __builtin_fminf(__builtin_fmaxf(v, -127.0f), 127.0f);
  return (int)v;
}
// quantize1 for a BYTE store (the low byte of the result is the int8): clamp(rint(t), -127, 127) == rint(clamp(t, -127, 127))
// (integer bounds, rint is monotonic), and adding 1.5 * 2^23 rounds to nearest even and leaves the integer's two's
// complement in the low mantissa bits -- multiply, v_med3 (a NaN product -> -127, like the max / min pair), add
// instead of multiply, v_rndne, v_med3, v_cvt_i32.
__device__ __forceinline__ int quantize1_byte(float x, float a_quant) {
  const float t = __builtin_amdgcn_fmed3f(x * a_quant, -127.0f, 127.0f);
  return __float_as_int(t + 12582912.0f);
}

// Correctly rounded quotients of SEVERAL numerators over ONE denominator (a LayerNorm row's sigma, a
// softmax row's sum). The compiler expands n / d (-fhip-fp32-correctly-rounded-divide-sqrt) into
//   v_div_scale x2, v_rcp, 2 fma (reciprocal refinement), mul, 3 fma (quotient refinement),
//   v_div_fmas, v_div_fixup
// and while neither operand needs scaling and nothing is special, scale / fmas / fixup are the
// identity: the quotient IS the fma chain below, whose first three operations depend on d alone.
// So the reciprocal is refined once per denominator and a quotient costs mul + 4 fma + its guard
// instead of twelve instructions (one of them a quarter-rate v_rcp) -- the same bits, not an
// approximation: tools/probes/div_probe.hip compares the two over 1.7e10 operand pairs per guard
// range (profiles/r03_div_probe.txt: 0 mismatches inside the ranges used here, mismatches outside).
// Outside the guard (any lane of the wave: the branch is uniform) the compiler's division runs.
struct SharedDiv {
  float d, r1;
  bool d_ok;
  __device__ __forceinline__ SharedDiv(float den, float d_lo, float d_hi) : d(den) {
    const float r0 = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r0, 1.0f);
    r1 = __builtin_fmaf(e, r0, r0);
    d_ok = den >= d_lo && den <= d_hi;
  }
  __device__ __forceinline__ float chain(float n) const {
    const float q0 = n * r1;
    const float m0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(m0, r1, q0);
    const float m1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(m1, r1, q1);
  }
  // n_lo <= |n| <= n_hi, or n == +0 exactly where ZERO (a softmax's masked keys). Bitwise, not
  // short-circuit: `&&` / `||` on per-lane conditions compile to nested exec-mask branches.
  template <bool ZERO>
  __device__ __forceinline__ unsigned inside(float n, float n_lo, float n_hi) const {
    const float an = __builtin_fabsf(n);
    unsigned in = (unsigned)(an >= n_lo);
    if (ZERO) return in | (unsigned)(__float_as_uint(n) == 0u);  // (a softmax term is exp(x - max) <= 1: no upper check;
    return in & (unsigned)(an <= n_hi);                           //  a NaN fails `>=` and takes the compiler's division)
  }
  // N quotients, one guard for all of them
  template <int N, bool ZERO>
  __device__ __forceinline__ void quot(float (&n)[N], float n_lo, float n_hi) const {
    unsigned in = (unsigned)d_ok;
#pragma unroll
    for (int i = 0; i < N; ++i) in &= inside<ZERO>(n[i], n_lo, n_hi);
    if (__builtin_amdgcn_ballot_w64(in == 0u) != 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) n[i] = n[i] / d;
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) n[i] = chain(n[i]);
    }
  }
};
// guard ranges verified by the probe: LayerNorm (v - mean) / sigma, sigma = sqrt(var + 1e-6) >= 2^-10;
// softmax e / sum, e in [0, 1], sum >= 1 (the maximum's own term)
#define SLIMT_DIV_LN_D 0x1p-12f, 0x1p20f
#define SLIMT_DIV_LN_N 0x1p-100f, 0x1p40f
#define SLIMT_DIV_SM_D 0x1p-1f, 0x1p12f
#define SLIMT_DIV_SM_N 0x1p-100f, 0x1p1f

__device__ __forceinline__ int pack4(int a, int b, int c, int d) {
  return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | ((d & 0xff) << 24);
}

// Canonical LayerNorm of one row by one wave (TensorOps.cc:542-580).
// x, y may alias. All 64 lanes must call.
__device__ __forceinline__ void wave_layer_norm_row(const float *x, const float *scale,
                                                    const float *bias, float eps, int D,
                                                    float *y, int lane) {
  float s = 0.0f;
  for (int i = lane; i < D; i += 64) s += x[i];
  s = wave_sum(s);
  float mean = s / (float)D;
  float q = 0.0f;
  for (int i = lane; i < D; i += 64) {
    float v = x[i] - mean;
    q += v * v;
  }
  q = wave_sum(q);
  float sigma = __builtin_sqrtf(q / (float)D + eps);
  for (int i = lane; i < D; i += 64) {
    float t = (x[i] - mean) / sigma;
    float sc = scale[i] * t;
    y[i] = sc + bias[i];
  }
}

// one event of the diagnostic occupancy trace (kernels.h, OccTrace); call from ONE thread
__device__ __forceinline__ void occ_trace_event(const OccTrace &t, unsigned kernel, unsigned end) {
  if (!t.buf) return;
  const unsigned long long slot = atomicAdd(t.buf, 1ull);
  if (slot >= t.capacity) return;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: CU / SH / SE
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  unsigned long long *r = t.buf + 1 + 3 * slot;
  r[0] = (unsigned long long)kernel | ((unsigned long long)end << 8) | ((unsigned long long)blockIdx.x << 16);
  r[1] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  r[2] = wall_clock64();
}

__device__ __forceinline__ int sum_bytes(int w) {
  return (int)(int8_t)(w & 0xff) + (int)(int8_t)((w >> 8) & 0xff) +
         (int)(int8_t)((w >> 16) & 0xff) + (int)(int8_t)((w >> 24) & 0xff);
}

// One 16-column tile of a packing job by `nthreads` (>= 256, a multiple of 64)
// threads of a workgroup: PrepareB's re-layout + PrepareBias' column sums
// (Intgemm.inl.cc:48-69,127-136). No barrier inside.
__device__ __forceinline__ void pack_weight_tile(const PackArgs &a, int ntile, int tid, int nthreads) {
  v4i *Wp = reinterpret_cast<v4i *>(a.Wp);
  const int N = a.n_dev ? (int)*a.n_dev : a.N;
  if (ntile * 16 >= N) return;  // device-side N: tiles past it are not part of the matrix
  const int KS = a.K / 64;
  const int chunks = a.K / 16;  // 16-byte chunks per row
  for (int c = tid; c < 16 * chunks; c += nthreads) {
    const int r = c / chunks, ch = c % chunks;
    const int n = ntile * 16 + r;
    v4i v = {0, 0, 0, 0};
    if (n < N) {
      size_t src = a.idx ? (size_t)a.idx[n] : (size_t)n;
      if (a.n_src > 0 && src >= (size_t)a.n_src) src = (size_t)a.n_src - 1;
      v = *reinterpret_cast<const v4i *>(a.W + src * a.K + (size_t)ch * 16);
    }
    const int ks = ch >> 2, kg = ch & 3;
    Wp[((size_t)ntile * KS + ks) * 64 + kg * 16 + r] = v;
  }
  if (tid < 256) {  // column sums: 16 threads per row
    const int r = tid >> 4, sub = tid & 15;
    const int n = ntile * 16 + r;
    int s = 0;
    size_t src = 0;
    if (n < N) {
      src = a.idx ? (size_t)a.idx[n] : (size_t)n;
      if (a.n_src > 0 && src >= (size_t)a.n_src) src = (size_t)a.n_src - 1;
      for (int ch = sub; ch < chunks; ch += 16) {
        v4i v = *reinterpret_cast<const v4i *>(a.W + src * a.K + (size_t)ch * 16);
        s += sum_bytes(v.x) + sum_bytes(v.y) + sum_bytes(v.z) + sum_bytes(v.w);
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (sub == 0) {
      float pbv = 0.0f;
      if (n < N) {
        float v = (float)s * a.mult;  // PrepareBias callback: cvt, mul, add
        pbv = v + (a.bias ? a.bias[src] : 0.0f);
      } else {
        s = 0;
      }
      a.colsum[n] = s;
      a.pb[n] = pbv;
      // the pair form (kernels.h, PreparedWeight::cp4); a tile without a partner zeroes the partner's half
      int *quad = a.colsum + epi_pair_offset_ints(a.N) + ((size_t)(16 * (ntile >> 5) + (ntile & 15)) * 16 + r) * 4;
      const int half = (ntile >> 4) & 1;
      quad[2 * half] = s;
      quad[2 * half + 1] = __float_as_int(pbv);
      // (N: the matrix's real column count -- with a device-side count the partner tile past it returns
      // early above and never writes its half, which an earlier, larger batch may have left behind)
      if (half == 0 && (ntile + 16) * 16 >= N) {
        quad[2] = 0;
        quad[3] = 0;
      }
    }
  }
}

// ---- merged launches (kernels.h, MergeIn / MergePack) ------------------------------------------------------------
// global sentence g of an encoder launch: its length (a hole: 0) ...
__device__ __forceinline__ int sentence_length(const FusedEncodeArgs &a, int g, int S) {
  if (a.n_sub == 0) return checked_length(a.lengths[g], S);
  const int j = merge_find(a.sub, a.n_sub, g), i = g - a.sub[j].first;
  return i < a.sub[j].n ? checked_length(a.sub[j].lengths[i], S) : 0;
}
// ... and its token ids: n of them (a hole: none; a sub-batch padded to fewer tokens than the launch: its own S). The
// caller embeds token 0 at the positions behind them -- padding, which nobody reads.
struct SentenceIds {
  const uint32_t *p;
  int n;
};
__device__ __forceinline__ SentenceIds sentence_ids(const FusedEncodeArgs &a, int g, int S) {
  if (a.n_sub == 0) return {a.ids + (size_t)g * S, S};
  const int j = merge_find(a.sub, a.n_sub, g), i = g - a.sub[j].first;
  if (i >= a.sub[j].n) return {nullptr, 0};
  return {a.sub[j].ids + (size_t)i * a.sub[j].S, a.sub[j].S};
}
// this workgroup's share (tiles first, first + step, ...) of the launch's packing jobs
__device__ __forceinline__ void pack_weight_share(const FusedEncodeArgs &a, int first, int step, int tid, int nthreads) {
  if (a.n_pack <= 1) {
    for (int pt = first; pt < a.pack_tiles; pt += step) pack_weight_tile(a.pack, pt, tid, nthreads);
    return;
  }
  for (int pt = first; pt < a.pack_tiles * a.n_pack; pt += step) {
    const int job = pt / a.pack_tiles, t = pt - job * a.pack_tiles;
    PackArgs pj = a.pack;
    pj.idx = a.pjob[job].idx;
    pj.N = a.pjob[job].N;
    pj.n_dev = a.pack.n_dev ? a.pack.n_dev + job : nullptr;  // (shortlists generated in this launch: job j's count is n_dev[j])
    pj.Wp = reinterpret_cast<char *>(a.pack.Wp) + (size_t)job * a.pack_stride_wp;
    pj.colsum = reinterpret_cast<int *>(reinterpret_cast<char *>(a.pack.colsum) + (size_t)job * a.pack_stride_cs);
    pj.pb = reinterpret_cast<float *>(reinterpret_cast<char *>(a.pack.pb) + (size_t)job * a.pack_stride_pb);
    pack_weight_tile(pj, t, tid, nthreads);
  }
}

}  // namespace slimt_hip
