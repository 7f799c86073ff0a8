// The persistent decoder's cross-attention over the PACKED K/V cache (kernels.h, FusedDecodeArgs::kv24 / kv_fmt): one body
// per SHAPE of sentence, one unpack policy per FORM of the cache. Textually included by decode_fused.hip inside its
// anonymous namespace (it uses that file's AttnRow, make_rsrc, the wave / half-wave reductions, quantize1_byte).
//
// Round 6: these were twelve hand-copied readers (24- / 20- / 16-bit x sentences of up to 32, 33..64, 65..128 tokens and
// d_head 64) plus eight out-of-line twins -- 1,470 lines that differed in how 32 cached values come back as floats and
// in nothing else. The policy is a compile-time type: no runtime branch enters the bodies (the one attempt at sharing them
// behind a uniform branch cost 40 spilled registers).
//
// The arithmetic is the hoisted (PORTABLE) order -- the CPU checker under oracle/ restates it as cross_attention_portable:
// the projections' unquantisation multiplier u and prepared bias pb are per-column constants, so they are applied AFTER the
// sums instead of to every cached value in every step:
//   t_j = fmaf chain over the head's columns of q_d * float(accK[j][d]);  c_h = row sum of q_d * pbK[d]
//   s_j = alpha * fmaf(t_j, uK, c_h) + mask_j;   p = softmax(s);   P_h = row sum of p
//   w_d = fmaf chain over the keys of p_j * float(accV[j][d]);            o_d = fmaf(w_d, uV, pbV[d] * P_h)
// (the reference dequantises first, Intgemm.inl.cc:146-153 / Modules.cc:24-86: same reals, other roundings). Every form
// hands the chains the SAME integers (times a power of two that its multiplier takes out again), so every form gives the
// same floats as every other and as the f32 cache.

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

// ---- the 24-bit form: any accumulator K = 256 can produce ---------------------------------------------------------------
// 12 bytes = four 24-bit accumulators accS -> float(accS * 256). A 24-bit field moved into the HIGH three bytes of a
// register is accS * 256 as a signed integer: one v_perm / shift per value instead of extract + sign-extend, and its
// conversion is exact (24 significant bits). The fmaf chains over 256 accS are 256 times those over accS (a power of two
// scales every partial sum exactly), and u / 256 takes the factor out again.
__device__ __forceinline__ f4 unpack24f(int d0, int d1, int d2) {
  const int y0 = d0 << 8;
  const int y1 = (int)__builtin_amdgcn_perm((unsigned)d1, (unsigned)d0, 0x0504030cu);
  const int y2 = (int)__builtin_amdgcn_perm((unsigned)d2, (unsigned)d1, 0x0403020cu);
  const int y3 = d2 & (int)0xffffff00;
  const f4 o = {(float)y0, (float)y1, (float)y2, (float)y3};
  return o;
}

// Layouts (per sentence): K [D/16][plane 0..2][S][16 B] -- 16 consecutive columns of one key = one quad in each of three
// planes --, V [ceil(S / 4)][plane 0..2][D/4][16 B] -- 4 keys x 4 consecutive columns, key-major.
struct Form24 {
  static constexpr int BITS = 24;
  static constexpr int KQ = 6;                // quads per key and 32 columns
  static constexpr int VQ = 3, VK = 4;        // quads / keys per V group
  static constexpr int NVF = 3, NVL = 5;      // V groups in flight: sentences of up to 64 tokens / longer ones
  float uk, uv;                               // u / 256 of the K / V projection
  struct VConst {};
  __device__ __forceinline__ VConst vconst(int) const { return {}; }
  // group g (12 bytes = 4 values) of the 48-byte item in planes p0, p1, p2
  static __device__ __forceinline__ f4 group(const v4i &p0, const v4i &p1, const v4i &p2, int g) {
    const int w[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
    return unpack24f(w[3 * g], w[3 * g + 1], w[3 * g + 2]);
  }
  // one key against the 32 columns d0 .. d0 + 31 of q: the ascending-column fmaf chain, continued in t. (The accumulator
  // travels by reference: returned by value, the same chains cost the S <= 32 kernels 23..45 spilled registers.)
  __device__ __forceinline__ void dot32(const v4i (&kq)[KQ], lcf_ptr q, int d0, float &t) const {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4 q4 = *(lcf4_ptr)(q + d0 + 16 * c + 4 * g);
        const f4 kk = group(kq[3 * c], kq[3 * c + 1], kq[3 * c + 2], g);
        t = __builtin_fmaf(q4.x, kk.x, t);
        t = __builtin_fmaf(q4.y, kk.y, t);
        t = __builtin_fmaf(q4.z, kk.z, t);
        t = __builtin_fmaf(q4.w, kk.w, t);
      }
      __builtin_amdgcn_sched_barrier(0);  // at most one chunk's q reads from LDS in flight
    }
  }
  // one V group (VK keys x this lane's four columns) times its keys' probabilities p[0 .. VK) (LDS), keys ascending
  __device__ __forceinline__ void acc(const v4i (&cur)[VQ], const SLIMT_LDS float *p, const VConst &, f2 &oa, f2 &ob) const {
    const f4 p4 = *(lcf4_ptr)p;
    const float pj[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // keys >= len: p == 0, fma(0, v, o) == o
      const f4 v4 = group(cur[0], cur[1], cur[2], c);
      const f2 pp = {pj[c], pj[c]}, va = {v4.x, v4.y}, vb = {v4.z, v4.w};
      oa = __builtin_elementwise_fma(pp, va, oa);
      ob = __builtin_elementwise_fma(pp, vb, ob);
    }
  }
};

// ---- the narrow form: 20 bits per value (kernels.h, FusedDecodeArgs::kv_fmt == 0) ------------------------------------------
// A sentence-layer whose K and V accumulators all lie in [-2^19, 2^19) is cached as hi = accS >> 4 (16 bits) and
// lo = accS & 15 (4 bits): 2.5 instead of 3 bytes per value, 5 instead of 6 sixteen-byte loads per 32 values. Eight values
// = one quad of hi halves + one dword of lo nibbles (device_common.h, pack20). Three bit operations per EIGHT values move
// the nibbles into the high halves of bytes (f0: even values, f1: odd ones); then one v_perm per value builds {hi byte 1,
// hi byte 0, lo << 4, 0} = accS << 12, whose conversion is exact (20 significant bits). The chains run 4096 times the
// accumulators' (exact, a power of two), and u / 4096 takes the factor out.
struct Lo20 {
  unsigned f0, f1;
};
__device__ __forceinline__ Lo20 expand20(int lo) {
  Lo20 e;
  e.f1 = (unsigned)lo & 0xf0f0f0f0u;
  e.f0 = ((unsigned)lo << 4) & 0xf0f0f0f0u;
  return e;
}
template <int C>  // value C (0..7) of a quad
__device__ __forceinline__ float unpack20(const v4i &hi, const Lo20 &e) {
  constexpr int k = C >> 1, n = C & 1;
  const unsigned hw = (unsigned)(k == 0 ? hi.x : k == 1 ? hi.y : k == 2 ? hi.z : hi.w);
  constexpr unsigned sel = ((4u + 2 * n + 1) << 24) | ((4u + 2 * n) << 16) | ((unsigned)k << 8) | 0x0cu;
  return (float)(int)__builtin_amdgcn_perm(hw, n ? e.f1 : e.f0, sel);
}

// Layouts (per sentence, in the slot the 24-bit form would take):
//   K [head][plane 0..4][S][16 B]          planes 0..3: hi halves of the head's columns 8 p .. 8 p + 7 of one key,
//                                           plane 4: the lo nibbles of all 32 (dword p belongs to plane p)
//   V [ceil(S / 8)][plane 0..4][D/4][16 B]  planes 0..3: keys 8 g + 2 p, 8 g + 2 p + 1 x 4 consecutive columns
//                                           (key-major), plane 4: their lo nibbles (dword p belongs to plane p)
struct Form20 {
  static constexpr int BITS = 20;
  static constexpr int KQ = 5, VQ = 5, VK = 8, NVF = 2, NVL = 3;
  float uk, uv;  // u / 4096 of the K / V projection
  struct VConst {};
  __device__ __forceinline__ VConst vconst(int) const { return {}; }
  // eight columns (one hi quad + its lo dword) of one key against q[0 .. 8): the chain continued in t
  static __device__ __forceinline__ void dot8(const v4i &hi, int lo, lcf_ptr q, float &t) {
    const f4 qa = *(lcf4_ptr)(q), qb = *(lcf4_ptr)(q + 4);
    const Lo20 e = expand20(lo);
    t = __builtin_fmaf(qa.x, unpack20<0>(hi, e), t);
    t = __builtin_fmaf(qa.y, unpack20<1>(hi, e), t);
    t = __builtin_fmaf(qa.z, unpack20<2>(hi, e), t);
    t = __builtin_fmaf(qa.w, unpack20<3>(hi, e), t);
    t = __builtin_fmaf(qb.x, unpack20<4>(hi, e), t);
    t = __builtin_fmaf(qb.y, unpack20<5>(hi, e), t);
    t = __builtin_fmaf(qb.z, unpack20<6>(hi, e), t);
    t = __builtin_fmaf(qb.w, unpack20<7>(hi, e), t);
  }
  __device__ __forceinline__ void dot32(const v4i (&kq)[KQ], lcf_ptr q, int d0, float &t) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lo[4] = {kq[4].x, kq[4].y, kq[4].z, kq[4].w};
      dot8(kq[i], lo[i], q + (d0 + 8 * i), t);
      if (i & 1) __builtin_amdgcn_sched_barrier(0);  // at most sixteen q values from LDS in flight
    }
  }
  // d_head 64: planes 0..7 the hi halves of the head's 64 columns, planes 8 and 9 the lo nibbles of columns 0..31 / 32..63
  __device__ __forceinline__ void dot64(const v4i (&kq)[2 * KQ], lcf_ptr q, int d0, float &t) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const v4i &lq = kq[8 + (i >> 2)];
      const int lo[4] = {lq.x, lq.y, lq.z, lq.w};
      dot8(kq[i], lo[i & 3], q + (d0 + 8 * i), t);
      if (i & 1) __builtin_amdgcn_sched_barrier(0);  // at most sixteen q values from LDS in flight
    }
  }
  __device__ __forceinline__ void acc(const v4i (&cur)[VQ], const SLIMT_LDS float *p, const VConst &, f2 &oa, f2 &ob) const {
    const int lo[4] = {cur[4].x, cur[4].y, cur[4].z, cur[4].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // keys 8 g + 2 c, 8 g + 2 c + 1 (keys >= len: p == 0, fma(0, v, o) == o)
      const Lo20 e = expand20(lo[c]);
      const f2 pp = *(const SLIMT_LDS f2 *)(p + 2 * c);  // (two at a time: eight would be eight registers)
      const f2 p0 = {pp.x, pp.x}, p1 = {pp.y, pp.y};
      const f2 va0 = {unpack20<0>(cur[c], e), unpack20<1>(cur[c], e)}, vb0 = {unpack20<2>(cur[c], e), unpack20<3>(cur[c], e)};
      const f2 va1 = {unpack20<4>(cur[c], e), unpack20<5>(cur[c], e)}, vb1 = {unpack20<6>(cur[c], e), unpack20<7>(cur[c], e)};
      oa = __builtin_elementwise_fma(p0, va0, oa);
      ob = __builtin_elementwise_fma(p0, vb0, ob);
      oa = __builtin_elementwise_fma(p1, va1, oa);
      ob = __builtin_elementwise_fma(p1, vb1, ob);
    }
  }
};

// ---- the tight form: 16 bits per value (kernels.h, FusedDecodeArgs::kv_fmt == 2) --------------------------------------------
// A sentence-layer whose K and V accumulators, less their columns' centres (r = accS - centre[d]; kernels.h,
// FusedDecodeArgs::kv_centre: the column means of a calibration batch), all lie in [-2^15, 2^15) is cached as plain int16:
// 2 bytes per value, 4 loads per 32 values, and the cheapest unpack of the three forms -- the conversion reads its half of
// the register itself (v_cvt_f32_i32 with an SDWA word select: no extract at all), and float(accS) = float(r) +
// float(centre) is one EXACT addition (both floats are integers, their sum is accS, everything below 2^24), half a packed
// add per value: 1.5 instructions per value where the 20-bit form spends 2.4 and the 24-bit one 2. The chains run over
// float(accS) itself (no power of two to take out: plain u). This is the ONE form inlined in its kernels (KVI = 16); the
// 20- and the 24-bit ones are out-of-line fallbacks there.
//   K [head][plane 0..3][S][16 B]            plane p: the head's columns 8 p .. 8 p + 7 of one key (eight int16)
//   V [ceil(S / 8)][plane 0..3][D/4][16 B]   plane p: keys 8 g + 2 p, 8 g + 2 p + 1 x 4 consecutive columns (key-major)
// CK / CV: where the centres of the K / V projection [D] are read from.
struct CentreLds {  // in LDS, as floats, next to the prepared biases
  lcf_ptr p;
  __device__ __forceinline__ f4 at4(int d) const { return *(lcf4_ptr)(p + d); }
};
struct CentreGlobal {  // the V pass reads its four once per row: where LDS is short (the 32-sentence tiling) they stay in global memory
  const int *p;
  __device__ __forceinline__ f4 at4(int d) const {
    const v4i c = *reinterpret_cast<const v4i *>(p + d);
    return f4{(float)c.x, (float)c.y, (float)c.z, (float)c.w};
  }
};
template <class CK, class CV>
struct Form16 {
  static constexpr int BITS = 16;
  static constexpr int KQ = 4, VQ = 4, VK = 8, NVF = 2, NVL = 3;
  float uk, uv;  // the K / V projection's u itself
  CK ck;
  CV cv;
  struct VConst {
    f2 c01, c23;
  };
  // the centres of this lane's four V columns d .. d + 3
  __device__ __forceinline__ VConst vconst(int d) const {
    const f4 cv4 = cv.at4(d);
    return {f2{cv4.x, cv4.y}, f2{cv4.z, cv4.w}};
  }
  // two int16 of one register -> float(accS) of the two columns / keys: conversion in place + the column's centre (exact)
  static __device__ __forceinline__ f2 pair16(int d, f2 c) {
    const f2 v = {(float)(short)(d & 0xffff), (float)(d >> 16)};
    return v + c;
  }
  // eight columns d .. d + 7 (one quad of int16) of one key: the chain continued in t
  __device__ __forceinline__ void dot8(const v4i &k8, lcf_ptr q, int d, float &t) const {
    const f4 qa = *(lcf4_ptr)(q + d), qb = *(lcf4_ptr)(q + d + 4);
    const f4 ca = ck.at4(d), cb = ck.at4(d + 4);
    const f2 k01 = pair16(k8.x, f2{ca.x, ca.y}), k23 = pair16(k8.y, f2{ca.z, ca.w});
    const f2 k45 = pair16(k8.z, f2{cb.x, cb.y}), k67 = pair16(k8.w, f2{cb.z, cb.w});
    t = __builtin_fmaf(qa.x, k01.x, t);
    t = __builtin_fmaf(qa.y, k01.y, t);
    t = __builtin_fmaf(qa.z, k23.x, t);
    t = __builtin_fmaf(qa.w, k23.y, t);
    t = __builtin_fmaf(qb.x, k45.x, t);
    t = __builtin_fmaf(qb.y, k45.y, t);
    t = __builtin_fmaf(qb.z, k67.x, t);
    t = __builtin_fmaf(qb.w, k67.y, t);
  }
  __device__ __forceinline__ void dot32(const v4i (&kq)[KQ], lcf_ptr q, int d0, float &t) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dot8(kq[i], q, d0 + 8 * i, t);
      if (i & 1) __builtin_amdgcn_sched_barrier(0);  // at most sixteen q / centre values from LDS in flight
    }
  }
  __device__ __forceinline__ void dot64(const v4i (&kq)[2 * KQ], lcf_ptr q, int d0, float &t) const {  // d_head 64: eight planes
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dot8(kq[i], q, d0 + 8 * i, t);
      if (i & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  __device__ __forceinline__ void acc(const v4i (&cur)[VQ], const SLIMT_LDS float *p, const VConst &vc, f2 &oa, f2 &ob) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // keys 8 g + 2 c, 8 g + 2 c + 1 (keys >= len: p == 0, fma(0, v, o) == o)
      const f2 pp = *(const SLIMT_LDS f2 *)(p + 2 * c);
      const f2 p0 = {pp.x, pp.x}, p1 = {pp.y, pp.y};
      oa = __builtin_elementwise_fma(p0, pair16(cur[c].x, vc.c01), oa);
      ob = __builtin_elementwise_fma(p0, pair16(cur[c].y, vc.c23), ob);
      oa = __builtin_elementwise_fma(p1, pair16(cur[c].z, vc.c01), oa);
      ob = __builtin_elementwise_fma(p1, pair16(cur[c].w, vc.c23), ob);
    }
  }
};

// c_h = row sum (canonical order) of q_d * pbK[d] over head h's DH columns, for the heads this lane meets in the score
// passes. DH = 32, D = 256: lane l holds column l + 64 i of head 2 i + (l >> 5) -- the head of pass i in the
// two-heads-per-pass shape, whose lanes of half hh score head 2 i + hh: no shuffle at all.
__device__ __forceinline__ void head_constants32(lcf_ptr q, lcf_ptr pbk, int lane, float (&c)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = half_sum(q[lane + 64 * i] * pbk[lane + 64 * i]);
}

// this lane's four output columns: o_d = fmaf(w_d, uV, pbV[d] * P_h), quantised for the O projection
__device__ __forceinline__ int attention_out4(const f2 &oa, const f2 &ob, float uv, const f4 &pv4, float P, float aq_o) {
  const float o0 = __builtin_fmaf(oa.x, uv, pv4.x * P), o1 = __builtin_fmaf(oa.y, uv, pv4.y * P);
  const float o2 = __builtin_fmaf(ob.x, uv, pv4.z * P), o3 = __builtin_fmaf(ob.y, uv, pv4.w * P);
  return pack4(quantize1_byte(o0, aq_o), quantize1_byte(o1, aq_o), quantize1_byte(o2, aq_o), quantize1_byte(o3, aq_o));
}

// ---- S <= 32, d_head 32, D = 256 -------------------------------------------------------------------------------------------
// Lane = (half hh, key j): two heads per score pass, the 32-column softmax on half waves, all heads' probabilities in LDS
// (pbuf [H][32]), then V as whole rows (lane l: head l / 8, columns 4 l .. 4 l + 3; keys ascending). pbk / pbv: the K / V
// projections' prepared biases [D] in LDS. Masked keys are not fetched: past the descriptors a load returns zeros -- a
// score that the mask overrides, a value weighted by a probability that is exactly 0.
template <int KV_AUX, class F>
__device__ __forceinline__ void attention_packed32(AttnRow r, int lane, lcf_ptr pbk, lcf_ptr pbv, const F f) {
  constexpr int D = 256, DH = 32, H = D / DH;
  constexpr int KQ = F::KQ, VQ = F::VQ, VK = F::VK, NVF = F::NVF, NG = 32 / VK;
  const int S = r.S, len = r.len;
  const int lenf = len > 0 ? len : S;
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  const int hh = lane >> 5, j = lane & 31;
  const int jc = j < S ? j : S - 1;
  const float mask = (1.0f - (j < len ? 1.0f : 0.0f)) * minus_inf;
  v4i kq[KQ];        // this lane's key, its head's 32 columns
  v4i vq[NVF][VQ];   // V rows in flight: NVF groups of VK rows
  const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * (KQ * 16 * H)));
  const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(((__builtin_amdgcn_readfirstlane(lenf) + VK - 1) / VK) * (VQ * 1024)));
  const int koff = j < lenf ? (hh * KQ * S + jc) * 16 : kPastDescriptor;
  const int voff = lane * 16;
  auto load_k = [&](int hp) {
#pragma unroll
    for (int i = 0; i < KQ; ++i)
      kq[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, ((2 * KQ * hp + i) * S) * 16, KV_AUX));
  };
  auto load_v = [&](v4i(&vv)[VQ], int g) {  // rows VK g .. VK g + VK - 1
#pragma unroll
    for (int i = 0; i < VQ; ++i)
      vv[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, (VQ * g + i) * 1024, KV_AUX));
  };
  load_k(0);
  load_v(vq[0], 0);
  __builtin_amdgcn_sched_barrier(0);
  float ckh[4];  // c_h of this lane's head in pass hp (under the first loads' round trip)
  head_constants32(r.qrow, pbk, lane, ckh);
#pragma unroll
  for (int hp = 0; hp < H / 2; ++hp) {
    const int h = 2 * hp + hh;
    float t = 0.0f;
    f.dot32(kq, r.qrow, h * DH, t);
    if (hp + 1 < H / 2) {
      load_k(hp + 1);
    } else {  // the K registers are free: the other groups of V rows
#pragma unroll
      for (int k = 1; k < NVF; ++k) load_v(vq[k], k);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = __builtin_fmaf(t, f.uk, ckh[hp]);
    if (r.alpha != 1.0f) s = r.alpha * s;
    s = s + mask;
    if (j >= S) s = lowest;
    const float m = half_max(s);
    const float e = j < S ? exp_p(s - m) : 0.0f;
    const float sum = half_sum(e);
    const float p = e / sum;  // keys >= S: exactly 0
    const float ps = half_sum(p);  // P_h
    if (r.attn && j < S) r.attn[(size_t)h * S + j] = p;
    if (r.align && hp == 0 && hh == 0 && j < len) r.align[j] = p;
    r.pbuf[h * 32 + j] = p;
    if (j == 0) r.hsum[h] = ps;
  }
  const int ph = (lane >> 3) * 32;
  const f4 pv4 = *(lcf4_ptr)(pbv + 4 * lane);
  const typename F::VConst vc = f.vconst(4 * lane);
  const float P = r.hsum[lane >> 3];
  f2 oa = {0.0f, 0.0f}, ob = {0.0f, 0.0f};  // columns (0, 1) and (2, 3): v_pk_fma_f32 is one fma per column
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f.acc(vq[g % NVF], r.pbuf + ph + VK * g, vc, oa, ob);
    // pin this group's sums here: the unpack + fma chains are pure arithmetic, and without a use the optimiser sinks every
    // group's work below the last load -- all the packed rows live at once (seen as 134 spilled VGPRs)
    asm volatile("" : "+v"(oa), "+v"(ob));
    if (g + NVF < NG) load_v(vq[g % NVF], g + NVF);
    __builtin_amdgcn_sched_barrier(0);
  }
  *(SLIMT_LDS int *)(r.arow + 4 * lane) = attention_out4(oa, ob, f.uv, pv4, P, r.aq_o);
}

// ---- 33..64 tokens (written by encode_tall_kernel<., 4>) ---------------------------------------------------------------------
// Lane = key, one head per score pass, the 64-column softmax in the canonical order (one element per lane, the 64-lane
// butterfly), all heads' probabilities in LDS (pbuf [H][64]), then V as whole rows like the S <= 32 shape.
// PART / SPW: as for attention_packed128 below (1 = the score passes of heads wave / SPW, + 16 / SPW, ...; 2 = the context pass).
template <int KV_AUX, class F, int PART = 0, int SPW = 16>
__device__ __forceinline__ void attention_packed64(AttnRow r, int lane, lcf_ptr pbk, lcf_ptr pbv, const F f) {
  constexpr int D = 256, DH = 32, H = D / DH;
  constexpr int KQ = F::KQ, VQ = F::VQ, VK = F::VK, NVF = F::NVF, NG = 64 / VK;
  const int S = r.S, len = r.len;
  const int lenf = len > 0 ? len : S;
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  const int j = lane;
  const int jc = j < S ? j : S - 1;
  const float mask = (1.0f - (j < len ? 1.0f : 0.0f)) * minus_inf;
  const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * (KQ * 16 * H)));
  const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(((__builtin_amdgcn_readfirstlane(lenf) + VK - 1) / VK) * (VQ * 1024)));
  const int koff = j < lenf ? jc * 16 : kPastDescriptor;
  const int voff = lane * 16;
  const int h_first = PART == 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) / SPW : 0;
  constexpr int h_step = PART == 1 ? 16 / SPW : 1;
  if constexpr (PART != 2) {
  v4i kq[KQ];
  auto load_k = [&](int h) {
#pragma unroll
    for (int i = 0; i < KQ; ++i)
      kq[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, ((KQ * h + i) * S) * 16, KV_AUX));
  };
  load_k(h_first);
  {  // c_h of every head -> hsum[8 + h] (the sentence's scratch; read back wave-uniformly in pass h)
    float ckh[4];
    head_constants32(r.qrow, pbk, lane, ckh);
    if ((lane & 31) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r.hsum[8 + 2 * i + (lane >> 5)] = ckh[i];
    }
  }
#pragma unroll 1
  for (int h = h_first; h < H; h += h_step) {
    float t = 0.0f;
    f.dot32(kq, r.qrow, h * DH, t);
    load_k(h + h_step < H ? h + h_step : h);  // the next head's keys travel under this head's softmax
    __builtin_amdgcn_sched_barrier(0);
    float s = __builtin_fmaf(t, f.uk, r.hsum[8 + h]);
    if (r.alpha != 1.0f) s = r.alpha * s;
    s = s + mask;
    if (j >= S) s = lowest;
    const float m = wave_max(s);
    const float e = j < S ? exp_p(s - m) : 0.0f;
    const float sum = wave_sum(e);
    const float p = e / sum;  // keys >= S: exactly 0
    const float ps = wave_sum(p);  // P_h
    if (r.attn && j < S) r.attn[(size_t)h * S + j] = p;
    if (r.align && h == 0 && j < len) r.align[j] = p;
    r.pbuf[h * 64 + j] = p;
    if (lane == 0) r.hsum[h] = ps;
  }
  }  // PART != 2
  if constexpr (PART == 1) return;
  v4i vq[NVF][VQ];  // V rows in flight
  auto load_v = [&](v4i(&vv)[VQ], int g) {  // rows VK g .. VK g + VK - 1
#pragma unroll
    for (int i = 0; i < VQ; ++i)
      vv[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, (VQ * g + i) * 1024, KV_AUX));
  };
#pragma unroll
  for (int k = 0; k < NVF; ++k) load_v(vq[k], k);
  __builtin_amdgcn_sched_barrier(0);
  const int ph = (lane >> 3) * 64;
  const f4 pv4 = *(lcf4_ptr)(pbv + 4 * lane);
  const typename F::VConst vc = f.vconst(4 * lane);
  const float P = r.hsum[lane >> 3];
  f2 oa = {0.0f, 0.0f}, ob = {0.0f, 0.0f};
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f.acc(vq[g % NVF], r.pbuf + ph + VK * g, vc, oa, ob);
    asm volatile("" : "+v"(oa), "+v"(ob));  // pin this group's sums here (see attention_packed32)
    if (g + NVF < NG) load_v(vq[g % NVF], g + NVF);
    __builtin_amdgcn_sched_barrier(0);
  }
  *(SLIMT_LDS int *)(r.arow + 4 * lane) = attention_out4(oa, ob, f.uv, pv4, P, r.aq_o);
}

// ---- 65..128 tokens (written by encode_long16_kernel) ----------------------------------------------------------------------
// Lane L holds keys L and L + 64 -- one head per score pass, in two half passes --, the 128-column softmax in the canonical
// order (lane L first adds keys L and L + 64, then the 64-lane butterfly), all heads' probabilities in LDS (pbuf
// [H][128]), then V as whole rows, key groups past the sentence skipped.
// Two K buffers: a half pass's keys are requested one half pass ahead (keys L + 64 of head h before keys L are scored, keys
// L of head h + 1 before keys L + 64 are), so that a round trip to the cache runs under 64 values' worth of unpacking
// instead of in front of it. No store inside the loop (the probabilities leave through LDS): loads and stores share one
// counter, and a conditional store would make the compiler drain every load in flight at the loop head.
// PART: 0 = the whole attention by this wave; 1 = the score passes of heads wave / SPW, + 16 / SPW, ... only (probabilities and
// P_h into r.pbuf / r.hsum: the sentence's owner and the idle waves of an SPW-sentence workgroup share the heads); 2 = the
// context pass only, from the probabilities all of them left there (a workgroup barrier lies between the two).
template <int KV_AUX, class F, int PART = 0, int SPW = 16>
__device__ __forceinline__ void attention_packed128(AttnRow r, int lane, lcf_ptr pbk, lcf_ptr pbv, const F f) {
  constexpr int D = 256, DH = 32, H = D / DH;
  constexpr int KQ = F::KQ, VQ = F::VQ, VK = F::VK, NV = F::NVL;
  const int S = __builtin_amdgcn_readfirstlane(r.S), len = __builtin_amdgcn_readfirstlane(r.len);
  const int lenf = len > 0 ? len : S;
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  const int j0 = lane, j1 = lane + 64;
  const float mask0 = (1.0f - (j0 < len ? 1.0f : 0.0f)) * minus_inf;
  const float mask1 = (1.0f - (j1 < len ? 1.0f : 0.0f)) * minus_inf;
  const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * (KQ * 16 * H)));
  const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(((lenf + VK - 1) / VK) * (VQ * 1024)));
  // masked keys are not fetched (zeros past the descriptor: the value pb, weighted by exactly 0)
  const int koff0 = j0 < lenf ? j0 * 16 : kPastDescriptor;
  const int koff1 = j1 < lenf ? j1 * 16 : kPastDescriptor;
  const int voff = lane * 16;
  const int h_first = PART == 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) / SPW : 0;
  constexpr int h_step = PART == 1 ? 16 / SPW : 1;
  if constexpr (PART != 2) {
  v4i ka[KQ], kb[KQ];
  auto load_k = [&](v4i(&kq)[KQ], int h, int koff) {
#pragma unroll
    for (int i = 0; i < KQ; ++i)
      kq[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, ((KQ * h + i) * S) * 16, KV_AUX));
  };
  load_k(ka, h_first, koff0);
  {  // c_h of every head -> hsum[8 + h] (the sentence's scratch; read back wave-uniformly in pass h -- sharing waves all write
     // the same eight values)
    float ckh[4];
    head_constants32(r.qrow, pbk, lane, ckh);
    if ((lane & 31) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r.hsum[8 + 2 * i + (lane >> 5)] = ckh[i];
    }
  }
  auto head = [&](int h, bool last) {
    load_k(kb, h, koff1);
    __builtin_amdgcn_sched_barrier(0);
    float s0 = 0.0f, s1 = 0.0f;  // this lane's two keys against head h: the ascending-column fmaf chains t_j
    f.dot32(ka, r.qrow, h * DH, s0);
    if (!last) load_k(ka, h + h_step, koff0);  // (the last head requests nothing it would have to wait out again)
    __builtin_amdgcn_sched_barrier(0);
    f.dot32(kb, r.qrow, h * DH, s1);
    const float ch = r.hsum[8 + h];
    s0 = __builtin_fmaf(s0, f.uk, ch);
    s1 = __builtin_fmaf(s1, f.uk, ch);
    if (r.alpha != 1.0f) {
      s0 = r.alpha * s0;
      s1 = r.alpha * s1;
    }
    s0 = s0 + mask0;
    s1 = s1 + mask1;
    if (j0 >= S) s0 = lowest;
    if (j1 >= S) s1 = lowest;
    const float m = wave_max(fmaxf(s0, s1));
    const float e0 = j0 < S ? exp_p(s0 - m) : 0.0f;
    const float e1 = j1 < S ? exp_p(s1 - m) : 0.0f;
    const float sum = wave_sum(e0 + e1);
    const float p0 = e0 / sum, p1 = e1 / sum;  // keys >= S: exactly 0
    r.pbuf[h * 128 + j0] = p0;
    r.pbuf[h * 128 + j1] = p1;
    const float ps = wave_sum(p0 + p1);  // P_h: lane L adds keys L and L + 64, then the butterfly
    if (lane == 0) r.hsum[h] = ps;
  };
  {
    int h = h_first;
#pragma unroll 1
    for (; h + h_step < H; h += h_step) head(h, false);
    head(h, true);
  }
  if (r.align) {  // head 0 over the sentence's own keys (update_alignment, Model.cc:84-108)
    if (j0 < len) r.align[j0] = r.pbuf[j0];
    if (j1 < len) r.align[j1] = r.pbuf[j1];
  }
  if (r.attn) {
    for (int h = h_first; h < H; h += h_step) {
      if (j0 < S) r.attn[(size_t)h * S + j0] = r.pbuf[h * 128 + j0];
      if (j1 < S) r.attn[(size_t)h * S + j1] = r.pbuf[h * 128 + j1];
    }
  }
  }  // PART != 2
  if constexpr (PART == 1) return;
  v4i vq[NV][VQ];  // V key groups in flight
  auto load_v = [&](v4i(&vv)[VQ], int g) {  // rows VK g .. VK g + VK - 1 (past the descriptor: zeros)
#pragma unroll
    for (int i = 0; i < VQ; ++i)
      vv[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, (VQ * g + i) * 1024, KV_AUX));
  };
  __builtin_amdgcn_sched_barrier(0);
  // the groups are requested in the order the loop re-requests them, nothing else behind them: the pending-load order at
  // the loop head is then the same from both of its entries (exact waits)
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    load_v(vq[k], k);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int ph = (lane >> 3) * 128;
  const f4 pv4 = *(lcf4_ptr)(pbv + 4 * lane);
  const typename F::VConst vc = f.vconst(4 * lane);
  const float P = r.hsum[lane >> 3];
  f2 oa = {0.0f, 0.0f}, ob = {0.0f, 0.0f};
  const int ng = (lenf + VK - 1) / VK;  // key groups that hold a key with a non-zero weight
  auto group = [&](const v4i(&cur)[VQ], int g) {
    if (g >= ng) return;  // (uniform) past the sentence: pbuf holds the next head's probabilities there
    f.acc(cur, r.pbuf + ph + VK * g, vc, oa, ob);
    asm volatile("" : "+v"(oa), "+v"(ob));  // pin this group's sums here (see attention_packed32)
  };
#pragma unroll 1
  for (int g = 0; g < ng; g += NV) {  // (groups past ng inside the last round are skipped; their rows: zeros)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      group(vq[k], g + k);
      load_v(vq[k], g + k + NV);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  *(SLIMT_LDS int *)(r.arow + 4 * lane) = attention_out4(oa, ob, f.uv, pv4, P, r.aq_o);
}

// ---- D = 512, d_head 64 ("base"), S <= 32 ------------------------------------------------------------------------------------
// The narrow and the tight form cache the SHIFTED accumulator like at D = 256 (20 / 16 bits hold it where they hold
// anything), in the layouts above with two quads' worth of columns per head: K [head][plane 0 .. 2 KQ - 1][S][16 B] (the
// narrow form: planes 0..7 hi halves, 8 and 9 the lo nibbles of columns 0..31 / 32..63), V as at D = 256 with 128 column
// quads per plane (a lane owns quads lane and 64 + lane). Two heads per score pass (lanes 0..31 / 32..63).
// kc: LDS constants of this layer, [K pb | K c127 | V pb | V c127][D] (the c127 vectors serve the 24-bit form below only).
template <int KV_AUX, class F>
__device__ __forceinline__ void attention_packed32_d64(AttnRow r, int lane, lcf_ptr kc, const F f) {
  constexpr int D = 512, DH = 64, H = D / DH;
  constexpr int KQ = F::KQ, VQ = F::VQ, VK = F::VK, NG = 32 / VK;
  static_assert(F::NVF == 2 && VK == 8, "the narrow and the tight form");
  const int S = r.S, len = r.len;
  const int lenf = len > 0 ? len : S;
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  const int hh = lane >> 5, j = lane & 31;
  const int jc = j < S ? j : S - 1;
  const float mask = (1.0f - (j < len ? 1.0f : 0.0f)) * minus_inf;
  const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * (2 * KQ * 16 * H)));
  const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(((__builtin_amdgcn_readfirstlane(lenf) + 7) >> 3) * (VQ * 2048)));
  const int koff = j < lenf ? (hh * 2 * KQ * S + jc) * 16 : kPastDescriptor;
  const lcf_ptr kpb = kc, vpb = kc + 2 * D;
#pragma unroll 1
  for (int hp = 0; hp < H / 2; ++hp) {
    const int h = 2 * hp + hh;
    v4i kq[2 * KQ];  // this lane's key, its head's 64 columns
#pragma unroll
    for (int i = 0; i < 2 * KQ; ++i)
      kq[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, ((4 * KQ * hp + i) * S) * 16, KV_AUX));
    // c_h = row sum of q_d * pbK[d] over the 64 columns of this lane's head (one per lane, the canonical 64-lane
    // butterfly; both heads of the pass, under the loads' round trip)
    const float c0 = wave_sum(r.qrow[(2 * hp) * DH + lane] * kpb[(2 * hp) * DH + lane]);
    const float c1 = wave_sum(r.qrow[(2 * hp + 1) * DH + lane] * kpb[(2 * hp + 1) * DH + lane]);
    const float ch = hh ? c1 : c0;
    float t = 0.0f;
    f.dot64(kq, r.qrow, h * DH, t);
    float s = __builtin_fmaf(t, f.uk, ch);
    if (r.alpha != 1.0f) s = r.alpha * s;
    s = s + mask;
    if (j >= S) s = lowest;
    const float m = half_max(s);
    const float e = j < S ? exp_p(s - m) : 0.0f;
    const float sum = half_sum(e);
    const float p = e / sum;  // keys >= S: exactly 0
    const float ps = half_sum(p);  // P_h
    if (r.attn && j < S) r.attn[(size_t)h * S + j] = p;
    if (r.align && h == 0 && j < len) r.align[j] = p;
    r.pbuf[h * 32 + j] = p;
    if (j == 0) r.hsum[h] = ps;
  }
  // V: a lane owns column quads `lane` (head lane / 16) and 64 + lane (head 4 + lane / 16): two independent passes over
  // the keys, each with two groups of eight rows in flight
  int packed[2];
#pragma unroll
  for (int slot = 0; slot < 2; ++slot) {
    const int voff = (slot * 64 + lane) * 16;
    v4i vq[2][VQ];
    auto load_v = [&](v4i(&vv)[VQ], int g) {  // rows 8 g .. 8 g + 7
#pragma unroll
      for (int i = 0; i < VQ; ++i)
        vv[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, (VQ * g + i) * 2048, KV_AUX));
    };
    load_v(vq[0], 0);
    load_v(vq[1], 1);
    __builtin_amdgcn_sched_barrier(0);
    const int head = 4 * slot + (lane >> 4);
    const int ph = head * 32;
    const f4 pv4 = *(lcf4_ptr)(vpb + slot * (D / 2) + 4 * lane);
    const typename F::VConst vc = f.vconst(slot * (D / 2) + 4 * lane);
    const float P = r.hsum[head];
    f2 oa = {0.0f, 0.0f}, ob = {0.0f, 0.0f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f.acc(vq[g & 1], r.pbuf + ph + 8 * g, vc, oa, ob);
      asm volatile("" : "+v"(oa), "+v"(ob));  // (attention_packed32: keeps a group's work next to its loads)
      if (g + 2 < NG) load_v(vq[g & 1], g + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    packed[slot] = attention_out4(oa, ob, f.uv, pv4, P, r.aq_o);
  }
  *(SLIMT_LDS int *)(r.arow + 4 * lane) = packed[0];
  *(SLIMT_LDS int *)(r.arow + D / 2 + 4 * lane) = packed[1];
}

// The 24-bit form at D = 512, d_head 64 ("base"). At K = 512 the shifted accumulator needs 25 bits, so
// the cache holds the SIGNED one (|acc| <= 127 * 128 * 512 < 2^23) and the column's 127 colsum term
// comes back here: c127 = float(127 colsum * 256) is exact, and so is float(acc * 256) + c127
// (= 256 accS, |accS| < 2^24): float(accS * 256) as unpack24f gives it, at one exact add per value.
__device__ __forceinline__ f4 unpack24cf(int d0, int d1, int d2, f4 c127) {
  const int y0 = d0 << 8;
  const int y1 = (int)__builtin_amdgcn_perm((unsigned)d1, (unsigned)d0, 0x0504030cu);
  const int y2 = (int)__builtin_amdgcn_perm((unsigned)d2, (unsigned)d1, 0x0403020cu);
  const int y3 = d2 & (int)0xffffff00;
  f2 a = {(float)y0, (float)y1}, b = {(float)y2, (float)y3};
  const f2 ca = {c127.x, c127.y}, cb = {c127.z, c127.w};
  a = a + ca;
  b = b + cb;
  const f4 o = {a.x, a.y, b.x, b.y};
  return o;
}

// kc: LDS constants of this layer, [K pb | K c127 | V pb | V c127][D]
template <int KV_AUX>
__device__ __forceinline__ void attention_row24_64(AttnRow r, int lane, lcf_ptr kc, float uk256, float uv256) {
  constexpr int D = 512, DH = 64, H = D / DH;
  const int S = r.S, len = r.len;
  const int lenf = len > 0 ? len : S;
  const float minus_inf = -99999999.0f;  // Input.cc:56-61
  const float lowest = -3.402823466e+38f;
  const int j = lane & 31;
  const int jc = j < S ? j : S - 1;
  const float mask = (1.0f - (j < len ? 1.0f : 0.0f)) * minus_inf;
  const rsrc_t rk = make_rsrc((const float *)r.kl, (unsigned)(S * D) * 3u);
  const rsrc_t rv = make_rsrc((const float *)r.vl, (unsigned)(((__builtin_amdgcn_readfirstlane(lenf) + 3) >> 2) * (3 * (D / 4) * 16)));
  const int koff = j < lenf ? jc * 16 : kPastDescriptor;  // [D/16][plane][S][16 B]
  const int voff = lane * 16;                             // [S/4][plane][D/4][16 B]: slots lane and 64 + lane
  auto group = [](const v4i &p0, const v4i &p1, const v4i &p2, int g, int w) -> int {
    const int d[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
    return d[3 * g + w];
  };
  const lcf_ptr kpb = kc, kcs = kc + D, vpb = kc + 2 * D, vcs = kc + 3 * D;
#pragma unroll 1
  for (int h = 0; h < H; ++h) {
    v4i kq[12];  // this lane's key, the head's 64 columns: four chunks of three planes
#pragma unroll
    for (int i = 0; i < 12; ++i)
      kq[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, ((12 * h + i) * S) * 16, KV_AUX));
    // c_h = row sum of q_d * pbK[d] over the head's 64 columns (one per lane, the canonical 64-lane butterfly)
    const float ch = wave_sum(r.qrow[h * DH + lane] * kpb[h * DH + lane]);
    float t = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = h * DH + 16 * c + 4 * g;
        const f4 q4 = *(lcf4_ptr)(r.qrow + d0);
        const f4 kk = unpack24cf(group(kq[3 * c], kq[3 * c + 1], kq[3 * c + 2], g, 0), group(kq[3 * c], kq[3 * c + 1], kq[3 * c + 2], g, 1),
                                 group(kq[3 * c], kq[3 * c + 1], kq[3 * c + 2], g, 2), *(lcf4_ptr)(kcs + d0));
        t = __builtin_fmaf(q4.x, kk.x, t);
        t = __builtin_fmaf(q4.y, kk.y, t);
        t = __builtin_fmaf(q4.z, kk.z, t);
        t = __builtin_fmaf(q4.w, kk.w, t);
      }
      __builtin_amdgcn_sched_barrier(0);  // one chunk's q / constant reads from LDS in flight
    }
    float s = __builtin_fmaf(t, uk256, ch);
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
  v4i vq[2][6];  // V rows in flight: two groups of four rows, two column slots of three planes each
  auto load_v = [&](v4i(&vv)[6], int g) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      vv[p] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, (3 * g + p) * (D / 4) * 16, KV_AUX));
      vv[3 + p] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, (3 * g + p) * (D / 4) * 16 + 1024, KV_AUX));
    }
  };
  load_v(vq[0], 0);
  load_v(vq[1], 1);
  __builtin_amdgcn_sched_barrier(0);
  const int ph0 = (lane >> 4) * 32, ph1 = (4 + (lane >> 4)) * 32;
  const f4 pv0 = *(lcf4_ptr)(vpb + 4 * lane), pv1 = *(lcf4_ptr)(vpb + D / 2 + 4 * lane);
  const f4 cv0 = *(lcf4_ptr)(vcs + 4 * lane), cv1 = *(lcf4_ptr)(vcs + D / 2 + 4 * lane);
  const float P0 = r.hsum[lane >> 4], P1 = r.hsum[4 + (lane >> 4)];
  f2 o0a = {0.0f, 0.0f}, o0b = {0.0f, 0.0f}, o1a = {0.0f, 0.0f}, o1b = {0.0f, 0.0f};  // column pairs: one v_pk_fma_f32 each
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    v4i(&cur)[6] = vq[g % 2];
    const f4 pa4 = *(lcf4_ptr)(r.pbuf + ph0 + 4 * g), pb4 = *(lcf4_ptr)(r.pbuf + ph1 + 4 * g);
    const float pa[4] = {pa4.x, pa4.y, pa4.z, pa4.w}, pb_[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // keys >= len: p == 0, fma(0, v, o) == o
      const f4 v0 = unpack24cf(group(cur[0], cur[1], cur[2], c, 0), group(cur[0], cur[1], cur[2], c, 1),
                               group(cur[0], cur[1], cur[2], c, 2), cv0);
      const f4 v1 = unpack24cf(group(cur[3], cur[4], cur[5], c, 0), group(cur[3], cur[4], cur[5], c, 1),
                               group(cur[3], cur[4], cur[5], c, 2), cv1);
      const f2 ppa = {pa[c], pa[c]}, ppb = {pb_[c], pb_[c]};
      o0a = __builtin_elementwise_fma(ppa, f2{v0.x, v0.y}, o0a);
      o0b = __builtin_elementwise_fma(ppa, f2{v0.z, v0.w}, o0b);
      o1a = __builtin_elementwise_fma(ppb, f2{v1.x, v1.y}, o1a);
      o1b = __builtin_elementwise_fma(ppb, f2{v1.z, v1.w}, o1b);
    }
    // pin this group's sums here (see attention_packed32)
    asm volatile("" : "+v"(o0a), "+v"(o0b), "+v"(o1a), "+v"(o1b));
    if (g + 2 < 8) load_v(vq[g % 2], g + 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  const float a0 = __builtin_fmaf(o0a.x, uv256, pv0.x * P0), a1 = __builtin_fmaf(o0a.y, uv256, pv0.y * P0);
  const float a2 = __builtin_fmaf(o0b.x, uv256, pv0.z * P0), a3 = __builtin_fmaf(o0b.y, uv256, pv0.w * P0);
  const float b0 = __builtin_fmaf(o1a.x, uv256, pv1.x * P1), b1 = __builtin_fmaf(o1a.y, uv256, pv1.y * P1);
  const float b2 = __builtin_fmaf(o1b.x, uv256, pv1.z * P1), b3 = __builtin_fmaf(o1b.y, uv256, pv1.w * P1);
  *(SLIMT_LDS int *)(r.arow + 4 * lane) =
      pack4(quantize1_byte(a0, r.aq_o), quantize1_byte(a1, r.aq_o), quantize1_byte(a2, r.aq_o), quantize1_byte(a3, r.aq_o));
  *(SLIMT_LDS int *)(r.arow + D / 2 + 4 * lane) =
      pack4(quantize1_byte(b0, r.aq_o), quantize1_byte(b1, r.aq_o), quantize1_byte(b2, r.aq_o), quantize1_byte(b3, r.aq_o));
}

// ---- one entry point per SHAPE: 0 = sentences of up to 32 tokens, 1 = 33..64, 2 = 65..128 (D = 256), 3 = D = 512 (up to 32) --
// c0 / c1: the layer's constants in LDS -- the K / V projections' prepared biases [D] (SHAPE 3: c0 = [K pb | K c127 | V pb |
// V c127][D], c1 unused).
template <int SHAPE, int KV_AUX, class F, int PART = 0, int SPW = 16>
__device__ __forceinline__ void attention_packed(AttnRow r, int lane, lcf_ptr c0, lcf_ptr c1, const F f) {
  static_assert(PART == 0 || SHAPE == 1 || SHAPE == 2, "shared heads: the 33..64- and the 65..128-token reader");
  if constexpr (SHAPE == 0)
    attention_packed32<KV_AUX>(r, lane, c0, c1, f);
  else if constexpr (SHAPE == 1)
    attention_packed64<KV_AUX, F, PART, SPW>(r, lane, c0, c1, f);
  else if constexpr (SHAPE == 2)
    attention_packed128<KV_AUX, F, PART, SPW>(r, lane, c0, c1, f);
  else if constexpr (F::BITS == 24)
    attention_row24_64<KV_AUX>(r, lane, c0, f.uk, f.uv);
  else
    attention_packed32_d64<KV_AUX>(r, lane, c0, f);
}

// The same out of line: where a narrower form is the expected one, the wider is the fallback of the rare sentence-layer that
// does not fit -- as a second inlined copy of the attention it costs every sentence registers (the kernel's allocation is
// the maximum over both paths); as a call it costs the rare one a few saved registers. Function arguments arrive in
// VGPRs: the wave-uniform ones are declared so again (uniform_ptr), else every load becomes a waterfall loop.
template <int SHAPE, int KV_AUX, class F>
__device__ __noinline__ void attention_packed_cold(AttnRow r, int lane, lcf_ptr c0, lcf_ptr c1, const F f) {
  r.kl = (gcf_ptr)uniform_ptr((const float *)r.kl);
  r.vl = (gcf_ptr)uniform_ptr((const float *)r.vl);
  if constexpr (SHAPE != 2) {  // (the 65..128-token body does it itself)
    r.S = __builtin_amdgcn_readfirstlane(r.S);
    r.len = __builtin_amdgcn_readfirstlane(r.len);
  }
  attention_packed<SHAPE, KV_AUX>(r, lane, c0, c1, f);
}
