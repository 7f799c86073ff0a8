// Lexical shortlist generation on the device: ShortlistGenerator::generate
// (slimt/Shortlist.cc:115-175), called once per batch by Model::forward
// (slimt/Model.cc:117-120) on the batch's source words (Input::words(),
// slimt/Input.cc:24 -- the tokens before each row's padding).
//
// The reference walks two O(V) truth tables on the host. Here one workgroup
// keeps both as bitmaps in LDS (V = 32000 -> 2 x 4 KB): mark the `frequent`
// ids, let every source token OR in its aligned target ids (first occurrence
// of a source word only, like the reference -- later ones would set the same
// bits), count, apply the multiple-of-eight patch, then emit the set bits in
// ascending order through a prefix sum of per-word popcounts. Integer work
// only: the result is the reference's id list, bit for bit.
#include "device_common.h"
#include "kernels.h"

namespace slimt_hip {

__global__ __launch_bounds__(1024) void shortlist_generate_kernel(ShortlistArgs a) {
  extern __shared__ uint32_t sl_smem[];
  const int tid = threadIdx.x;
  const int TW = (a.tgt_vocab + 31) / 32, SW = (a.src_vocab + 31) / 32;
  uint32_t *tb = sl_smem;     // target truth table
  uint32_t *sb = tb + TW;     // source truth table
  uint32_t *scan = sb + SW;   // [1024] per-thread counts / offsets
  for (int i = tid; i < TW + SW; i += 1024) tb[i] = 0;
  __syncthreads();
  // most frequent words (Shortlist.cc:125-127)
  const unsigned long long nf = a.frequent < (unsigned long long)a.tgt_vocab
                                    ? a.frequent : (unsigned long long)a.tgt_vocab;
  for (int i = tid; i < (int)nf; i += 1024) atomicOr(&tb[i >> 5], 1u << (i & 31));
  // source words -> aligned target words (Shortlist.cc:131-145)
  const int n_tok = a.B * a.S;
  for (int idx = tid; idx < n_tok; idx += 1024) {
    const int b = idx / a.S, j = idx - b * a.S;
    if (j >= (int)a.lengths[b]) continue;
    const uint32_t w = a.ids[idx];
    if (w >= (uint32_t)a.src_vocab) continue;  // undefined in the reference; ignored here
    if (a.shared && w < (uint32_t)a.tgt_vocab) atomicOr(&tb[w >> 5], 1u << (w & 31));
    const uint32_t bit = 1u << (w & 31);
    const uint32_t old = atomicOr(&sb[w >> 5], bit);
    if (!(old & bit)) {
      const unsigned long long begin = a.w2o[w], end = a.w2o[w + 1];
      for (unsigned long long k = begin; k < end; ++k) {
        const uint32_t t = a.lists[k];
        atomicOr(&tb[t >> 5], 1u << (t & 31));  // t < tgt_vocab: checked at load
      }
    }
  }
  __syncthreads();
  // contiguous words per thread, so that offsets follow the id order
  const int wpt = (TW + 1023) / 1024;
  const int w0 = tid * wpt, w1 = (w0 + wpt) < TW ? (w0 + wpt) : TW;
  auto valid_mask = [&](int w) -> uint32_t {  // bits of word w that are real vocabulary ids
    const int rem = a.tgt_vocab - 32 * w;
    return rem >= 32 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
  };
  uint32_t cnt = 0;
  for (int w = w0; w < w1; ++w) cnt += __popc(tb[w] & valid_mask(w));
  scan[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    // multiple-of-eight patch (Shortlist.cc:148-165): lowest unset ids >= frequent
    uint32_t ones = 0;
    for (int i = 0; i < 1024; ++i) ones += scan[i];
    for (unsigned long long i = a.frequent; i < (unsigned long long)a.tgt_vocab && (ones % 8u) != 0u; ++i) {
      const uint32_t bit = 1u << (i & 31);
      if (!(tb[i >> 5] & bit)) {
        tb[i >> 5] |= bit;
        ones++;
      }
    }
    *a.n_out = ones;
  }
  __syncthreads();
  cnt = 0;
  for (int w = w0; w < w1; ++w) cnt += __popc(tb[w] & valid_mask(w));
  __syncthreads();
  scan[tid] = cnt;
  __syncthreads();
  // inclusive Hillis-Steele scan over the 1024 per-thread counts
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? scan[tid - d] : 0u;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  uint32_t off = scan[tid] - cnt;  // exclusive
  // bucket sort (Shortlist.cc:168-173)
  for (int w = w0; w < w1; ++w) {
    uint32_t bits = tb[w] & valid_mask(w);
    while (bits) {
      const int bpos = __ffs((int)bits) - 1;
      a.out[off++] = (uint32_t)(32 * w + bpos);
      bits &= bits - 1;
    }
  }
}

size_t shortlist_lds_bytes(int src_vocab, int tgt_vocab) {
  return ((size_t)(tgt_vocab + 31) / 32 + (size_t)(src_vocab + 31) / 32 + 1024) * sizeof(uint32_t);
}

hipError_t launch_shortlist_generate(const ShortlistArgs &a, hipStream_t st) {
  const size_t lds = shortlist_lds_bytes(a.src_vocab, a.tgt_vocab);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(shortlist_generate_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(shortlist_generate_kernel, dim3(1), dim3(1024), lds, st, a);
  return hipGetLastError();
}

}  // namespace slimt_hip
