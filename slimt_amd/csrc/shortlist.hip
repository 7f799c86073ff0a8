// Lexical shortlist generation on the device: ShortlistGenerator::generate
// (slimt/Shortlist.cc:115-175), called once per batch by Model::forward
// (slimt/Model.cc:117-120) on the batch's source words (Input::words(),
// slimt/Input.cc:24 -- the tokens before each row's padding).
//
// The reference walks two O(V) truth tables on the host. Here they are bitmaps
// (V = 32000 -> 2 x 4 KB, L2-resident scratch owned by the caller's context):
// pass 1 lets every source token OR in its aligned target ids (first
// occurrence of a source word only, like the reference -- later ones would
// set the same bits); pass 2, one workgroup, adds the
// `frequent` ids, counts, applies the multiple-of-eight patch and emits the
// set bits in ascending order through a prefix sum of per-word popcounts.
// Integer work only: the result is the reference's id list, bit for bit.
#include "device_common.h"
#include "kernels.h"

namespace slimt_hip {

// Pass 1, a few large workgroups with a private LDS bitmap each. A wave takes 64
// tokens at a time, one per lane: the lane claims its source word (first
// occurrence only) and fetches the word's list bounds -- three memory round
// trips for 64 tokens. The wave then walks the claimed tokens, all lanes ORing
// 64 list entries per step into LDS, the next token's entries already in flight.
// At the end the non-zero words of the LDS bitmap are ORed into the global one.
__global__ __launch_bounds__(1024) void shortlist_mark_kernel(ShortlistArgs a) {
  extern __shared__ uint32_t sl_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int TW = (a.tgt_vocab + 31) / 32;
  uint32_t *lb = sl_smem;
  uint32_t *tb = a.scratch, *sb = a.scratch + TW;
  for (int i = tid; i < TW; i += 1024) lb[i] = 0;
  __syncthreads();
  const int wave = (blockIdx.x * 1024 + tid) >> 6, n_waves = (gridDim.x * 1024) >> 6;
  const int n_tok = a.B * a.S;
  const uint32_t kNone = 0xffffffffu;
  for (int base = wave * 64; base < n_tok; base += n_waves * 64) {
    const int idx = base + lane;
    unsigned long long begin = 0, end = 0;
    if (idx < n_tok) {
      const int b = idx / a.S, j = idx - b * a.S;
      if (j < (int)a.lengths[b]) {  // padding is not a word (Input::words())
        const uint32_t w = a.ids[idx];
        if (w < (uint32_t)a.src_vocab) {  // out of range: undefined in the reference; ignored
          if (a.shared && w < (uint32_t)a.tgt_vocab) atomicOr(&lb[w >> 5], 1u << (w & 31));
          const uint32_t bit = 1u << (w & 31);
          if (!(atomicOr(&sb[w >> 5], bit) & bit)) {
            begin = a.w2o[w];
            end = a.w2o[w + 1];
          }
        }
      }
    }
    unsigned long long todo = __ballot(end > begin);
    auto fetch = [&](int l) -> uint32_t {  // first 64 entries of lane l's list
      const unsigned long long bl = __shfl(begin, l, 64), el = __shfl(end, l, 64);
      return bl + lane < el ? a.lists[bl + lane] : kNone;
    };
    int cur_l = -1;
    uint32_t cur = kNone;
    if (todo) {
      cur_l = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      cur = fetch(cur_l);
    }
    while (cur_l >= 0) {
      int nxt_l = -1;
      uint32_t nxt = kNone;
      if (todo) {
        nxt_l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        nxt = fetch(nxt_l);
      }
      if (cur != kNone) atomicOr(&lb[cur >> 5], 1u << (cur & 31));  // < tgt_vocab: checked at load
      const unsigned long long bl = __shfl(begin, cur_l, 64), el = __shfl(end, cur_l, 64);
      for (unsigned long long k = bl + 64 + lane; k < el; k += 64) {  // lists longer than 64
        const uint32_t t = a.lists[k];
        atomicOr(&lb[t >> 5], 1u << (t & 31));
      }
      cur_l = nxt_l;
      cur = nxt;
    }
  }
  __syncthreads();
  for (int i = tid; i < TW; i += 1024)
    if (lb[i]) atomicOr(&tb[i], lb[i]);
}

// Pass 2, one workgroup: frequent ids, multiple-of-eight patch, ordered emission;
// leaves the scratch bitmaps zeroed for the next call.
__global__ __launch_bounds__(1024) void shortlist_compact_kernel(ShortlistArgs a) {
  extern __shared__ uint32_t sl_smem[];
  const int tid = threadIdx.x;
  const int TW = (a.tgt_vocab + 31) / 32, SW = (a.src_vocab + 31) / 32;
  uint32_t *tb = sl_smem;    // target truth table
  uint32_t *scan = tb + TW;  // [1024] per-thread counts / offsets
  for (int i = tid; i < TW; i += 1024) {
    tb[i] = a.scratch[i];
    a.scratch[i] = 0;
  }
  for (int i = tid; i < SW; i += 1024) a.scratch[TW + i] = 0;
  __syncthreads();
  // most frequent words (Shortlist.cc:125-127)
  const unsigned long long nf = a.frequent < (unsigned long long)a.tgt_vocab
                                    ? a.frequent : (unsigned long long)a.tgt_vocab;
  for (int i = tid; i < (int)nf; i += 1024) atomicOr(&tb[i >> 5], 1u << (i & 31));
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
  {  // total over the workgroup: per-wave shuffles, then 16 partial sums
    uint32_t t = cnt;
    for (int x = 32; x >= 1; x >>= 1) t += __shfl_xor(t, x, 64);
    if ((tid & 63) == 0) scan[tid >> 6] = t;
  }
  __syncthreads();
  if (tid < 64) {
    // multiple-of-eight patch (Shortlist.cc:148-165): the lowest unset ids >=
    // frequent, found by one wave 64 bitmap words at a time (the reference's
    // id-by-id scan is O(V) when the table is nearly full)
    const int lane = tid;
    uint32_t ones = 0;
    for (int i = 0; i < 16; ++i) ones += scan[i];
    uint32_t need = (8u - ones % 8u) % 8u;
    const unsigned long long f = a.frequent;
    const int fw = f < (unsigned long long)a.tgt_vocab ? (int)(f >> 5) : TW;
    for (int base = fw; base < TW && need > 0; base += 64) {
      const int w = base + lane;
      uint32_t z = 0;
      if (w < TW) {
        z = ~tb[w] & valid_mask(w);
        if (w == fw) z &= ~((1u << (f & 31)) - 1u);
      }
      const uint32_t c = (uint32_t)__popc(z);
      uint32_t inc = c;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(inc, d, 64);
        if (lane >= d) inc += v;
      }
      const uint32_t exc = inc - c;
      uint32_t take = exc < need ? (c < need - exc ? c : need - exc) : 0u;
      uint32_t add = 0;
      for (; take > 0; --take) {
        const uint32_t low = z & (0u - z);
        add |= low;
        z ^= low;
      }
      if (add) tb[w] |= add;
      const uint32_t total = __shfl(inc, 63, 64);
      const uint32_t used = total < need ? total : need;
      ones += used;
      need -= used;
    }
    if (lane == 0) {
      *a.n_out = ones;
      if (a.n_out_host) *a.n_out_host = ones;
    }
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

size_t shortlist_scratch_bytes(int src_vocab, int tgt_vocab) {
  return ((size_t)(tgt_vocab + 31) / 32 + (size_t)(src_vocab + 31) / 32) * sizeof(uint32_t);
}

size_t shortlist_lds_bytes(int src_vocab, int tgt_vocab) {
  (void)src_vocab;
  return ((size_t)(tgt_vocab + 31) / 32 + 1024) * sizeof(uint32_t);
}

hipError_t launch_shortlist_generate(const ShortlistArgs &a, hipStream_t st) {
  const size_t lds = shortlist_lds_bytes(a.src_vocab, a.tgt_vocab);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (!a.scratch) return hipErrorInvalidValue;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void *>(shortlist_compact_kernel), (int)lds);
  if (e != hipSuccess) return e;
  const int n_tok = a.B * a.S;
  int blocks = (n_tok + 1023) / 1024;  // 64 tokens per wave pass, 16 waves per block
  blocks = blocks < 1 ? 1 : (blocks > 16 ? 16 : blocks);
  const size_t lds1 = (size_t)((a.tgt_vocab + 31) / 32) * sizeof(uint32_t);
  e = set_dynamic_lds_once(reinterpret_cast<const void *>(shortlist_mark_kernel), (int)lds1);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(shortlist_mark_kernel, dim3(blocks), dim3(1024), lds1, st, a);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(shortlist_compact_kernel, dim3(1), dim3(1024), lds, st, a);
  return hipGetLastError();
}

}  // namespace slimt_hip
