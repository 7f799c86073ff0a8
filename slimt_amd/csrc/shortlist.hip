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
#include "shortlist_device.h"

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
  shortlist_mark(a, lb, sb, lane, (blockIdx.x * 1024 + tid) >> 6, (gridDim.x * 1024) >> 6);
  __syncthreads();
  for (int i = tid; i < TW; i += 1024)
    if (lb[i]) atomicOr(&tb[i], lb[i]);
}

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
  shortlist_compact(a, tb, scan, tid);
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
