// Device-side pieces of the lexical shortlist generator (shortlist.hip): shared by its two kernels and by
// the 64-row encoder, whose first workgroup generates the batch's shortlist inside the encoder launch
// (encode_tall.hip) -- a one-workgroup kernel of its own waited about 0.5 ms for a free CU behind the
// persistent kernels of the other batches.
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace slimt_hip {

// Pass 1 by `n_waves` waves (this one is `wave`): every source token ORs its aligned target ids into the
// target bitmap `lb` (LDS), first occurrence of a source word only (claimed in the source bitmap `sb`, LDS
// or global). A wave takes 64 tokens at a time, one per lane: the lane claims its source word and fetches
// the word's list bounds -- three memory round trips for 64 tokens. The wave then walks the claimed tokens,
// all lanes ORing 64 list entries per step into LDS, the next token's entries already in flight.
__device__ __forceinline__ void shortlist_mark(const ShortlistArgs &a, uint32_t *lb, uint32_t *sb, int lane, int wave,
                                               int n_waves) {
  const int n_tok = a.B * a.S;
  const uint32_t kNone = 0xffffffffu;
  for (int base = wave * 64; base < n_tok; base += n_waves * 64) {
    const int idx = base + lane;
    unsigned long long begin = 0, end = 0;
    if (idx < n_tok) {
      const int b = idx / a.S, j = idx - b * a.S;
      if ((uint32_t)j < a.lengths[b]) {  // padding is not a word (Input::words()); a length past S is S
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
}

// Pass 2 by one workgroup of 1024 threads over the complete target bitmap `tb` (LDS, TW words) with `scan`
// (LDS, 1024 words): adds the `frequent` ids, counts, applies the multiple-of-eight patch and emits the set
// bits in ascending order through a prefix sum of per-word popcounts. The caller has synchronised before.
__device__ __forceinline__ void shortlist_compact(const ShortlistArgs &a, uint32_t *tb, uint32_t *scan, int tid) {
  const int TW = (a.tgt_vocab + 31) / 32;
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

// Both passes by ONE workgroup of 1024 threads, bitmaps in LDS (lds: TW + SW + 1024 words); no global scratch.
__device__ __forceinline__ void shortlist_generate_block(const ShortlistArgs &a, uint32_t *lds, int tid) {
  const int TW = (a.tgt_vocab + 31) / 32, SW = (a.src_vocab + 31) / 32;
  uint32_t *tb = lds, *sb = tb + TW, *scan = sb + SW;
  for (int i = tid; i < TW + SW; i += 1024) tb[i] = 0;
  __syncthreads();
  shortlist_mark(a, tb, sb, tid & 63, tid >> 6, 16);
  __syncthreads();
  shortlist_compact(a, tb, scan, tid);
}

// ---- ShortlistGenerator::generate inside an encoder launch (kernels.h, FusedEncodeArgs::gen) ------------------
// The workgroup that claims tile 0 -- the first to START: tiles are claimed by ticket, so it is running before any
// waiter exists, and it waits for nobody -- generates the batch's shortlist in the LDS the encoder is not using yet
// and publishes it: plain stores, every storing wave drained, the workgroup's barrier, ONE agent-scope release, the
// flag. `lds`: shortlist_in_launch_lds_bytes() of dynamic LDS. Ends with a barrier: the LDS is the caller's again.
__device__ __forceinline__ void shortlist_publish_in_launch(const ShortlistArgs &a, uint32_t *lds, unsigned *flag,
                                                            unsigned epoch, int tid) {
  shortlist_generate_block(a, lds, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

// Every workgroup, at the end of its encoder work and before it packs its share of the shortlisted output layer:
// one relaxed poll loop by one lane, ONE agent-scope acquire, the barrier; plain loads of the ids / the count behind it.
// The poll is bounded like every spin (`limit` polls, ~2 s by default; unreachable while the launch is ticketed: the
// publisher started first and waits for nobody). When it does run out the ids and the count were never published:
// returns false -- the caller must NOT pack from them -- and sets *error (pinned host memory, nullable) so that the
// host fails the batch instead of returning translations made from a stale or partial shortlist.
__device__ __forceinline__ bool shortlist_await_in_launch(unsigned *flag, unsigned epoch, int tid, unsigned *error,
                                                          unsigned limit = 1u << 24) {
  __shared__ int published;
  if (tid == 0) {
    unsigned spin = 0;
    bool ok = true;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      if (++spin >= limit) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok && error) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    published = ok ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return published != 0;
}

// ---- a merged launch (kernels.h, MergeIn): every sub-batch has ITS shortlist (Model.cc:117-120) -----------------------
// Sub-batch j's generator arguments: the launch's (tables, vocabularies, `frequent`) with j's own sentences; its ids go
// to out + j * tgt_vocab, its count to n_out[j], its flag is flag[j].
__device__ __forceinline__ ShortlistArgs shortlist_args_of(const FusedEncodeArgs &a, int j) {
  ShortlistArgs g = a.gen;
  if (a.n_sub == 0) return g;
  g.ids = a.sub[j].ids;
  g.lengths = a.sub[j].lengths;
  g.B = a.sub[j].n;
  g.S = a.sub[j].S;
  g.out = a.gen.out + (size_t)j * a.gen.tgt_vocab;
  g.n_out = a.gen.n_out + j;
  g.n_out_host = j == 0 ? a.gen.n_out_host : nullptr;
  return g;
}
// The publishers: the workgroups that claimed tiles 0 .. n - 1 -- the first n to START, each running before any waiter
// that could wait for it exists and waiting for nobody while it generates (tile t also takes t + n_wg, ... when a launch
// has fewer tiles than shortlists: sentences of one or two tokens).
__device__ __forceinline__ void shortlists_publish_in_launch(const FusedEncodeArgs &a, uint32_t *lds, int tile, int n_wg, int tid) {
  const int n = a.n_sub ? a.n_sub : 1;
  for (int j = tile; j < n; j += n_wg)
    shortlist_publish_in_launch(shortlist_args_of(a, j), lds, a.gen_flag + j, a.gen_epoch, tid);
}
// ... and the waiters: one lane polls the flags one after the other, ONE acquire behind the last (see shortlist_await_in_launch)
__device__ __forceinline__ bool shortlists_await_in_launch(const FusedEncodeArgs &a, int tid) {
  const int n = a.n_sub ? a.n_sub : 1;
  if (n == 1) return shortlist_await_in_launch(a.gen_flag, a.gen_epoch ^ a.gen_wait_xor, tid, a.dev_error, a.gen_spin_limit);
  __shared__ int published_all;
  if (tid == 0) {
    bool ok = true;
    const unsigned epoch = a.gen_epoch ^ a.gen_wait_xor;
    for (int j = 0; j < n && ok; ++j) {
      unsigned spin = 0;
      while (__hip_atomic_load(a.gen_flag + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        if (++spin >= a.gen_spin_limit) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (!ok && a.dev_error) __hip_atomic_store(a.dev_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    published_all = ok ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return published_all != 0;
}

}  // namespace slimt_hip
