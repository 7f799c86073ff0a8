// Device-resident model + per-worker context for the MI355X slimt backend.
// Mirrors slimt::Transformer / Encoder / Decoder (slimt/Transformer.hh:15-72)
// and the loop of slimt::Model::forward/decode (slimt/Model.cc:111-204).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <chrono>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/slimt_hip.h"
#include "kernels.h"

namespace slimt_hip {

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t n);  // grows (never shrinks); contents undefined after growth
  void release();
  template <class T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

struct AffineW {  // slimt::Affine / slimt::Linear (Modules.hh:14-22), prepared
  PreparedWeight w;
  DevBuf Wp, colsum, pb;
};

struct LnW {
  DevBuf scale, bias;
};

struct AttnW {
  AffineW q, k, v, o;
  LnW ln;
};

struct EncLayerW {
  AttnW attn;
  AffineW ffn1, ffn2;
  LnW ffn_ln;
};

struct DecLayerW {
  AffineW rnn_f, rnn_w;
  LnW rnn_ln;
  AttnW attn;
  AffineW ffn1, ffn2;
  LnW ffn_ln;
};

}  // namespace slimt_hip

struct slimt_hip_ctx;

struct slimt_hip_model {
  int device = 0;
  int D = 0, F = 0, H = 0, V = 0, Le = 0, Ld = 0;
  std::vector<slimt_hip::EncLayerW> enc;
  std::vector<slimt_hip::DecLayerW> dec;
  slimt_hip::DevBuf wemb;      // int8 [V][D] (embedding lookup, Io.cc:191-200)
  float wemb_mult = 0.f;       // quantisation multiplier of Wemb
  slimt_hip::DevBuf out_raw;   // Wemb_intgemm8: int8 [V][D] (Io.cc:206-224)
  slimt_hip::DevBuf out_bias;  // decoder_ff_logit_out_b [V]
  float out_a_quant = 0.f;     // none_QuantMultA (Transformer.cc:111-113)
  slimt_hip::AffineW out_full; // full-vocabulary output layer
  // Admission of persistent decoders (all contexts of this model): at most
  // `decoder_budget` decoder workgroups are meant to run at a time; launch k waits
  // (on its stream, not the host) for launch k - n, n = budget / its workgroups.
  // The other CUs stay with the encoders of the batches behind. 0 = no limit.
  std::mutex submit_mu;  // held while a persistent translate is queued (translate_device)
  std::mutex gate_mu;
  std::vector<hipEvent_t> gate_ev;  // ring, created on first use
  size_t gate_seq = 0;
  int decoder_budget = 0;
  // per context: its latest admitted decoder launch and that batch's K/V bytes (choice of
  // the K/V cache policy: how much K/V do the contexts with a pending decoder hold together)
  struct GateCtx {
    const slimt_hip_ctx *ctx;
    size_t seq;
    double kv_bytes;
    std::chrono::steady_clock::time_point when;  // host time of that launch call
  };
  std::vector<GateCtx> gate_ctx;
  // XCD-affine placement of a batch's decoder workgroups (kernels.h, FusedDecodeArgs::home_mask):
  // 0 = off, 1 / 2 = a batch's tiles are claimed on one / two XCDs. Launch k takes the home of launch
  // k - n (the one whose end admits it: its CUs are the ones that have just come free).
  int xcd_affinity = 0;
  std::vector<unsigned> gate_home;  // ring, parallel to gate_ev: home mask of each admitted launch
  int kv_policy = 0;  // 0 = chosen per launch, 1 = always temporal, 2 = always non-temporal K/V loads
  std::atomic<unsigned long long> kv_call_seq{0};  // K/V cache policy per launch (engine.cpp): a slot per translate call,
  std::atomic<int> kv_k_last{8};                   // and the k (kept launches of every 8) of the last admission
  bool adaptive_rows = true;  // decode mode 0: 8 or 4 sentences per decoder workgroup while CUs would idle (engine.cpp)
  // 0 = the packed cache where the kernels have it (kernels.h, kv24), 20 bits per value for the sentence-layers whose
  // accumulators fit (kv_fmt) and 24 for the others; 1 = always f32; 2 = packed, always 24 bits
  int kv_format = 0;
  int kv_narrow_limit = 1 << 19;  // accumulators in [-limit, limit) take the narrow form (slimt_hip_debug_kv_narrow_limit)
  // Format 0 watches how its sentences fare: the encoders add 1 to *kv_wide_count (pinned host memory, allocated on first
  // use) per sentence-layer that had to take the 24-bit form, the engine counts the sentence-layers it submitted. A
  // sentence in the 24-bit form is read through an out-of-line call that makes its whole workgroup wait (every sentence
  // through it: 28.8 against 34.5 M tok/s with the 24-bit form inlined, base 10.7 against 12.0;
  // profiles/r05_kv_fallback_call_vs_inlined.txt), so a model whose accumulators mostly do not fit 20 bits is switched,
  // once, to the 24-bit form for every batch (kv_auto_wide): the kernels round 4 ran.
  unsigned long long *kv_wide_count = nullptr;
  std::atomic<unsigned long long> kv_layers_submitted{0};
  std::atomic<bool> kv_auto_wide{false};
  // The tight (16-bit) form below the narrow one (kernels.h, FusedDecodeArgs::kv_tight), per decoder layer: a sentence that
  // tries it and does not fit costs its encoder workgroup a second K/V pass and its decoder wave the slower reader, so a
  // layer whose sentences mostly do not fit stops trying (bit l of kv_tight_off; counters kv_wide_count[1 + l] on the
  // device side, kv_tight_submitted[l] sentences on this one). kv_tight_limit: 2^15 (int16); tests lower it, 0 = never.
  int kv_tight_limit = 1 << 15;
  std::atomic<unsigned long long> kv_tight_submitted[4] = {};
  std::atomic<unsigned> kv_tight_off{0};
  // The tight form's per-column centres [Ld][K, V][D] (kernels.h, FusedDecodeArgs::kv_centre): the column means of the
  // first batch of at least 1024 rows that could have taken the form -- that batch is cached as f32 (its decoder reads
  // that form), a reduction behind its encoder writes the centres, and the form is tried from the first batch submitted
  // after the reduction's event has completed. Written once (or set by the caller before any batch), never changed after:
  // a batch's encoder and decoder read the same numbers.
  // Round 6: centres come in GENERATIONS. A layer whose sentences mostly miss the form under the current centres (the watch
  // above) used to stop trying for good -- but the first batch of >= 1024 rows need not look like the traffic behind it.
  // Now the first kv_recal_max trips start a new generation instead: the next suitable batch is cached as f32 and
  // calibrates fresh centres into ITS OWN buffer (batches in flight keep reading the generation they were encoded with:
  // a context remembers it, slimt_hip_ctx::kv_gen), the form is tried again with them, and only a generation past the
  // limit switches a layer off. The miss counters are per generation (kv_wide_count[1 + 4 gen + layer]): stale batches
  // of the generation before cannot trip the new one.
  static constexpr int kKvGens = 4;
  slimt_hip::DevBuf kv_centre[kKvGens], kv_centre_sums;
  hipEvent_t kv_centre_ev = nullptr;
  std::atomic<int> kv_centre_state{0};  // of the CURRENT generation: 0 = none, 1 = a calibration batch is in flight, 2 = ready
  std::atomic<bool> kv_centre_claimed{false};
  std::atomic<int> kv_gen{0};           // the current generation
  int kv_recal_max = 2;                 // re-calibrations allowed (at most kKvGens - 1)
  const int *kv_centre_of(int gen, int layer, int kv) const {
    return kv_centre[gen].as<int>() + (size_t)(2 * layer + kv) * (size_t)D;
  }
};

struct slimt_hip_ctx {
  slimt_hip_model *model = nullptr;
  hipStream_t stream = nullptr;
  slimt_hip::DevBuf gen_flag;         // in-launch shortlist generation: the word its publisher sets to gen_epoch (kernels.h, FusedEncodeArgs::gen)
  unsigned gen_epoch = 0;
  // slimt_hip_debug_break_shortlist_handoff: the waiters of the next launches poll a word the publisher never sets, `limit` times
  bool gen_break = false;
  unsigned gen_spin_limit = 1u << 24;
  hipEvent_t sync_event = nullptr;  // blocking-sync event of slimt_hip_ctx_synchronize (created on first use)
  std::vector<uint32_t> sl_host;    // the shortlist last uploaded by translate_host (re-uploaded only when it changes)
  bool own_stream = false;
  size_t max_B = 0, max_S = 0;
  size_t max_M = 0;  // padded tokens (B * S) the workspace holds
  // current batch
  int B = 0, S = 0;
  int n_sl = 0;  // 0 => full vocabulary
  bool have_encoder_out = false;
  bool decode_ready = false;
  bool kv_ready = false;  // cross-attention K/V already produced by the fused encoder
  slimt_hip::DevBuf dbg_embed, dbg_layers;
  int encode_rows = 0;  // rows per workgroup of the persistent D = 256 encoder: 0 auto, 32, 64
  int decode_mode = 0;  // 0 auto (fused when supported), 1 step-wise launches, 2 / 3 fused with 16 / 32 rows per workgroup
  slimt_hip::DevBuf stamps;  // diagnostic phase stamps of the fused decoder
  int stamp_step = -1;
  // encoder workspace
  slimt_hip::DevBuf pos;  // [max_S][D]
  slimt_hip::DevBuf ids, lengths;
  slimt_hip::DevBuf x0, x1, q, k, v, att, h8, a8;
  slimt_hip::DevBuf ticket;      // ticket counter of the over-subscribed decoder launches
  unsigned ticket_base = 0;      // tickets handed out by earlier launches
  unsigned enc_ticket_base = 0;  // the same for the fused encoder (second counter of `ticket`)
  unsigned xarr_base = 0, xclaim_base = 0;  // XCD-affine claims (64-bit state word behind the two counters)
  slimt_hip::DevBuf kv;  // [Ld][2][B*S][D]
  slimt_hip::DevBuf cl_act, cl_part, cl_sync;  // cluster logits (kernels.h, FusedDecodeArgs::cluster): the members' hand-over buffers
  slimt_hip::DevBuf kv_fmt;  // [Ld][B] bytes: the form of each sentence-layer's packed cache (kernels.h, FusedDecodeArgs::kv_fmt)
  bool kv_fmt_valid = false;  // the encoder of the current batch recorded kv_fmt (else every cache is in the 24-bit form)
  int kv_fmt_B = 0;           // the batch size kv_fmt was recorded for
  bool kv_tight = false;      // ... and some of its sentence-layers may be in the tight form: the decoder with its reader
  int kv_gen = 0;             // ... relative to the centres of this generation (engine.h, slimt_hip_model::kv_gen)
  bool expect_large_output = false;  // the last decoder launch of this context had an output layer of > 16384 columns (mode 0: the 32-sentence tiling, no tight reader)
  // decoder workspace
  slimt_hip::DevBuf dx, dx_pre, dh, datt8, dout, df8, state;
  slimt_hip::DevBuf part_val, part_idx;
  slimt_hip::DevBuf prev, out_ids, out_len, finished, n_finished, align;
  slimt_hip::DevBuf shortlist;
  slimt_hip::DevBuf sl_scratch;  // bitmaps of slimt_hip_shortlist_generate_device (kept zeroed)
  slimt_hip::DevBuf n_sl_dev;    // [1] size of a shortlist generated on this context's stream
  slimt_hip::AffineW out_sl;  // shortlisted output layer (per batch)
  slimt_hip::DevBuf logits, attn_dbg;
  int *n_finished_host = nullptr;  // pinned
  // profiling
  int prof_kernel = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_macs = 0, prof_bytes = 0;
};

// ShortlistGenerator (slimt/Shortlist.hh:38-90): the binary shortlist on the device
struct slimt_hip_shortlist {
  int device = 0;
  uint64_t frequent = 0, best = 0;
  size_t source_vocab = 0, target_vocab = 0;
  bool shared = false;
  slimt_hip::DevBuf w2o, lists;                // word_to_offset (uint64), shortlist (uint32)
  slimt_hip::DevBuf ids, lengths, out, n_out;  // staging of the host entry point
  slimt_hip::DevBuf scratch;                   // ... and its bitmaps
  std::mutex mu;  // the host entry point stages through the buffers above: one caller at a time
};

