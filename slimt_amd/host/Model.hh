// Host-side mirror of the reference's model driver for the HIP backend:
// slimt::Input (slimt/Input.hh:10-36), slimt::Model::Config / forward
// (slimt/Model.hh:33-56) and the Histories result types (slimt/Types.hh:34-62),
// implemented on top of the C ABI (include/slimt_hip.h). One `Model` owns the
// device weights; each worker thread owns one `Worker` (stream + workspace),
// like one slimt::Async worker calling the const Model::forward
// (slimt/Frontend.cc:212-226).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <vector>

#include "slimt_hip.h"

namespace slimt {

using Word = uint32_t;
using Words = std::vector<Word>;
using Distribution = std::vector<float>;
using Alignment = std::vector<Distribution>;

struct Hypothesis {  // slimt/Types.hh:55-61
  Words target;
  Alignment alignment;
  // The same rows as ONE block, [target.size()][source tokens] row-major, filled INSTEAD of
  // `alignment` when the producer was asked for flat rows (ServiceConfig::flat_alignments: the C ABI,
  // whose callers want arrays -- one allocation per sentence instead of one per target token)
  std::vector<float> alignment_flat;
  size_t padded_length = 0;  // diagnostic: the S of the batch this sentence was translated in
  uint64_t batch = 0;        // diagnostic: serial number of that batch (Service: batches in launch order)
};
using History = std::shared_ptr<Hypothesis>;
// alignment_flat -> alignment: target.size() rows of equal length (the sentence's source tokens), then the block is released
void expand_alignment(Hypothesis &hypothesis);
using Histories = std::vector<History>;

// Padded batch (slimt/Input.cc:20-63): indices [B,S], lengths, limit factor.
class Input {
 public:
  Input(size_t batch_size, size_t sequence_length, uint32_t pad_id, float limit_factor)
      : batch_size_(batch_size), sequence_length_(sequence_length), pad_id_(pad_id),
        limit_factor_(limit_factor), indices_(batch_size * sequence_length, pad_id) {}
  void add(const Words &words);  // one sentence, EOS included
  const std::vector<uint32_t> &indices() const { return indices_; }
  const std::vector<uint32_t> &lengths() const { return lengths_; }
  const Words &words() const { return words_; }
  size_t batch_size() const { return batch_size_; }
  size_t sequence_length() const { return sequence_length_; }
  float limit_factor() const { return limit_factor_; }

 private:
  size_t batch_size_, sequence_length_;
  uint32_t pad_id_;
  float limit_factor_;
  std::vector<uint32_t> indices_;
  std::vector<uint32_t> lengths_;
  Words words_;
};

// Device contexts that threads keep for a Model between calls (the class-level mirror: Encoder::forward / Decoder::step
// have no "end of sequence" call, so the workspace of a sequence lives with the thread: host/Transformer.cc). A context
// points into the model, so the model and those threads share this registry: whichever goes first -- the Model or a
// thread -- destroys the context, under the registry's lock, and the other finds it gone. Keyed by the registry itself,
// never by a Model's address (a later Model at the same address is another registry).
struct ThreadContexts {
  struct Entry {
    slimt_hip_ctx *ctx = nullptr;
  };
  std::mutex mutex;
  bool model_alive = true;
  std::vector<Entry *> entries;  // every thread's entry for this model (owned by the thread)
};

class Model {
 public:
  struct Config {  // slimt/Model.hh:33-51 (tiny preset defaults, Model.cc:206-231)
    size_t encoder_layers = 6;
    size_t decoder_layers = 2;
    size_t num_heads = 8;
    uint32_t eos_id = 0;
    int device = 0;
  };
  // `model_bin`: a Marian .bin held in memory for the duration of the call. `lexical_shortlist`
  // (optional): the binary lexical shortlist of the reference's Package (make_shortlist_generator,
  // slimt/Model.cc:60-82), copied to the device; forward() then generates every batch's shortlist.
  Model(const Config &config, const void *model_bin, size_t size, const void *lexical_shortlist = nullptr,
        size_t lexical_shortlist_size = 0);
  // A view of weights that live elsewhere (created through the C ABI): not destroyed with the Model,
  // and they must outlive it (forward()'s pooled contexts are released by ~Model).
  Model(const Config &config, slimt_hip_model *borrowed) : config_(config), model_(borrowed), owned_(false) {
    int32_t heads = 0;
    if (borrowed && slimt_hip_model_info(borrowed, nullptr, nullptr, nullptr, &heads) == 0 && heads > 0)
      config_.num_heads = static_cast<size_t>(heads);
    config_.device = slimt_hip_model_device(borrowed);
  }
  ~Model();
  Model(const Model &) = delete;
  Model &operator=(const Model &) = delete;
  slimt_hip_model *handle() const { return model_; }
  const Config &config() const { return config_; }
  // slimt::Model::forward (slimt/Model.hh:56, Model.cc:187-204): one const, re-entrant call --
  // the batch's shortlist (when the model has a generator), encoder, greedy decode, a Hypothesis
  // with its alignment rows per sentence. Any number of threads may call it on one Model, like the
  // reference's Async workers (Frontend.cc:212-226): each call borrows a device context (stream +
  // workspace) from the model's pool, or builds one when every context is in use; a context that
  // is too small for the batch is replaced by one that fits.
  Histories forward(const Input &input) const;
  // contexts forward() has built so far (at most the largest number of concurrent callers)
  size_t contexts_built() const;
  // the per-thread contexts of the class-level mirror (host/Transformer.cc); destroyed with the Model at the latest
  const std::shared_ptr<ThreadContexts> &thread_contexts() const { return thread_contexts_; }

 private:
  struct Lease {  // a pooled context and the batch sizes its workspace holds
    slimt_hip_ctx *ctx = nullptr;
    size_t max_batch = 0, max_length = 0, max_tokens = 0;
  };
  Config config_;
  slimt_hip_model *model_ = nullptr;
  bool owned_ = true;
  slimt_hip_shortlist *generator_ = nullptr;
  mutable std::mutex pool_mu_;
  mutable std::vector<Lease> idle_;
  mutable size_t built_ = 0;
  std::shared_ptr<ThreadContexts> thread_contexts_ = std::make_shared<ThreadContexts>();
};

class Worker {
 public:
  // max_tokens: padded-token budget (B * S) of the workspace; 0 = max_batch * max_length
  Worker(const Model &model, size_t max_batch, size_t max_length, size_t max_tokens = 0);
  ~Worker();
  Worker(const Worker &) = delete;
  Worker &operator=(const Worker &) = delete;
  // Model::forward (slimt/Model.cc:187-204): shortlist = sorted target ids of
  // the batch (slimt/Model.cc:117-120) or nullopt for the full vocabulary.
  Histories forward(const Input &input, const std::optional<Words> &shortlist = std::nullopt,
                    bool with_alignments = true);
  // The same pass without waiting for it (slimt_hip_translate_async): every array belongs to
  // the caller, must stay untouched until wait() returns and should be pinned. One pass in
  // flight per Worker; out_ids [B][T], out_len [B], align nullable [B][T][S] with
  // T = max(1, (size_t)(limit_factor * S)).
  void forward_async(const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S,
                     const uint32_t *shortlist, size_t n_shortlist, float limit_factor,
                     uint32_t *out_ids, uint32_t *out_len, float *align);
  // The same with the batch's lexical shortlist generated on the device, on this worker's stream,
  // from the ids themselves (Model.cc:117-120; slimt_hip_translate_async_generated).
  void forward_async_generated(slimt_hip_shortlist *generator, const uint32_t *ids, const uint32_t *lengths,
                               size_t B, size_t S, float limit_factor, uint32_t *out_ids, uint32_t *out_len,
                               float *align);
  // Several batches of one padded length S in ONE launch pair (slimt_hip_translate_many_async): what `n` workers'
  // concurrent forward calls are in the reference (Frontend.cc:207-227). Each batch keeps its own arrays, its own
  // shortlist and its own results; the workspace must hold slimt_hip_translate_many_rows() sentences.
  void forward_many_async(const slimt_hip_batch *batches, size_t n, size_t S, float limit_factor);
  // ... each batch with its own lexical shortlist, generated inside the encoder launch (slimt_hip_translate_many_async_generated)
  void forward_many_async_generated(slimt_hip_shortlist *generator, const slimt_hip_batch *batches, size_t n, size_t S,
                                    float limit_factor);
  void wait();

 private:
  const Model &model_;
  slimt_hip_ctx *ctx_ = nullptr;
};

// Raw outputs of a pass -> Histories (targets cut at out_len, alignment rows cut at the
// sentence's own length, slimt/Model.cc:95-106,163-176).
Histories collect(const uint32_t *out_ids, const uint32_t *out_len, const float *align,
                  const uint32_t *lengths, size_t B, size_t S, size_t T, bool flat = false);

}  // namespace slimt
