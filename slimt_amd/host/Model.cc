#include "Model.hh"

#include <algorithm>
#include <cassert>
#include <stdexcept>
#include <string>

#include "Io.hh"

namespace slimt {

namespace {
[[noreturn]] void raise(const char *what) {
  throw std::runtime_error(std::string(what) + ": " + slimt_hip_last_error());
}
}  // namespace

void Input::add(const Words &words) {
  assert(words.size() <= sequence_length_);
  assert(lengths_.size() < batch_size_);
  const size_t row = lengths_.size();
  for (size_t j = 0; j < words.size(); ++j) indices_[row * sequence_length_ + j] = words[j];
  words_.insert(words_.end(), words.begin(), words.end());
  lengths_.push_back(static_cast<uint32_t>(words.size()));
}

Model::Model(const Config &config, const void *model_bin, size_t size, const void *lexical_shortlist,
             size_t lexical_shortlist_size)
    : config_(config) {
  std::vector<io::Item> items = io::load_items(model_bin, size);
  std::vector<slimt_hip_param> params;
  params.reserve(items.size());
  for (const io::Item &it : items) {
    if (it.type != io::ItemType::f32 && it.type != io::ItemType::ig8) continue;  // e.g. special:model.yml
    slimt_hip_param p;
    p.name = it.name.c_str();
    p.type = it.type == io::ItemType::f32 ? 0 : 1;
    p.rows = it.shape.size() >= 2 ? it.shape[it.shape.size() - 2] : 1;
    p.cols = it.shape.empty() ? 1 : it.shape.back();
    p.data = it.data;
    p.bytes = it.bytes;
    params.push_back(p);
  }
  slimt_hip_dims dims;
  dims.encoder_layers = static_cast<int32_t>(config.encoder_layers);
  dims.decoder_layers = static_cast<int32_t>(config.decoder_layers);
  dims.num_heads = static_cast<int32_t>(config.num_heads);
  if (slimt_hip_model_create(params.data(), params.size(), &dims, config.device, &model_))
    raise("slimt_hip_model_create");
  if (lexical_shortlist && lexical_shortlist_size) {
    // both Vocabulary arguments of the reference's generator are the model's (Model.cc:73-74)
    int32_t vocab = 0;
    if (slimt_hip_model_info(model_, nullptr, nullptr, &vocab, nullptr) ||
        slimt_hip_shortlist_create(lexical_shortlist, lexical_shortlist_size, static_cast<size_t>(vocab),
                                   static_cast<size_t>(vocab), /*shared=*/0, /*check=*/0, config.device,
                                   &generator_)) {
      const std::string why = slimt_hip_last_error();
      slimt_hip_model_destroy(model_);
      throw std::runtime_error("slimt_hip_shortlist_create: " + why);
    }
  }
}

Model::~Model() {
  {  // the contexts threads still keep for this model (host/Transformer.cc): gone before the weights they point into
    std::lock_guard<std::mutex> lock(thread_contexts_->mutex);
    for (ThreadContexts::Entry *e : thread_contexts_->entries) {
      slimt_hip_ctx_destroy(e->ctx);
      e->ctx = nullptr;
    }
    thread_contexts_->entries.clear();
    thread_contexts_->model_alive = false;
  }
  for (Lease &l : idle_) slimt_hip_ctx_destroy(l.ctx);
  if (generator_) slimt_hip_shortlist_destroy(generator_);
  if (owned_) slimt_hip_model_destroy(model_);
}

size_t Model::contexts_built() const {
  std::lock_guard<std::mutex> lock(pool_mu_);
  return built_;
}

Histories Model::forward(const Input &input) const {
  const size_t B = input.lengths().size(), S = input.sequence_length();
  if (B == 0) return {};
  const size_t Tmax = static_cast<size_t>(input.limit_factor() * static_cast<float>(S));
  const size_t T = Tmax ? Tmax : 1;
  Lease lease;
  {
    std::lock_guard<std::mutex> lock(pool_mu_);
    if (!idle_.empty()) {
      lease = idle_.back();
      idle_.pop_back();
    }
  }
  if (lease.ctx && (B > lease.max_batch || S > lease.max_length || B * S > lease.max_tokens)) {
    slimt_hip_ctx_destroy(lease.ctx);  // grows to the largest batch seen: at most a few rebuilds
    lease.ctx = nullptr;
  }
  if (!lease.ctx) {
    lease.max_batch = std::max(B, lease.max_batch);
    lease.max_length = std::max(S, lease.max_length);
    lease.max_tokens = std::max(B * S, lease.max_tokens);
    if (slimt_hip_ctx_create_budget(model_, lease.max_batch, lease.max_length, lease.max_tokens, nullptr, &lease.ctx))
      raise("slimt_hip_ctx_create_budget");
    std::lock_guard<std::mutex> lock(pool_mu_);
    ++built_;
  }
  std::vector<uint32_t> out_ids(B * T), out_len(B);
  std::vector<float> align(B * T * S);  // Model::decode records alignments unconditionally (Model.cc:156,170)
  const int rc = generator_
                     ? slimt_hip_translate_generated(lease.ctx, generator_, input.indices().data(), input.lengths().data(),
                                                     B, S, input.limit_factor(), config_.eos_id, out_ids.data(),
                                                     out_len.data(), align.data())
                     : slimt_hip_translate(lease.ctx, input.indices().data(), input.lengths().data(), B, S, nullptr, 0,
                                           input.limit_factor(), config_.eos_id, out_ids.data(), out_len.data(),
                                           align.data());
  const std::string why = rc ? slimt_hip_last_error() : "";
  if (rc) {
    // whatever failed (rc > 0: a hipError_t from the runtime, rc < 0: one of the library's own checks), part of the
    // call may already be queued on this context's stream while the host vectors above are about to be freed:
    // the context is destroyed (which drains its stream), never pooled
    slimt_hip_ctx_destroy(lease.ctx);
    lease.ctx = nullptr;
  } else {
    std::lock_guard<std::mutex> lock(pool_mu_);
    idle_.push_back(lease);
  }
  if (rc) throw std::runtime_error((generator_ ? "slimt_hip_translate_generated: " : "slimt_hip_translate: ") + why);
  return collect(out_ids.data(), out_len.data(), align.data(), input.lengths().data(), B, S, T);
}

Worker::Worker(const Model &model, size_t max_batch, size_t max_length, size_t max_tokens)
    : model_(model) {
  if (slimt_hip_ctx_create_budget(model.handle(), max_batch, max_length,
                                  max_tokens ? max_tokens : max_batch * max_length, nullptr, &ctx_))
    raise("slimt_hip_ctx_create_budget");
}

Worker::~Worker() { slimt_hip_ctx_destroy(ctx_); }

Histories Worker::forward(const Input &input, const std::optional<Words> &shortlist,
                          bool with_alignments) {
  const size_t B = input.lengths().size(), S = input.sequence_length();
  const size_t Tmax = static_cast<size_t>(input.limit_factor() * static_cast<float>(S));
  const size_t T = Tmax ? Tmax : 1;
  std::vector<uint32_t> out_ids(B * T), out_len(B);
  std::vector<float> align(with_alignments ? B * T * S : 0);
  const uint32_t *sl = shortlist ? shortlist->data() : nullptr;
  const size_t n_sl = shortlist ? shortlist->size() : 0;
  if (slimt_hip_translate(ctx_, input.indices().data(), input.lengths().data(), B, S, sl, n_sl,
                          input.limit_factor(), model_.config().eos_id, out_ids.data(),
                          out_len.data(), with_alignments ? align.data() : nullptr))
    raise("slimt_hip_translate");
  return collect(out_ids.data(), out_len.data(), with_alignments ? align.data() : nullptr,
                 input.lengths().data(), B, S, T);
}

void Worker::forward_async(const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S,
                           const uint32_t *shortlist, size_t n_shortlist, float limit_factor,
                           uint32_t *out_ids, uint32_t *out_len, float *align) {
  if (slimt_hip_translate_async(ctx_, ids, lengths, B, S, shortlist, n_shortlist, limit_factor,
                                model_.config().eos_id, out_ids, out_len, align))
    raise("slimt_hip_translate_async");
}

void Worker::forward_async_generated(slimt_hip_shortlist *generator, const uint32_t *ids,
                                     const uint32_t *lengths, size_t B, size_t S, float limit_factor,
                                     uint32_t *out_ids, uint32_t *out_len, float *align) {
  if (slimt_hip_translate_async_generated(ctx_, generator, ids, lengths, B, S, limit_factor,
                                          model_.config().eos_id, out_ids, out_len, align))
    raise("slimt_hip_translate_async_generated");
}

void Worker::forward_many_async(const slimt_hip_batch *batches, size_t n, size_t S, float limit_factor) {
  if (slimt_hip_translate_many_async(ctx_, batches, n, S, limit_factor, model_.config().eos_id))
    raise("slimt_hip_translate_many_async");
}

void Worker::forward_many_async_generated(slimt_hip_shortlist *generator, const slimt_hip_batch *batches, size_t n, size_t S,
                                          float limit_factor) {
  if (slimt_hip_translate_many_async_generated(ctx_, generator, batches, n, S, limit_factor, model_.config().eos_id))
    raise("slimt_hip_translate_many_async_generated");
}

void Worker::wait() {
  if (slimt_hip_ctx_synchronize(ctx_)) raise("slimt_hip_ctx_synchronize");
}

void expand_alignment(Hypothesis &h) {
  const size_t n = h.target.size();
  if (n == 0 || !h.alignment.empty()) return;
  const size_t len = h.alignment_flat.size() / n;  // (an empty source sentence: n empty rows, as collect() builds them)
  h.alignment.reserve(n);
  for (size_t t = 0; t < n; ++t) {
    const float *row = h.alignment_flat.data() + t * len;
    h.alignment.emplace_back(row, row + len);
  }
  std::vector<float>().swap(h.alignment_flat);
}

Histories collect(const uint32_t *out_ids, const uint32_t *out_len, const float *align,
                  const uint32_t *lengths, size_t B, size_t S, size_t T, bool flat) {
  Histories histories;
  histories.reserve(B);
  for (size_t b = 0; b < B; ++b) {
    auto hyp = std::make_shared<Hypothesis>();
    hyp->padded_length = S;
    const size_t n = out_len[b] < T ? out_len[b] : T;
    hyp->target.assign(out_ids + b * T, out_ids + b * T + n);
    if (align && flat) {
      const size_t len = lengths[b];
      hyp->alignment_flat.resize(n * len);
      for (size_t t = 0; t < n; ++t) {
        const float *row = align + (b * T + t) * S;
        std::copy(row, row + len, hyp->alignment_flat.begin() + static_cast<std::ptrdiff_t>(t * len));
      }
    } else if (align) {
      const size_t len = lengths[b];
      hyp->alignment.reserve(n);
      for (size_t t = 0; t < n; ++t) {
        const float *row = align + (b * T + t) * S;
        hyp->alignment.emplace_back(row, row + len);
      }
    }
    histories.push_back(std::move(hyp));
  }
  return histories;
}

}  // namespace slimt
