#include "Transformer.hh"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <memory>
#include <mutex>
#include <cstring>
#include <stdexcept>

#include "Io.hh"

namespace slimt {

namespace {

[[noreturn]] void raise(const char *what) {
  throw std::runtime_error(std::string(what) + ": " + slimt_hip_last_error());
}

// The calling thread's device workspace for one model: created on first use, regrown when a larger batch arrives.
// Remembers which encoder output its K/V cache belongs to. Its context points into the model, so it is registered
// with the model (Model.hh, ThreadContexts): the Model destroys it when it goes first, the thread when it exits first --
// a thread that outlives its Model used to destroy a context of freed weights, and a new Model at the same address
// would have inherited the old one's workspace (the same flaw as the engine hooks' per-thread cache).
struct Workspace {
  std::shared_ptr<ThreadContexts> registry;  // identifies the model (never its address)
  ThreadContexts::Entry entry;               // .ctx: guarded by registry->mutex against the Model's destructor
  size_t max_B = 0, max_S = 0;
  const void *kv_of = nullptr;  // encoder_out.data() of the sequence being decoded
  size_t kv_B = 0, kv_S = 0, kv_n_shortlist = 0;
  slimt_hip_ctx *&ctx = entry.ctx;
  ~Workspace() {
    if (!registry) return;
    std::lock_guard<std::mutex> lock(registry->mutex);
    if (registry->model_alive) {  // (else the Model has destroyed the context already)
      auto &v = registry->entries;
      v.erase(std::remove(v.begin(), v.end(), &entry), v.end());
      slimt_hip_ctx_destroy(entry.ctx);
    }
    entry.ctx = nullptr;
  }
};

Workspace &workspace(const Model &model, size_t B, size_t S) {
  thread_local std::vector<std::unique_ptr<Workspace>> all;
  const std::shared_ptr<ThreadContexts> &registry = model.thread_contexts();
  Workspace *w = nullptr;
  for (size_t i = 0; i < all.size();) {
    if (all[i]->registry == registry) {
      w = all[i].get();
      ++i;
    } else if (!all[i]->registry->model_alive) {  // (a plain read: set once, under the lock, never cleared)
      all.erase(all.begin() + static_cast<std::ptrdiff_t>(i));  // that model is gone: drop its (empty) workspace
    } else {
      ++i;
    }
  }
  if (!w) {
    all.push_back(std::make_unique<Workspace>());
    w = all.back().get();
    w->registry = registry;
    std::lock_guard<std::mutex> lock(registry->mutex);
    registry->entries.push_back(&w->entry);
  }
  if (!w->ctx || B > w->max_B || S > w->max_S) {
    std::lock_guard<std::mutex> lock(registry->mutex);  // (the Model is alive: this is a call on it)
    slimt_hip_ctx_destroy(w->ctx);
    w->ctx = nullptr;
    w->max_B = std::max(B, w->max_B);
    w->max_S = std::max(S, w->max_S);
    w->kv_of = nullptr;
    if (slimt_hip_ctx_create(model.handle(), w->max_B, w->max_S, nullptr, &w->ctx)) raise("slimt_hip_ctx_create");
    if (slimt_hip_ctx_set_decode_mode(w->ctx, 1)) raise("slimt_hip_ctx_set_decode_mode");  // per-stage kernels
  }
  return *w;
}

// mask rows are prefix masks (Input.cc:49-63): length = number of unmasked keys
std::vector<uint32_t> lengths_of(const Tensor &mask, size_t B, size_t S) {
  std::vector<uint32_t> lengths(B);
  const float *m = mask.data<float>();
  for (size_t b = 0; b < B; ++b) {
    uint32_t n = 0;
    while (n < S && m[b * S + n] == 0.0F) ++n;
    lengths[b] = n;
  }
  return lengths;
}

}  // namespace

Tensor Encoder::forward(const Tensor &embedding, const Tensor &mask) const {
  const size_t B = embedding.dim(-3), S = embedding.dim(-2), D = embedding.dim(-1);
  assert(D == owner_->dim_emb() && mask.dim(-2) == B && mask.dim(-1) == S);
  Workspace &w = workspace(owner_->model(), B, S);
  const std::vector<uint32_t> lengths = lengths_of(mask, B, S);
  Tensor out(Type::f32, Shape({B, S, D}), "encoder_out");
  if (slimt_hip_encode_embedded(w.ctx, embedding.data<float>(), lengths.data(), B, S, out.data<float>()))
    raise("slimt_hip_encode_embedded");
  w.kv_of = nullptr;
  return out;
}

std::vector<Tensor> Decoder::start_states(size_t batch_size) const {
  std::vector<Tensor> states;
  for (size_t l = 0; l < owner_->decoder_layers(); ++l)
    states.emplace_back(Type::f32, Shape({batch_size, owner_->dim_emb()}), "start_state");  // zeros
  return states;
}

std::tuple<Tensor, Tensor> Decoder::step(const Tensor &encoder_out, const Tensor &mask, std::vector<Tensor> &states,
                                         const Words &previous_step, const std::optional<Words> &shortlist) const {
  const size_t B = encoder_out.dim(-3), S = encoder_out.dim(-2), D = encoder_out.dim(-1);
  const size_t Ld = owner_->decoder_layers(), H = owner_->num_heads();
  assert(states.size() == Ld && D == owner_->dim_emb());
  Workspace &w = workspace(owner_->model(), B, S);
  const size_t n_sl = shortlist ? shortlist->size() : 0;
  const bool first = previous_step.empty();
  if (first || w.kv_of != encoder_out.data<float>() || w.kv_B != B || w.kv_S != S || w.kv_n_shortlist != n_sl) {
    // a new sequence: upload the encoder output, build its cross-attention K/V once (the
    // reference recomputes them in every step, Modules.cc:248) and gather the shortlist
    const std::vector<uint32_t> lengths = lengths_of(mask, B, S);
    if (slimt_hip_decode_begin_from(w.ctx, encoder_out.data<float>(), lengths.data(), B, S,
                                    shortlist ? shortlist->data() : nullptr, n_sl))
      raise("slimt_hip_decode_begin_from");
    w.kv_of = encoder_out.data<float>();
    w.kv_B = B;
    w.kv_S = S;
    w.kv_n_shortlist = n_sl;
  }
  const size_t N = n_sl ? n_sl : owner_->vocab();
  std::vector<float> cells(Ld * B * D);
  for (size_t l = 0; l < Ld; ++l) std::memcpy(cells.data() + l * B * D, states[l].data<float>(), B * D * sizeof(float));
  Tensor logits(Type::f32, Shape({B, 1, N}), "logits");
  Tensor attn(Type::f32, Shape({B, H, 1, S}), "attn");
  if (slimt_hip_decode_step_states(w.ctx, first ? nullptr : previous_step.data(), cells.data(), logits.data<float>(),
                                   attn.data<float>(), cells.data()))
    raise("slimt_hip_decode_step_states");
  for (size_t l = 0; l < Ld; ++l) std::memcpy(states[l].data<float>(), cells.data() + l * B * D, B * D * sizeof(float));
  return {std::move(logits), std::move(attn)};
}

namespace {
size_t first_max(const float *row, size_t n) {
  size_t best = 0;
  float value = row[0];
  for (size_t c = 1; c < n; ++c)
    if (row[c] > value) {  // strict: the first maximum wins
      best = c;
      value = row[c];
    }
  return best;
}
}  // namespace

Words greedy_sample(const Tensor &logits, const Vocabulary &vocabulary, size_t batch_size) {
  const size_t n = vocabulary.size();
  Words out;
  out.reserve(batch_size);
  for (size_t b = 0; b < batch_size; ++b) out.push_back(static_cast<Word>(first_max(logits.data<float>() + b * n, n)));
  return out;
}

Words greedy_sample_from_words(const Tensor &logits, const Vocabulary & /*vocabulary*/, const Words &words,
                               size_t batch_size) {
  const size_t n = words.size();
  Words out;
  out.reserve(batch_size);
  for (size_t b = 0; b < batch_size; ++b) out.push_back(words[first_max(logits.data<float>() + b * n, n)]);
  return out;
}

void transform_embedding(Tensor &word_embedding, size_t start) {
  const size_t D = word_embedding.dim(-1), S = word_embedding.dim(-2), B = word_embedding.dim(-3);
  float *x = word_embedding.data<float>();
  const float scale = std::sqrt(static_cast<float>(D));  // Transformer.cc:34
  // sinusoidal_signal (TensorOps.cc:245-265), host libm like the reference and like the
  // table the engine uploads for its own embedding kernels (engine.cpp sinusoid_table)
  const float num_timescales = static_cast<float>(D) / 2;
  const float log_10000 = 9.210340371976184F;  // std::log(10000.0F), correctly rounded
  const float increment = log_10000 / (num_timescales - 1.0F);
  std::vector<float> signal(S * D, 0.0F);
  for (size_t p = 0; p < S; ++p)
    for (int i = 0; i < num_timescales; ++i) {
      const float v = static_cast<float>(start + p) * std::exp(static_cast<float>(i) * -increment);
      signal[p * D + static_cast<size_t>(i)] = std::sin(v);
      signal[p * D + static_cast<size_t>(i) + static_cast<size_t>(num_timescales)] = std::cos(v);
    }
  for (size_t b = 0; b < B; ++b)
    for (size_t p = 0; p < S; ++p)
      for (size_t d = 0; d < D; ++d) {
        float &e = x[(b * S + p) * D + d];
        const float scaled = e * scale;        // mul_scalar, then add_positional_embedding:
        e = scaled + signal[p * D + d];        // two roundings, like the reference's two passes
      }
}

Tensor index_select(const Tensor &embedding, const Tensor &indices, const std::string &name) {
  const size_t D = embedding.dim(-1), B = indices.dim(-2), S = indices.dim(-1);
  Tensor out(Type::f32, Shape({B, S, D}), name);
  const uint32_t *ids = indices.data<uint32_t>();
  for (size_t i = 0; i < B * S; ++i)
    std::memcpy(out.data<float>() + i * D, embedding.data<float>() + static_cast<size_t>(ids[i]) * D, D * sizeof(float));
  return out;
}

Transformer::Transformer(size_t encoder_layers, size_t decoder_layers, size_t num_heads,
                         size_t /*feed_forward_depth: read from the weights' shapes*/, View model, int device)
    : encoder_(this), decoder_(this), heads_(num_heads), decoder_layers_(decoder_layers) {
  Model::Config config;
  config.encoder_layers = encoder_layers;
  config.decoder_layers = decoder_layers;
  config.num_heads = num_heads;
  config.device = device;
  model_ = std::make_unique<Model>(config, model.data, model.size);
  int32_t D = 0, V = 0;
  if (slimt_hip_model_info(model_->handle(), &D, nullptr, &V, nullptr)) raise("slimt_hip_model_info");
  dim_emb_ = static_cast<size_t>(D);
  vocab_ = static_cast<size_t>(V);
  // the f32 embedding table of Transformer::embedding(): Wemb dequantised with the
  // reciprocal of its multiplier (Io.cc:275-283)
  for (const io::Item &item : io::load_items(model.data, model.size)) {
    if (item.name != "Wemb" || item.type != io::ItemType::ig8) continue;
    const auto *q = static_cast<const int8_t *>(item.data);
    float multiplier = 0.0F;
    std::memcpy(&multiplier, q + vocab_ * dim_emb_, sizeof(float));
    const float inverse = 1 / multiplier;
    embedding_ = Tensor(Type::f32, Shape({vocab_, dim_emb_}), "Wemb");
    float *e = embedding_.data<float>();
    for (size_t i = 0; i < vocab_ * dim_emb_; ++i) e[i] = static_cast<float>(q[i]) * inverse;
  }
  if (embedding_.size() == 0) throw std::runtime_error("Transformer: the model holds no intgemm8 Wemb");
}

Transformer::~Transformer() = default;

}  // namespace slimt
