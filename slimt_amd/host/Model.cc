#include "Model.hh"

#include <algorithm>
#include <cassert>
#include <stdexcept>

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

Model::Model(const Config &config, const void *model_bin, size_t size) : config_(config) {
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
}

Model::~Model() {
  if (owned_) slimt_hip_model_destroy(model_);
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

void Worker::wait() {
  if (slimt_hip_ctx_synchronize(ctx_)) raise("slimt_hip_ctx_synchronize");
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
