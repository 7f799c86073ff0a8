// slimt/hip/Engine.cc -- see Engine.hh. Added to SLIMT_SOURCES when WITH_HIP is ON.
#include "slimt/hip/Engine.hh"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

namespace slimt::hip {

namespace {

[[noreturn]] void die(const char* what) {
  std::fprintf(stderr, "slimt (hip): %s: %s\n", what, slimt_hip_last_error());
  std::abort();
}

}  // namespace

struct ModelHandle::State {
  slimt_hip_model* model = nullptr;
  std::mutex mutex;
  std::vector<Lease> idle;
  ~State() {  // contexts first: slimt_hip_ctx_destroy reads the model they were built on
    for (Lease& lease : idle) slimt_hip_ctx_destroy(lease.ctx);
    slimt_hip_model_destroy(model);
  }
};

ModelHandle::ModelHandle() = default;
ModelHandle::ModelHandle(slimt_hip_model* model) : state_(std::make_unique<State>()) {
  state_->model = model;
}
ModelHandle::ModelHandle(ModelHandle&& other) noexcept = default;
ModelHandle& ModelHandle::operator=(ModelHandle&& other) noexcept = default;
ModelHandle::~ModelHandle() = default;

slimt_hip_model* ModelHandle::get() const { return state_ ? state_->model : nullptr; }

ModelHandle::Lease ModelHandle::acquire(size_t batch, size_t length) const {
  if (!state_) die("forward on an empty model handle");
  Lease lease;
  {
    std::lock_guard<std::mutex> lock(state_->mutex);
    if (!state_->idle.empty()) {
      lease = state_->idle.back();
      state_->idle.pop_back();
    }
  }
  if (lease.ctx != nullptr &&
      (batch > lease.batch || length > lease.length || batch * length > lease.tokens)) {
    slimt_hip_ctx_destroy(lease.ctx);  // grows to the largest batch seen: a few rebuilds at most
    lease.ctx = nullptr;
  }
  if (lease.ctx == nullptr) {
    lease.batch = std::max(lease.batch, batch);
    lease.length = std::max(lease.length, length);
    lease.tokens = std::max(lease.tokens, batch * length);
    if (slimt_hip_ctx_create_budget(state_->model, lease.batch, lease.length, lease.tokens,
                                    nullptr, &lease.ctx) != 0)
      die("slimt_hip_ctx_create_budget");
  }
  return lease;
}

void ModelHandle::release(Lease lease) const {
  std::lock_guard<std::mutex> lock(state_->mutex);
  state_->idle.push_back(lease);
}

void ModelHandle::discard(Lease lease) const { slimt_hip_ctx_destroy(lease.ctx); }

ModelHandle create_model(View model, size_t encoder_layers, size_t decoder_layers,
                         size_t num_heads, int device) {
  slimt_hip_dims dims;
  dims.encoder_layers = static_cast<int32_t>(encoder_layers);
  dims.decoder_layers = static_cast<int32_t>(decoder_layers);
  dims.num_heads = static_cast<int32_t>(num_heads);
  slimt_hip_model* handle = nullptr;
  if (slimt_hip_model_create_from_bin(model.data, model.size, &dims, device,
                                      &handle) != 0)
    die("slimt_hip_model_create_from_bin");
  return ModelHandle(handle);
}

ShortlistHandle create_shortlist(View view, size_t source_vocabulary_size,
                                 size_t target_vocabulary_size, bool shared,
                                 bool check, int device) {
  if (view.data == nullptr || view.size == 0) return ShortlistHandle();
  slimt_hip_shortlist* handle = nullptr;
  if (slimt_hip_shortlist_create(view.data, view.size, source_vocabulary_size,
                                 target_vocabulary_size, shared ? 1 : 0,
                                 check ? 1 : 0, device, &handle) != 0)
    die("slimt_hip_shortlist_create");
  return ShortlistHandle(handle);
}

Words generate(slimt_hip_shortlist* generator, const Words& words,
               size_t target_vocabulary_size) {
  Words indices(target_vocabulary_size);
  if (words.empty()) {  // nothing but the frequent words: one padded row of length 0
    const uint32_t none = 0, pad = 0;
    size_t n = 0;
    if (slimt_hip_shortlist_generate(generator, &pad, &none, 1, 1,
                                     indices.data(), &n) != 0)
      die("slimt_hip_shortlist_generate");
    indices.resize(n);
    return indices;
  }
  // Input::words() is the batch's sentences back to back (slimt/Input.cc:24): one row
  const uint32_t length = static_cast<uint32_t>(words.size());
  size_t n = 0;
  if (slimt_hip_shortlist_generate(generator, words.data(), &length, 1,
                                   words.size(), indices.data(), &n) != 0)
    die("slimt_hip_shortlist_generate");
  indices.resize(n);
  return indices;
}

Histories forward(const ModelHandle& model, slimt_hip_shortlist* generator,
                  const Input& input, uint32_t eos_id) {
  const std::vector<size_t>& source_lengths = input.lengths();
  const size_t batch = source_lengths.size();  // rows in use (Input::add calls)
  const size_t length = input.indices().dim(-1);
  Histories histories;
  if (batch == 0) return histories;
  // the first step is unconditional, then steps run while i < limit_factor * S (Model.cc:141-161)
  const size_t limit = static_cast<size_t>(input.limit_factor() * static_cast<float>(length));
  const size_t steps = limit != 0 ? limit : 1;
  std::vector<uint32_t> lengths(source_lengths.begin(), source_lengths.end());
  std::vector<uint32_t> tokens(batch * steps), produced(batch);
  std::vector<float> rows(batch * steps * length);  // Model::decode records alignments always
  ModelHandle::Lease lease = model.acquire(batch, length);
  const uint32_t* ids = input.indices().data<uint32_t>();
  const int rc =
      generator != nullptr
          ? slimt_hip_translate_generated(lease.ctx, generator, ids, lengths.data(),
                                          batch, length, input.limit_factor(), eos_id,
                                          tokens.data(), produced.data(), rows.data())
          : slimt_hip_translate(lease.ctx, ids, lengths.data(), batch, length,
                                nullptr, 0, input.limit_factor(), eos_id,
                                tokens.data(), produced.data(), rows.data());
  if (rc != 0) {  // part of the call may be queued on the context's stream: it is not reused
    model.discard(lease);
    die(generator != nullptr ? "slimt_hip_translate_generated" : "slimt_hip_translate");
  }
  model.release(lease);
  histories.reserve(batch);
  for (size_t b = 0; b < batch; b++) {
    const size_t n = std::min<size_t>(produced[b], steps);
    Hypothesis hypothesis;
    hypothesis.target.assign(tokens.begin() + static_cast<std::ptrdiff_t>(b * steps),
                             tokens.begin() + static_cast<std::ptrdiff_t>(b * steps + n));
    hypothesis.alignment.reserve(n);
    for (size_t t = 0; t < n; t++) {  // update_alignment: the first lengths[b] keys of head 0 (Model.cc:84-108)
      const float* row = rows.data() + (b * steps + t) * length;
      hypothesis.alignment.emplace_back(row, row + source_lengths[b]);
    }
    histories.push_back(std::make_shared<Hypothesis>(std::move(hypothesis)));
  }
  return histories;
}

}  // namespace slimt::hip
