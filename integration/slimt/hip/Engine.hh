// slimt/hip/Engine.hh -- what slimt's classes call under SLIMT_HAS_HIP to keep the whole
// forward pass on the MI355X (integration/patches/0002-engine-hooks.patch adds the call sites
// to Transformer::Transformer, Model::Model / Model::forward and ShortlistGenerator). Written
// against the reference's own types only (slimt/Types.hh, Input.hh), so that it compiles with
// nothing but the reference's headers and include/slimt_hip.h
// (tests/test_integration_patch.py does exactly that).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>

#include "slimt/Input.hh"
#include "slimt/Types.hh"
#include "slimt_hip.h"

namespace slimt::hip {

// The device copy of a model and the device contexts (stream + workspace) that run on it. The
// contexts point into the model, so they live and die with it: ~ModelHandle destroys them before
// the model (a per-thread cache keyed by the raw model pointer would outlive it: a worker
// thread that exits after its Model would free contexts of a freed model, and a new model at the
// same address would inherit them). Movable, not copyable; an empty handle holds nothing.
class ModelHandle {
 public:
  ModelHandle();
  explicit ModelHandle(slimt_hip_model* model);
  ModelHandle(ModelHandle&& other) noexcept;
  ModelHandle& operator=(ModelHandle&& other) noexcept;
  ModelHandle(const ModelHandle&) = delete;
  ModelHandle& operator=(const ModelHandle&) = delete;
  ~ModelHandle();

  slimt_hip_model* get() const;
  explicit operator bool() const { return get() != nullptr; }

  // A context for a batch of `batch` rows of `length` tokens, taken from the idle ones (grown
  // when the batch outgrows it) or built; one per concurrent caller ends up in the pool.
  // Thread-safe: Model::forward is const and runs on every Async worker (Frontend.cc:212-226).
  struct Lease {
    slimt_hip_ctx* ctx = nullptr;
    size_t batch = 0, length = 0, tokens = 0;
  };
  Lease acquire(size_t batch, size_t length) const;
  void release(Lease lease) const;   // back to the pool
  void discard(Lease lease) const;   // after a failed call: never reused

 private:
  struct State;
  std::unique_ptr<State> state_;
};

struct ShortlistDeleter {
  void operator()(slimt_hip_shortlist* shortlist) const { slimt_hip_shortlist_destroy(shortlist); }
};
using ShortlistHandle = std::unique_ptr<slimt_hip_shortlist, ShortlistDeleter>;

// Transformer::Transformer (slimt/Transformer.cc:87-94): the weights of the Marian .bin `model`
// on `device`; aborts with the library's message when that fails (the reference aborts on a
// malformed file as well).
ModelHandle create_model(View model, size_t encoder_layers, size_t decoder_layers,
                         size_t num_heads, int device = 0);

// Model::make_shortlist_generator (slimt/Model.cc:73-82): an empty handle for an empty view.
ShortlistHandle create_shortlist(View view, size_t source_vocabulary_size,
                                 size_t target_vocabulary_size, bool shared = false,
                                 bool check = false, int device = 0);

// ShortlistGenerator::generate (slimt/Shortlist.cc:115-175) on the device, for callers that
// keep using the class (the op-by-op drop-in).
Words generate(slimt_hip_shortlist* generator, const Words& words,
               size_t target_vocabulary_size);

// Model::forward (slimt/Model.cc:187-204) = embedding + Encoder::forward + the greedy loop of
// Model::decode (:111-185), the per-batch shortlist (:117-120) included when `generator` is
// not null. Re-entrant like the const method it replaces: every concurrent call borrows a
// device context from the model's own pool.
Histories forward(const ModelHandle& model, slimt_hip_shortlist* generator,
                  const Input& input, uint32_t eos_id);

}  // namespace slimt::hip
