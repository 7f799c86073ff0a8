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

// Owning handles (movable, not copyable; an empty handle is a null pointer).
struct ModelDeleter {
  void operator()(slimt_hip_model* model) const { slimt_hip_model_destroy(model); }
};
struct ShortlistDeleter {
  void operator()(slimt_hip_shortlist* shortlist) const { slimt_hip_shortlist_destroy(shortlist); }
};
using ModelHandle = std::unique_ptr<slimt_hip_model, ModelDeleter>;
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
// not null. Re-entrant like the const method it replaces: every calling thread keeps its own
// device context (stream + workspace) per model, built on first use and grown on demand.
Histories forward(slimt_hip_model* model, slimt_hip_shortlist* generator,
                  const Input& input, uint32_t eos_id);

}  // namespace slimt::hip
