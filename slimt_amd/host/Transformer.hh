// Class-level mirror of slimt/Transformer.hh:15-72 for the HIP backend: Encoder,
// Decoder, Transformer, greedy_sample*, transform_embedding with the REFERENCE'S
// signatures (host Tensors in and out), so that slimt::Model::forward / decode
// (slimt/Model.cc:111-204) compile against it unchanged. The compute runs on the
// device through the C ABI (include/slimt_hip.h); the workspace (stream + device
// buffers) is per calling thread, so the const methods stay re-entrant the way the
// reference's are (Async calls them from several workers, Frontend.cc:212-226).
//
// This is the drop-in path: it moves every intermediate tensor through host memory
// because the reference interface does. The fast path is Model.hh's Worker /
// Service.hh's Service (slimt_hip_translate*: two launches per batch).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include "Model.hh"
#include "Shortlist.hh"  // View
#include "Tensor.hh"

namespace slimt {

// What the sampling functions need of slimt::Vocabulary (slimt/Vocabulary.hh): its
// size and special ids. Text processing itself is outside this backend.
class Vocabulary {
 public:
  explicit Vocabulary(size_t size, Word eos_id = 0, Word pad_id = 0) : size_(size), eos_(eos_id), pad_(pad_id) {}
  size_t size() const { return size_; }
  Word eos_id() const { return eos_; }
  Word pad_id() const { return pad_; }

 private:
  size_t size_;
  Word eos_, pad_;
};

class Transformer;

class Encoder {  // slimt/Transformer.hh:15-25
 public:
  // embedding f32 [B,S,D] (after transform_embedding), mask f32 [B,S] (0 token /
  // -99999999 pad, Input.cc:49-63) -> encoder output f32 [B,S,D]
  Tensor forward(const Tensor &embedding, const Tensor &mask) const;

 private:
  friend class Transformer;
  explicit Encoder(const Transformer *owner) : owner_(owner) {}
  const Transformer *owner_;
};

class Decoder {  // slimt/Transformer.hh:27-44
 public:
  // one f32 [B,D] tensor of zeros per decoder layer (Transformer.cc:78-85)
  std::vector<Tensor> start_states(size_t batch_size) const;
  // -> (logits f32 [B,1,N], attention of the last layer f32 [B,H,1,S]); `states` are
  // updated in place; previous_step empty = first step (Transformer.cc:138-144)
  std::tuple<Tensor, Tensor> step(const Tensor &encoder_out, const Tensor &mask, std::vector<Tensor> &states,
                                  const Words &previous_step, const std::optional<Words> &shortlist) const;

 private:
  friend class Transformer;
  explicit Decoder(const Transformer *owner) : owner_(owner) {}
  const Transformer *owner_;
};

// slimt/Transformer.hh:48-52 (first maximum wins, Transformer.cc:287-297,319-329)
Words greedy_sample(const Tensor &logits, const Vocabulary &vocabulary, size_t batch_size);
Words greedy_sample_from_words(const Tensor &logits, const Vocabulary &vocabulary, const Words &words,
                               size_t batch_size);
// slimt/Transformer.hh:54 (x * sqrt(D) + sinusoid(start + position), Transformer.cc:24-49)
void transform_embedding(Tensor &word_embedding, size_t start = 0);
// slimt/TensorOps.hh index_select as Model::forward uses it (Model.cc:195-197):
// rows of `embedding` [V,D] for indices u32 [B,S] -> f32 [B,S,D]
Tensor index_select(const Tensor &embedding, const Tensor &indices, const std::string &name = "");

class Transformer {  // slimt/Transformer.hh:56-72
 public:
  // `model`: a Marian .bin in memory, borrowed for the duration of the constructor
  Transformer(size_t encoder_layers, size_t decoder_layers, size_t num_heads, size_t feed_forward_depth,
              View model, int device = 0);
  ~Transformer();
  const Tensor &embedding() const { return embedding_; }  // f32 [V,D], dequantised (Io.cc:275-283)
  const Encoder &encoder() const { return encoder_; }
  const Decoder &decoder() const { return decoder_; }
  const Model &model() const { return *model_; }
  size_t dim_emb() const { return dim_emb_; }
  size_t num_heads() const { return heads_; }
  size_t decoder_layers() const { return decoder_layers_; }
  size_t vocab() const { return vocab_; }

 private:
  std::unique_ptr<Model> model_;
  Tensor embedding_;
  Encoder encoder_;
  Decoder decoder_;
  size_t dim_emb_ = 0, heads_ = 0, decoder_layers_ = 0, vocab_ = 0;
};

}  // namespace slimt
