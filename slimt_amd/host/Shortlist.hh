// Host-side mirror of slimt::Shortlist / slimt::ShortlistGenerator
// (slimt/Shortlist.hh:15-90) for the HIP backend: the binary lexical shortlist
// lives on the device, generate() runs there (slimt_hip_shortlist_generate,
// include/slimt_hip.h) and returns the reference's sorted id list.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <utility>
#include <vector>

#include "Model.hh"
#include "slimt_hip.h"

namespace slimt {

struct View {  // slimt/Types.hh:37-40
  const void *data = nullptr;
  size_t size = 0;
};

class Shortlist {  // slimt/Shortlist.hh:15-36
 public:
  explicit Shortlist(Words words) : words_(std::move(words)) {}
  const std::vector<Word> &words() const { return words_; }
  Word reverse_map(int idx) const { return words_[static_cast<size_t>(idx)]; }
  int try_forward_map(Word w_idx) const {
    auto first = std::lower_bound(words_.begin(), words_.end(), w_idx);
    if (first != words_.end() && *first == w_idx) return static_cast<int>(std::distance(words_.begin(), first));
    return -1;
  }

 private:
  std::vector<Word> words_;  // [packed shortlist index] -> word index
};

class ShortlistGenerator {  // slimt/Shortlist.hh:38-90
 public:
  // source_vocab / target_vocab: Vocabulary::size() of the reference's two
  // Vocabulary arguments (the binding only needs their sizes)
  ShortlistGenerator(View view, size_t source_vocab, size_t target_vocab, bool shared = false,
                     bool check = false, int device = 0);
  ~ShortlistGenerator();
  ShortlistGenerator(const ShortlistGenerator &) = delete;
  ShortlistGenerator &operator=(const ShortlistGenerator &) = delete;
  // words: the batch's source tokens without padding (Input::words())
  Shortlist generate(const Words &words) const;

 private:
  slimt_hip_shortlist *handle_ = nullptr;
  size_t target_vocab_ = 0;
};

}  // namespace slimt
