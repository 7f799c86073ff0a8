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

// The batch's selected target ids, sorted: position in this list = column of the
// shortlisted output layer (slimt/Shortlist.hh:15-36: words / reverse_map /
// try_forward_map).
class Shortlist {
 public:
  explicit Shortlist(Words sorted_ids) : ids_(std::move(sorted_ids)) {}
  const Words &words() const { return ids_; }
  // column -> vocabulary id
  Word reverse_map(int column) const { return ids_.at(static_cast<size_t>(column)); }
  // vocabulary id -> column, or -1 when the id is not in the shortlist
  int try_forward_map(Word id) const {
    const auto range = std::equal_range(ids_.begin(), ids_.end(), id);
    return range.first == range.second ? -1 : static_cast<int>(range.first - ids_.begin());
  }

 private:
  Words ids_;
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
  // the device-side generator, for Worker::forward_async_generated (one handle serves every
  // worker of its device: the asynchronous path only reads it)
  slimt_hip_shortlist *handle() const { return handle_; }

 private:
  slimt_hip_shortlist *handle_ = nullptr;
  size_t target_vocab_ = 0;
};

}  // namespace slimt
