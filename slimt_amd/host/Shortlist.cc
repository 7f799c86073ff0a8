#include "Shortlist.hh"

#include <stdexcept>
#include <string>

namespace slimt {

ShortlistGenerator::ShortlistGenerator(View view, size_t source_vocab, size_t target_vocab,
                                       bool shared, bool check, int device)
    : target_vocab_(target_vocab) {
  if (slimt_hip_shortlist_create(view.data, view.size, source_vocab, target_vocab, shared ? 1 : 0,
                                 check ? 1 : 0, device, &handle_))
    throw std::runtime_error(std::string("slimt_hip_shortlist_create: ") + slimt_hip_last_error());
}

ShortlistGenerator::~ShortlistGenerator() { slimt_hip_shortlist_destroy(handle_); }

Shortlist ShortlistGenerator::generate(const Words &words) const {
  Words indices(target_vocab_);
  size_t n = 0;
  if (words.empty()) {  // only the frequent words + the multiple-of-eight patch
    const uint32_t pad = 0, len = 0;
    if (slimt_hip_shortlist_generate(handle_, &pad, &len, 1, 1, indices.data(), &n))
      throw std::runtime_error(std::string("slimt_hip_shortlist_generate: ") + slimt_hip_last_error());
  } else {
    // one row holding every word: the generator only looks at the set of words
    const uint32_t len = static_cast<uint32_t>(words.size());
    if (slimt_hip_shortlist_generate(handle_, words.data(), &len, 1, words.size(), indices.data(), &n))
      throw std::runtime_error(std::string("slimt_hip_shortlist_generate: ") + slimt_hip_last_error());
  }
  indices.resize(n);
  return Shortlist(std::move(indices));
}

}  // namespace slimt
