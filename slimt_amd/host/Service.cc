#include "Service.hh"

#include <algorithm>
#include <stdexcept>
#include <string>

namespace slimt {

// ---- Pending -----------------------------------------------------------------

Pending::Pending(std::vector<Words> sentences)
    : sentences_(std::move(sentences)), results_(sentences_.size()), left_(sentences_.size()) {
  if (sentences_.empty()) {
    settled_ = true;
    promise_.set_value({});
  }
}

void Pending::deliver(size_t i, History history) {
  results_[i] = std::move(history);  // every slot is written by exactly one worker
  if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1 && !settled_.exchange(true))
    promise_.set_value(std::move(results_));
}

void Pending::fail(const std::exception_ptr &error) {
  if (!settled_.exchange(true)) promise_.set_exception(error);
}

// ---- LengthQueue ---------------------------------------------------------------

namespace {
struct Later {  // std::*_heap build max-heaps: "later arrival" on top of the comparison = min-heap on order
  bool operator()(const Unit &a, const Unit &b) const { return a.order > b.order; }
};
}  // namespace

LengthQueue::LengthQueue(size_t max_words, size_t longest) : max_words_(max_words), heaps_(longest + 1) {
  if (longest > max_words)
    throw std::invalid_argument("wrap_length > max_words: the longest sentence would not fit a batch (" +
                                std::to_string(longest) + " > " + std::to_string(max_words) + ")");
  low_ = longest + 1;
}

void LengthQueue::push(Unit unit) {
  const size_t len = unit.length;
  if (len >= heaps_.size()) throw std::invalid_argument("sentence longer than the queue accepts");
  auto &heap = heaps_[len];
  heap.push_back(std::move(unit));
  std::push_heap(heap.begin(), heap.end(), Later());
  low_ = std::min(low_, len);
  high_ = std::max(high_, len);
  ++waiting_;
}

std::vector<Unit> LengthQueue::take() {
  std::vector<Unit> batch;
  if (waiting_ == 0) return batch;
  for (size_t len = low_; len <= high_; ++len) {
    auto &heap = heaps_[len];
    while (!heap.empty()) {
      // all rows are padded to the longest one = the one being added (lengths ascend)
      if ((batch.size() + 1) * len > max_words_) goto done;
      std::pop_heap(heap.begin(), heap.end(), Later());
      batch.push_back(std::move(heap.back()));
      heap.pop_back();
    }
  }
done:
  waiting_ -= batch.size();
  while (low_ <= high_ && heaps_[low_].empty()) ++low_;
  if (waiting_ == 0) {
    low_ = heaps_.size();
    high_ = 0;
  }
  return batch;
}

// ---- Service -------------------------------------------------------------------

namespace {
[[noreturn]] void raise(const char *what) {
  throw std::runtime_error(std::string(what) + ": " + slimt_hip_last_error());
}

// Pinned host array that only grows.
template <class T>
class Pinned {
 public:
  Pinned() = default;
  Pinned(const Pinned &) = delete;
  Pinned &operator=(const Pinned &) = delete;
  ~Pinned() { slimt_hip_host_free(p_); }
  T *ensure(size_t n) {
    if (n > cap_) {
      slimt_hip_host_free(p_);
      p_ = nullptr;
      cap_ = 0;
      void *q = nullptr;
      if (slimt_hip_host_alloc(n * sizeof(T), &q)) raise("slimt_hip_host_alloc");
      p_ = static_cast<T *>(q);
      cap_ = n;
    }
    return p_;
  }
  T *get() const { return p_; }

 private:
  T *p_ = nullptr;
  size_t cap_ = 0;
};
}  // namespace

// One of a worker's two pipelines: a device context and the staging of one batch.
struct Service::Slot {
  std::unique_ptr<Worker> worker;
  Pinned<uint32_t> ids, lengths, out_ids, out_len, shortlist;
  Pinned<float> align;
  std::vector<Unit> batch;  // non-empty while a translate is in flight on this slot
  size_t B = 0, S = 0, T = 0;
};

Service::Service(const ServiceConfig &config, std::vector<const Model *> replicas)
    : config_(config),
      // the engine handles sources up to 128 tokens (the reference's wrap length, Frontend.hh:27)
      longest_(std::min<size_t>(config.wrap_length, 128)),
      queue_(config.max_words, longest_) {
  if (replicas.empty()) throw std::invalid_argument("Service needs at least one model replica");
  if (config.workers_per_device == 0) throw std::invalid_argument("Service needs at least one worker");
  for (const Model *model : replicas)
    for (size_t w = 0; w < config.workers_per_device; ++w)
      threads_.emplace_back([this, model]() { work(model); });
}

Service::~Service() {
  {
    std::lock_guard<std::mutex> lock(mutex_);
    closing_ = true;
  }
  wake_.notify_all();
  for (auto &t : threads_) t.join();
}

std::future<Histories> Service::translate(std::vector<Words> sentences) {
  for (const Words &s : sentences) {
    if (s.empty()) throw std::invalid_argument("empty sentence (a sentence holds at least its EOS)");
    if (s.size() > longest_)
      throw std::invalid_argument("sentence of " + std::to_string(s.size()) + " tokens: longer than " +
                                  std::to_string(longest_) + " (wrap it first)");
  }
  auto pending = std::make_shared<Pending>(std::move(sentences));
  std::future<Histories> result = pending->future();
  if (pending->size() == 0) return result;
  if (pending->size() >= (1u << 24)) throw std::invalid_argument("request of more than 2^24 sentences");
  {
    std::lock_guard<std::mutex> lock(mutex_);
    if (closing_) throw std::runtime_error("Service is shutting down");
    const uint64_t seq = sequence_++;
    for (size_t i = 0; i < pending->size(); ++i) {
      Unit u;
      u.order = (seq << 24) | i;
      u.owner = pending;
      u.index = static_cast<uint32_t>(i);
      u.length = static_cast<uint32_t>(pending->sentence(i).size());
      queue_.push(std::move(u));
    }
  }
  wake_.notify_all();
  return result;
}

// The next batch, or an empty one: nothing waits (may_block false) / the service closes.
std::vector<Unit> Service::next_batch(bool may_block) {
  std::unique_lock<std::mutex> lock(mutex_);
  if (may_block) wake_.wait(lock, [this]() { return queue_.waiting() > 0 || closing_; });
  return queue_.take();
}

void Service::launch(Slot &slot, std::vector<Unit> batch) {
  const size_t B = batch.size();
  const size_t S = batch.back().length;  // lengths ascend inside a batch
  const size_t T = std::max<size_t>(1, static_cast<size_t>(config_.tgt_length_limit_factor * static_cast<float>(S)));
  uint32_t *ids = slot.ids.ensure(B * S);
  uint32_t *lengths = slot.lengths.ensure(B);
  std::fill(ids, ids + B * S, config_.pad_id);
  for (size_t b = 0; b < B; ++b) {
    const Words &w = batch[b].owner->sentence(batch[b].index);
    std::copy(w.begin(), w.end(), ids + b * S);
    lengths[b] = static_cast<uint32_t>(w.size());
  }
  const uint32_t *sl = nullptr;
  size_t n_sl = 0;
  if (config_.shortlist) {
    n_sl = config_.shortlist->size();
    sl = slot.shortlist.get();
  }
  slot.B = B;
  slot.S = S;
  slot.T = T;
  slot.batch = std::move(batch);
  slot.worker->forward_async(ids, lengths, B, S, sl, n_sl, config_.tgt_length_limit_factor,
                             slot.out_ids.ensure(B * T), slot.out_len.ensure(B),
                             config_.alignments ? slot.align.ensure(B * T * S) : nullptr);
}

void Service::finish(Slot &slot) {
  std::vector<Unit> batch = std::move(slot.batch);
  slot.batch.clear();
  slot.worker->wait();
  Histories histories = collect(slot.out_ids.get(), slot.out_len.get(),
                                config_.alignments ? slot.align.get() : nullptr, slot.lengths.get(), slot.B,
                                slot.S, slot.T);
  for (size_t b = 0; b < batch.size(); ++b) batch[b].owner->deliver(batch[b].index, std::move(histories[b]));
}

void Service::work(const Model *model) {
  Slot slots[2];
  size_t cur = 0;
  auto fail_batch = [](std::vector<Unit> &batch, const std::exception_ptr &error) {
    for (Unit &u : batch) u.owner->fail(error);
    batch.clear();
  };
  try {
    for (Slot &s : slots) {
      // (B + 1) * S <= max_words: at most max_words - 1 rows, at most max_words padded tokens
      s.worker = std::make_unique<Worker>(*model, config_.max_words, longest_, config_.max_words);
      if (config_.shortlist) {
        uint32_t *sl = s.shortlist.ensure(config_.shortlist->size());
        std::copy(config_.shortlist->begin(), config_.shortlist->end(), sl);
      }
    }
  } catch (...) {
    // this worker cannot run: every batch it would have taken fails instead of hanging
    const std::exception_ptr error = std::current_exception();
    for (std::vector<Unit> batch = next_batch(true); !batch.empty(); batch = next_batch(true))
      fail_batch(batch, error);
    return;
  }
  for (;;) {
    Slot &mine = slots[cur], &other = slots[cur ^ 1];
    // with a batch in flight on the other slot, only take work that is already there
    std::vector<Unit> batch = next_batch(other.batch.empty());
    if (batch.empty()) {
      if (other.batch.empty()) break;  // closing and drained
      try {
        finish(other);
      } catch (...) {
        fail_batch(other.batch, std::current_exception());
      }
      continue;
    }
    try {
      launch(mine, std::move(batch));
    } catch (...) {
      fail_batch(mine.batch.empty() ? batch : mine.batch, std::current_exception());
    }
    if (!other.batch.empty()) {
      try {
        finish(other);  // the previous batch, while `mine` runs on the GPU
      } catch (...) {
        fail_batch(other.batch, std::current_exception());
      }
    }
    cur ^= 1;
  }
}

}  // namespace slimt
