#include "Batcher.hh"

#include <algorithm>
#include <cassert>
#include <stdexcept>

namespace slimt {

Request::Request(size_t id, std::vector<Segment> segments)
    : id_(id), segments_(std::move(segments)), histories_(segments_.size()), pending_(segments_.size()) {
  if (pending_ == 0) promise_.set_value(histories_);
}

void Request::process(size_t index, History history) {
  std::unique_lock<std::mutex> lock(mutex_);
  histories_[index] = std::move(history);
  if (--pending_ == 0) promise_.set_value(histories_);
}

bool operator<(const SegmentRef &a, const SegmentRef &b) {
  // among requests only the sequence id gives priority (Batcher.cc:33-44)
  if (a.request_ == b.request_) return a.index_ < b.index_;
  if (a.request_->id() != b.request_->id()) return a.request_->id() < b.request_->id();
  return a.request_ < b.request_;
}

void Batch::add(const SegmentRef &segment_ref) {  // Batcher.cc:54-58
  segment_refs_.push_back(segment_ref);
  token_count_ += segment_ref.size();
  max_length_ = std::max<size_t>(max_length_, segment_ref.size());
}

void Batch::complete(const Histories &histories) const {  // Batcher.cc:60-64
  for (size_t i = 0; i < segment_refs_.size(); i++) segment_refs_[i].complete(histories[i]);
}

Batcher::Batcher(size_t max_words, size_t wrap_length, float tgt_length_limit_factor)
    : max_words_(max_words) {
  // slack for sentences that overflow the wrap length (Batcher.cc:77-92)
  size_t pivot_slack = static_cast<size_t>(wrap_length * tgt_length_limit_factor - wrap_length);
  bucket_.resize(wrap_length + pivot_slack + 1);
  if (bucket_.size() - 1 > max_words_)
    throw std::invalid_argument(
        "wrap_length > max_words will lead to sentences longer than what can fit in a batch");
}

Batch Batcher::generate() {  // Batcher.cc:95-120: shortest buckets first, greedy on the padded size
  Batch batch;
  size_t padded_batch_size = 0;
  for (size_t length = 0; length <= running_bucket_max_size_; length++) {
    auto p = bucket_[length].begin();
    while (p != bucket_[length].end()) {
      padded_batch_size = (batch.size() + 1) * length;
      if (padded_batch_size <= max_words_) {
        auto q = p++;
        batch.add(*q);
        bucket_[length].erase(q);
      } else {
        assert(!batch.empty());
        return batch;
      }
    }
  }
  return batch;
}

size_t Batcher::enqueue(const Ptr<Request> &request) {  // Batcher.cc:122-147 (no cache here)
  size_t to_be_translated = 0;
  for (size_t i = 0; i < request->size(); i++) {
    SegmentRef sentence(i, request);
    size_t bucket_id = sentence.size();
    if (bucket_id >= bucket_.size()) bucket_.resize(bucket_id + 1);
    bucket_[bucket_id].insert(sentence);
    running_bucket_max_size_ = std::max<size_t>(bucket_id, running_bucket_max_size_);
    to_be_translated += 1;
  }
  return to_be_translated;
}

void Batcher::clear() {
  for (auto &item : bucket_) item.clear();
}

void ThreadsafeBatcher::enqueue(const Ptr<Request> &request) {
  std::unique_lock<std::mutex> lock(mutex_);
  assert(!shutdown_);
  enqueued_ += backend_.enqueue(request);
  work_.notify_all();
}

void ThreadsafeBatcher::shutdown() {
  std::unique_lock<std::mutex> lock(mutex_);
  shutdown_ = true;
  work_.notify_all();
}

Batch ThreadsafeBatcher::generate() {
  std::unique_lock<std::mutex> lock(mutex_);
  work_.wait(lock, [this]() { return enqueued_ || shutdown_; });
  Batch batch = backend_.generate();
  assert(!batch.empty() || shutdown_);
  enqueued_ -= batch.size();
  return batch;
}

Input convert(const Batch &batch, uint32_t pad_id, float limit_factor) {
  Input input(batch.size(), batch.max_length(), pad_id, limit_factor);
  for (const auto &segment_ref : batch.segment_refs()) input.add(segment_ref.get());
  return input;
}

Async::Async(const Config &config, std::vector<const Model *> models)
    : config_(config), batcher_(config.max_words, config.wrap_length, config.tgt_length_limit_factor) {
  if (models.empty()) throw std::invalid_argument("Async needs at least one model replica");
  // the engine handles sources up to 128 tokens (the reference's wrap length, Frontend.hh:27);
  // a batch satisfies (B + 1) * S <= max_words (Batcher.cc:103-104)
  const size_t max_length = std::min<size_t>(config.wrap_length, 128);
  for (size_t i = 0; i < config.workers; i++) {
    const Model *model = models[i % models.size()];
    workers_.emplace_back([this, model, max_length]() {
      // one stream + workspace per worker thread, on the worker's device
      Worker worker(*model, config_.max_words, max_length, config_.max_words + max_length);
      Batch batch = batcher_.generate();
      while (!batch.empty()) {
        Input input = convert(batch, config_.pad_id, config_.tgt_length_limit_factor);
        Histories histories = worker.forward(input, shortlist_, true);
        batch.complete(histories);
        batch = batcher_.generate();
      }
    });
  }
}

Async::~Async() {
  batcher_.shutdown();
  for (auto &w : workers_) w.join();
}

std::future<Histories> Async::translate(const Ptr<Request> &request,
                                        const std::optional<Words> &shortlist) {
  shortlist_ = shortlist;  // one shortlist policy for the service (set before the first request)
  std::future<Histories> f = request->future();
  batcher_.enqueue(request);
  return f;
}

}  // namespace slimt
