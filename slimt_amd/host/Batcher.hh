// Host-side mirror of the reference's batching pipeline for the HIP backend
// (SURVEY 8(f) row f2): SegmentRef / Batch / Batcher (slimt/Batcher.hh:20-115,
// slimt/Batcher.cc:20-147), the Threadsafe monitor (slimt/Batcher.hh:203-259)
// and the Async worker loop (slimt/Frontend.cc:207-227), with workers spread
// over the GPUs of a node: worker i owns one slimt::Worker (stream + workspace)
// on device i % n_devices and pulls length-bucketed, token-budgeted batches.
// Text processing is out of scope, so a Request here is just its tokenised
// segments and the slots for their Histories.
#pragma once
#include <condition_variable>
#include <cstddef>
#include <future>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "Model.hh"

namespace slimt {

template <class T>
using Ptr = std::shared_ptr<T>;
using Segment = Words;

// slimt::Request (slimt/Request.hh) reduced to what batching needs
class Request {
 public:
  Request(size_t id, std::vector<Segment> segments);
  size_t id() const { return id_; }
  size_t size() const { return segments_.size(); }
  size_t word_count(size_t index) const { return segments_[index].size(); }
  const Segment &segment(size_t index) const { return segments_[index]; }
  // Batch::complete -> SegmentRef::complete -> Request::process
  void process(size_t index, History history);
  // fulfilled when every segment has its History
  std::future<Histories> future() { return promise_.get_future(); }

 private:
  size_t id_;
  std::vector<Segment> segments_;
  Histories histories_;
  size_t pending_;
  std::mutex mutex_;
  std::promise<Histories> promise_;
};

class SegmentRef {  // slimt/Batcher.hh:20-45
 public:
  SegmentRef(size_t index, Ptr<Request> request) : index_(index), request_(std::move(request)) {}
  size_t size() const { return request_->word_count(index_); }
  const Segment &get() const { return request_->segment(index_); }
  void complete(History history) const { request_->process(index_, std::move(history)); }
  size_t index() const { return index_; }
  const Request &request() const { return *request_; }
  friend bool operator<(const SegmentRef &a, const SegmentRef &b);  // Batcher.cc:38-44

 private:
  size_t index_;
  Ptr<Request> request_;
};
using SegmentRefs = std::vector<SegmentRef>;

class Batch {  // slimt/Batcher.hh:50-84; an empty batch is poison
 public:
  size_t size() const { return segment_refs_.size(); }
  bool empty() const { return segment_refs_.empty(); }
  size_t max_length() const { return max_length_; }
  size_t token_count() const { return token_count_; }
  void add(const SegmentRef &segment_ref);
  const SegmentRefs &segment_refs() const { return segment_refs_; }
  void complete(const Histories &histories) const;

 private:
  SegmentRefs segment_refs_;
  size_t token_count_ = 0;
  size_t max_length_ = 0;
};

class Batcher {  // slimt/Batcher.hh:86-115
 public:
  Batcher(size_t max_words, size_t wrap_length, float tgt_length_limit_factor = 3.0F);
  size_t enqueue(const Ptr<Request> &request);
  Batch generate();
  void clear();

 private:
  size_t max_words_;
  std::vector<std::set<SegmentRef>> bucket_;
  size_t running_bucket_max_size_ = 0;
};

// slimt/Batcher.hh:203-259: monitor around a batcher (producer: enqueue,
// consumers: generate; an empty batch after shutdown ends a worker)
class ThreadsafeBatcher {
 public:
  ThreadsafeBatcher(size_t max_words, size_t wrap_length, float tgt_length_limit_factor)
      : backend_(max_words, wrap_length, tgt_length_limit_factor) {}
  ~ThreadsafeBatcher() { shutdown(); }
  void enqueue(const Ptr<Request> &request);
  void shutdown();
  Batch generate();

 private:
  Batcher backend_;
  size_t enqueued_ = 0;
  bool shutdown_ = false;
  std::mutex mutex_;
  std::condition_variable work_;
};

// Frontend.cc:30-40
Input convert(const Batch &batch, uint32_t pad_id, float limit_factor);

// slimt::Async (slimt/Frontend.cc:207-227) over one model replica per device.
class Async {
 public:
  struct Config {  // slimt/Frontend.hh:18-33
    size_t max_words = 1024;
    size_t wrap_length = 128;
    float tgt_length_limit_factor = 1.5F;
    size_t workers = 1;
    uint32_t pad_id = 0;
  };
  // models[d] lives on device d; worker i uses models[i % models.size()]
  Async(const Config &config, std::vector<const Model *> models);
  ~Async();
  std::future<Histories> translate(const Ptr<Request> &request,
                                   const std::optional<Words> &shortlist = std::nullopt);

 private:
  Config config_;
  ThreadsafeBatcher batcher_;
  std::optional<Words> shortlist_;
  std::vector<std::thread> workers_;
};

}  // namespace slimt
