// Multi-device translation service of the HIP backend.
//
// What it replaces in slimt: the request -> batch -> worker plumbing between
// `Async::translate` and `Model::forward` (slimt/Frontend.cc:207-227 with
// slimt/Batcher.{hh,cc}). This is NOT that code: the reference keeps ordered
// sets of segment references per length and hands one batch at a time to a
// worker that blocks in `forward`. Here
//   * waiting sentences live in per-length binary heaps keyed by arrival order
//     (LengthQueue) -- only the batch-FORMING RULE is the reference's, because
//     the padding a sentence travels with is visible in its result: walk the
//     lengths upwards and keep adding sentences while (count + 1) * length fits
//     the word budget (slimt/Batcher.cc:95-120);
//   * every worker owns TWO device contexts with pinned staging buffers and keeps
//     one batch running on the GPU while it assembles, uploads and launches the
//     next one and then unpacks the previous one's results (double buffering over
//     slimt_hip_translate_async): host work and PCIe copies hide behind kernels;
//   * workers are spread over any number of model replicas (one per GPU).
// Text processing is out of scope, so a request is its tokenised sentences.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <optional>
#include <thread>
#include <vector>

#include "Model.hh"
#include "Shortlist.hh"

namespace slimt {

// One translate() call in flight: its sentences, the slots for their results and
// the promise that is fulfilled by whichever worker delivers the last one.
class Pending {
 public:
  explicit Pending(std::vector<Words> sentences);
  size_t size() const { return sentences_.size(); }
  const Words &sentence(size_t i) const { return sentences_[i]; }
  std::future<Histories> future() { return promise_.get_future(); }
  void deliver(size_t i, History history);   // thread-safe; the last delivery fulfils the promise
  void fail(const std::exception_ptr &error);  // first failure wins; later deliveries are dropped

 private:
  std::vector<Words> sentences_;
  Histories results_;
  std::atomic<size_t> left_;
  std::atomic<bool> settled_{false};
  std::promise<Histories> promise_;
};

// A sentence waiting for a batch.
struct Unit {
  uint64_t order = 0;  // (request sequence number << 24) | sentence index: FIFO among equal lengths
  std::shared_ptr<Pending> owner;
  uint32_t index = 0;
  uint32_t length = 0;
};

// Waiting sentences, grouped by length. Not thread-safe (the service locks around it).
class LengthQueue {
 public:
  // `longest` = the longest sentence accepted; the budget must hold at least one
  // such sentence (the reference's check, slimt/Batcher.cc:85-91).
  LengthQueue(size_t max_words, size_t longest);
  // Units arrive in ascending `order` (the service numbers requests under the lock it pushes
  // under), so every length is a FIFO: O(1) per sentence -- a request of 65,536 sentences used to
  // hold the lock for its whole heap insertion while every worker waited.
  void push(Unit unit);
  // Next batch under the reference's rule (slimt/Batcher.cc:95-120): lengths
  // ascending, arrival order within a length, stop at the first sentence that
  // would push (count + 1) * its length over the budget. Empty when nothing waits.
  std::vector<Unit> take();
  // the padded length of the batch take() would return now (the length of its last sentence); 0 when nothing waits
  size_t peek_length() const;
  size_t waiting() const { return waiting_; }
  size_t longest() const { return fifos_.size() - 1; }

 private:
  size_t max_words_;
  std::vector<std::deque<Unit>> fifos_;  // fifos_[len]: ascending Unit::order
  size_t low_ = 0, high_ = 0;            // lengths outside [low_, high_] are empty
  size_t waiting_ = 0;
};

// Contexts (streams) one process can hold on one device before its hardware queues are time-sliced.
constexpr size_t kContextsPerDeviceCliff = 22;

struct ServiceConfig {
  size_t max_words = 8192;   // word budget of a batch: (B + 1) * S <= max_words
  size_t wrap_length = 128;  // longest sentence (slimt wraps there, Frontend.hh:27)
  float tgt_length_limit_factor = 1.5F;
  size_t workers_per_device = 10;  // x 2 contexts each: about 20 batches in flight per GPU
  // Merged launches (slimt_hip_translate_many_async): a worker that has taken a batch keeps taking the batches that
  // follow it in the queue -- each formed by the reference's rule under max_words, each with its own padded length,
  // arrays and results -- up to merge_batches of them, while the launch's rows x its longest length stay within
  // merge_words and no batch runs padded by more than a quarter; all of them run as ONE encoder and ONE decoder launch.
  // The reference's default batch is 1024 padded words (Frontend.hh:21-39): 32 sentences of 32 tokens fill 2 of 256 CUs
  // in the decoder; eight of them per launch pair fill what one batch of 8192 words does. 1 = never merge. With a lexical
  // shortlist every merged batch still gets ITS OWN list (Model.cc:117-120), generated inside the one encoder launch.
  size_t merge_batches = 8;
  size_t merge_words = 8192;
  uint32_t pad_id = 0;
  // one line on stderr when started with > 2 workers per device and < 8 hardware queues, or with more than
  // kContextsPerDeviceCliff contexts (2 per worker) on one device
  bool warn_hw_queues = true;
  bool alignments = true;
  bool flat_alignments = false;  // Hypothesis::alignment_flat instead of ::alignment (one block per sentence)
  // Output vocabulary of a batch, one policy for the service's lifetime:
  //  * lexical_shortlist set: the reference's own -- ShortlistGenerator::generate on every batch's
  //    source words (Model.cc:60-82,117-120; Shortlist.cc:115-175) -- run on the device, on the
  //    worker's stream, from the batch's ids in pinned memory: no host shortlist, no upload, no
  //    synchronisation. `lexical_shortlist` is the binary shortlist file (Shortlist.hh:77-84), borrowed
  //    for the constructor; one generator per device is built from it;
  //  * else `shortlist`: one fixed sorted id list (tests, benchmarks), or nullopt: the full vocabulary.
  View lexical_shortlist;
  size_t source_vocab = 0, target_vocab = 0;  // Vocabulary::size() of the two vocabularies
  bool shortlist_shared_vocab = false;        // ShortlistGenerator's `shared` (Shortlist.hh:51)
  bool shortlist_check = false;               // verify the file's checksum (Shortlist.cc:66-76)
  std::optional<Words> shortlist;
  // Test hook: called by every worker before it builds its contexts; throwing from it makes that
  // worker fail the way a failed allocation would (the others must keep serving).
  std::function<void(const Model *)> fail_worker_setup;
};

class Service {
 public:
  // replicas[d]: the model on GPU d. Worker w of device d = thread d * workers_per_device + w.
  Service(const ServiceConfig &config, std::vector<const Model *> replicas);
  ~Service();  // drains the queue, then joins the workers
  Service(const Service &) = delete;
  Service &operator=(const Service &) = delete;
  // Queue one request. Throws std::invalid_argument for an empty sentence or one longer
  // than the service accepts; a failure on a worker arrives through the future.
  std::future<Histories> translate(std::vector<Words> sentences);
  // restart the SLIMT_SERVICE_STATS counters (benchmarks: after the warm-up pass)
  void stats_reset() {
    stats_base_ = batches_.load();
    launches_base_ = launches_.load();
    merged_base_ = merged_launches_.load();
    for (auto *c : {&ns_idle_, &ns_lock_, &ns_starved_, &ns_launch_, &ns_wait_, &ns_collect_, &ns_deliver_}) c->store(0);
  }

 private:
  struct Slot;
  void work(const Model *model, slimt_hip_shortlist *generator);
  std::vector<Unit> next_batch(bool may_block, std::vector<size_t> *parts = nullptr);
  void launch(Slot &slot, std::vector<Unit> &batch, slimt_hip_shortlist *generator, std::vector<size_t> &parts);
  void finish(Slot &slot);
  void retire(const std::exception_ptr &error);

  ServiceConfig config_;
  size_t longest_;
  LengthQueue queue_;
  std::vector<std::unique_ptr<ShortlistGenerator>> generators_;  // one per device in use (lexical shortlist)
  uint64_t sequence_ = 0;
  std::atomic<uint64_t> batches_{0};  // batches launched so far (Hypothesis::batch)
  std::atomic<uint64_t> launches_{0}, merged_launches_{0};  // launch pairs so far, and those that carried more than one batch
  uint64_t stats_base_ = 0, launches_base_ = 0, merged_base_ = 0;
  // where the workers' time goes, in nanoseconds (printed by the destructor when SLIMT_SERVICE_STATS is set)
  std::atomic<uint64_t> ns_idle_{0}, ns_lock_{0}, ns_starved_{0}, ns_launch_{0}, ns_wait_{0}, ns_collect_{0}, ns_deliver_{0};
  bool closing_ = false;
  size_t live_workers_ = 0;          // workers that can take batches
  std::exception_ptr dead_error_;    // set once the last of them has failed: requests fail with it
  std::mutex mutex_;
  std::condition_variable wake_;
  std::vector<std::thread> threads_;
};

}  // namespace slimt
