#include "Service.hh"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <stdexcept>
#include <string>
#include <utility>

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

LengthQueue::LengthQueue(size_t max_words, size_t longest) : max_words_(max_words), fifos_(longest + 1) {
  if (longest > max_words)
    throw std::invalid_argument("wrap_length > max_words: the longest sentence would not fit a batch (" +
                                std::to_string(longest) + " > " + std::to_string(max_words) + ")");
  low_ = longest + 1;
}

void LengthQueue::push(Unit unit) {
  const size_t len = unit.length;
  if (len >= fifos_.size()) throw std::invalid_argument("sentence longer than the queue accepts");
  auto &fifo = fifos_[len];
  if (!fifo.empty() && fifo.back().order > unit.order) {  // out-of-order arrival (not the service's): keep it sorted
    auto at = std::upper_bound(fifo.begin(), fifo.end(), unit, [](const Unit &a, const Unit &b) { return a.order < b.order; });
    fifo.insert(at, std::move(unit));
  } else {
    fifo.push_back(std::move(unit));
  }
  low_ = std::min(low_, len);
  high_ = std::max(high_, len);
  ++waiting_;
}

std::vector<Unit> LengthQueue::take() {
  std::vector<Unit> batch;
  if (waiting_ == 0) return batch;
  for (size_t len = low_; len <= high_; ++len) {
    auto &fifo = fifos_[len];
    while (!fifo.empty()) {
      // all rows are padded to the longest one = the one being added (lengths ascend)
      if ((batch.size() + 1) * len > max_words_) goto done;
      batch.push_back(std::move(fifo.front()));
      fifo.pop_front();
    }
  }
done:
  waiting_ -= batch.size();
  while (low_ <= high_ && fifos_[low_].empty()) ++low_;
  if (waiting_ == 0) {
    low_ = fifos_.size();
    high_ = 0;
  }
  return batch;
}

size_t LengthQueue::peek_length() const {
  if (waiting_ == 0) return 0;
  size_t count = 0, last = 0;
  for (size_t len = low_; len <= high_; ++len) {
    const size_t have = fifos_[len].size();
    if (have == 0) continue;
    // (count + 1) * len <= max_words for every sentence added: at most max_words / len sentences in all
    const size_t room = len ? max_words_ / len : have;
    if (room <= count) break;
    last = len;
    if (count + have > room) break;
    count += have;
  }
  return last;
}

// ---- Service -------------------------------------------------------------------

namespace {
struct Lap {  // adds the time since construction (or the last lap) to a counter
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void to(std::atomic<uint64_t> &counter) {
    const auto now = std::chrono::steady_clock::now();
    counter.fetch_add(static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count()),
                      std::memory_order_relaxed);
    t = now;
  }
};

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
  uint64_t serial = 0;
  // `batch` is the concatenation of the launch's batches (one, or several merged: ServiceConfig::merge_batches), each with
  // its own padded length and its own arrays, one part behind the other in the slot's staging; part j's Hypothesis::batch
  // is serial + j
  struct Part {
    size_t first = 0, B = 0, S = 0, T = 0;         // its sentences in `batch`, its padded length, its row length of tokens
    size_t ids_at = 0, out_at = 0, align_at = 0;  // its arrays in ids / out_ids / align (lengths, out_len: at `first`)
  };
  std::vector<Part> parts;
};

Service::Service(const ServiceConfig &config, std::vector<const Model *> replicas)
    : config_(config),
      // the engine handles sources up to 128 tokens (the reference's wrap length, Frontend.hh:27)
      longest_(std::min<size_t>(config.wrap_length, 128)),
      queue_(config.max_words, longest_) {
  if (replicas.empty()) throw std::invalid_argument("Service needs at least one model replica");
  if (config.workers_per_device == 0) throw std::invalid_argument("Service needs at least one worker");
  config_.lexical_shortlist = View{};  // borrowed for this constructor only
  std::vector<slimt_hip_shortlist *> generator_of(replicas.size(), nullptr);
  if (config.lexical_shortlist.data) {
    std::vector<int> devices;
    for (size_t r = 0; r < replicas.size(); ++r) {
      const int device = replicas[r]->config().device;
      size_t g = 0;
      while (g < devices.size() && devices[g] != device) ++g;
      if (g == devices.size()) {
        devices.push_back(device);
        generators_.push_back(std::make_unique<ShortlistGenerator>(
            config.lexical_shortlist, config.source_vocab, config.target_vocab, config.shortlist_shared_vocab,
            config.shortlist_check, device));
      }
      generator_of[r] = generators_[g]->handle();
    }
  }
  // One stream per context, two contexts per worker: with the HIP runtime's default of four hardware queues most of
  // them would take turns (include/slimt_hip.h, slimt_hip_request_hw_queues). The models exist already, so it is too
  // late to ask for more here; say so once.
  if (config.warn_hw_queues && config.workers_per_device > 2 && slimt_hip_hw_queues() < 8) {
    static std::atomic<bool> warned{false};
    if (!warned.exchange(true))
      std::fprintf(stderr,
                   "slimt::Service: %zu workers per device but GPU_MAX_HW_QUEUES is %d: batches will queue behind each "
                   "other; call slimt_hip_request_hw_queues(32) (or set the variable) before the first HIP call\n",
                   config.workers_per_device, slimt_hip_hw_queues());
  }
  // ... and two contexts per worker are streams too: past 22 contexts on one device in one process the hardware
  // queues are time-sliced whatever that variable says (24: -30 %); the library says so as well when the
  // 23rd context is built, this says it before any is, with the setting that causes it
  if (config.warn_hw_queues) {
    std::vector<std::pair<int, size_t>> per_device;
    for (const Model *m : replicas) {
      auto it = std::find_if(per_device.begin(), per_device.end(), [&](auto &p) { return p.first == m->config().device; });
      if (it == per_device.end()) per_device.emplace_back(m->config().device, 1); else it->second += 1;
    }
    for (auto &[device, n] : per_device) {
      const size_t contexts = 2 * n * config.workers_per_device;
      static std::atomic<bool> warned{false};
      if (contexts > kContextsPerDeviceCliff && !warned.exchange(true))
        std::fprintf(stderr,
                     "slimt::Service: %zu replicas x %zu workers x 2 contexts = %zu streams on device %d: past %zu the "
                     "device's hardware queues are time-sliced (24 measured 30 %% below 20); use at most %zu workers per "
                     "device and process\n",
                     n, config.workers_per_device, contexts, device, kContextsPerDeviceCliff, kContextsPerDeviceCliff / 2);
    }
  }
  live_workers_ = replicas.size() * config.workers_per_device;
  for (size_t r = 0; r < replicas.size(); ++r)
    for (size_t w = 0; w < config.workers_per_device; ++w)
      threads_.emplace_back([this, model = replicas[r], generator = generator_of[r]]() { work(model, generator); });
}

Service::~Service() {
  {
    std::lock_guard<std::mutex> lock(mutex_);
    closing_ = true;
  }
  wake_.notify_all();
  for (auto &t : threads_) t.join();
  if (std::getenv("SLIMT_SERVICE_STATS")) {
    const double n = static_cast<double>(batches_.load() - stats_base_), ms = 1e-6;
    std::fprintf(stderr,
                 "service-stats: %.0f batches, %zu workers; worker time per batch (ms): waiting for work %.3f (the queue's lock %.3f, "
                 "nothing queued %.3f), launch %.3f, waiting for the GPU %.3f, collect %.3f, deliver %.3f\n",
                 n, threads_.size(), ms * ns_idle_ / n, ms * ns_lock_ / n, ms * ns_starved_ / n, ms * ns_launch_ / n, ms * ns_wait_ / n, ms * ns_collect_ / n,
                 ms * ns_deliver_ / n);
    std::fprintf(stderr, "service-stats: merged launches: %llu of %llu launches, %.0f batches\n",
                 static_cast<unsigned long long>(merged_launches_.load() - merged_base_),
                 static_cast<unsigned long long>(launches_.load() - launches_base_), n);
  }
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
    if (live_workers_ == 0) {  // every worker has failed: nobody would ever take this request
      pending->fail(dead_error_);
      return result;
    }
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
  if (config_.alignments && !config_.flat_alignments) {
    // The reference's Alignment is one small vector per target token (slimt/Types.hh:52-57). Built by the workers and freed
    // by the clients, 12,288 of them per batch of 256 cross threads both ways through the allocator (1.2 ms per batch in
    // the workers, the frees behind the arena locks: 20.0 against 28.3 M tok/s). The workers deliver each sentence's rows
    // as ONE block (Hypothesis::alignment_flat) and the rows are cut where the request is collected: a deferred stage of
    // the future, run by the thread that calls get() / wait() -- the thread that will free them. Same type, same values.
    return std::async(std::launch::deferred, [inner = std::move(result)]() mutable {
      Histories histories = inner.get();
      for (History &h : histories) expand_alignment(*h);
      return histories;
    });
  }
  return result;
}

// The next batch, or an empty one: nothing waits (may_block false) / the service closes.
// parts (nullable): merge -- the batches behind the first are taken as well (see ServiceConfig::merge_batches) while the
// launch stays within merge_words and no batch is padded by more than a quarter; their sizes, in order, when more than
// one was taken. Each part keeps the padded length the reference's rule gave it: that of its own last sentence.
std::vector<Unit> Service::next_batch(bool may_block, std::vector<size_t> *parts) {
  Lap lap, part;
  std::unique_lock<std::mutex> lock(mutex_);
  part.to(ns_lock_);  // (of "waiting for work": the queue's lock ...
  if (may_block) wake_.wait(lock, [this]() { return queue_.waiting() > 0 || closing_; });
  part.to(ns_starved_);  // ... and an empty queue)
  std::vector<Unit> batch = queue_.take();
  if (parts) parts->clear();
  if (parts && !batch.empty() && config_.merge_batches > 1) {
    auto rows_of = [](size_t n) { return (n + 31) / 32 * 32; };  // slimt_hip_translate_many_rows
    size_t lo = batch.back().length, hi = lo;  // shortest / longest padded length of the launch
    size_t rows = rows_of(batch.size()), n = 1;
    std::vector<size_t> sizes{batch.size()};
    while (n < config_.merge_batches) {
      const size_t S = queue_.peek_length();
      if (S == 0) break;
      const size_t lo2 = std::min(lo, S), hi2 = std::max(hi, S);
      if (4 * hi2 > 5 * lo2) break;  // the shortest batch would run padded by more than a quarter (encoder rows nobody needs)
      // (the next batch holds at most max_words / S sentences: does the launch still fit with all of them?)
      if ((rows + rows_of(config_.max_words / S)) * hi2 > config_.merge_words) break;
      std::vector<Unit> more = queue_.take();
      if (more.empty()) break;
      lo = std::min<size_t>(lo, more.back().length);
      hi = std::max<size_t>(hi, more.back().length);
      rows += rows_of(more.size());
      sizes.push_back(more.size());
      batch.insert(batch.end(), std::make_move_iterator(more.begin()), std::make_move_iterator(more.end()));
      ++n;
    }
    if (n > 1) *parts = std::move(sizes);
  }
  lap.to(ns_idle_);
  return batch;
}

void Service::launch(Slot &slot, std::vector<Unit> &batch, slimt_hip_shortlist *generator, std::vector<size_t> &parts) {
  Lap lap;
  struct Done {
    Lap &lap;
    std::atomic<uint64_t> &counter;
    ~Done() { lap.to(counter); }
  } done{lap, ns_launch_};
  // the slot owns the batch from here on: whatever throws below, the caller fails slot.batch
  slot.batch = std::move(batch);
  batch.clear();
  if (parts.empty()) parts.push_back(slot.batch.size());
  // every part's arrays, one part behind the other in the slot's staging: [B][S] ids, [B] lengths, [B][T] tokens,
  // [B][T][S] alignment rows with the part's OWN padded length S = that of its last sentence (lengths ascend inside a batch)
  slot.parts.clear();
  size_t first = 0, n_ids = 0, n_out = 0, n_align = 0, launch_S = 0;
  for (size_t count : parts) {
    Slot::Part p;
    p.first = first;
    p.B = count;
    p.S = slot.batch[first + count - 1].length;
    p.T = std::max<size_t>(1, static_cast<size_t>(config_.tgt_length_limit_factor * static_cast<float>(p.S)));
    p.ids_at = n_ids;
    p.out_at = n_out;
    p.align_at = n_align;
    n_ids += p.B * p.S;
    n_out += p.B * p.T;
    n_align += p.B * p.T * p.S;
    n_align = (n_align + 3) / 4 * 4;  // 16-byte aligned blocks: the decoder writes a sentence's rows as whole quads
    launch_S = std::max(launch_S, p.S);
    first += count;
    slot.parts.push_back(p);
  }
  parts.clear();
  const size_t B = slot.batch.size();
  slot.serial = batches_.fetch_add(slot.parts.size(), std::memory_order_relaxed);
  launches_.fetch_add(1, std::memory_order_relaxed);
  if (slot.parts.size() > 1) merged_launches_.fetch_add(1, std::memory_order_relaxed);
  uint32_t *ids = slot.ids.ensure(n_ids);
  uint32_t *lengths = slot.lengths.ensure(B);
  std::fill(ids, ids + n_ids, config_.pad_id);
  for (const Slot::Part &p : slot.parts)
    for (size_t b = 0; b < p.B; ++b) {
      const Unit &u = slot.batch[p.first + b];
      const Words &w = u.owner->sentence(u.index);
      std::copy(w.begin(), w.end(), ids + p.ids_at + b * p.S);
      lengths[p.first + b] = static_cast<uint32_t>(w.size());
    }
  uint32_t *out_ids = slot.out_ids.ensure(n_out), *out_len = slot.out_len.ensure(B);
  float *align = config_.alignments ? slot.align.ensure(n_align) : nullptr;
  const Slot::Part &p0 = slot.parts[0];
  if (generator && slot.parts.size() == 1) {  // the batch's own lexical shortlist, generated on the worker's stream (Model.cc:117-120)
    slot.worker->forward_async_generated(generator, ids, lengths, B, p0.S, config_.tgt_length_limit_factor, out_ids,
                                         out_len, align);
    return;
  }
  const uint32_t *sl = nullptr;
  size_t n_sl = 0;
  if (config_.shortlist) {
    n_sl = config_.shortlist->size();
    sl = slot.shortlist.get();
  }
  if (slot.parts.size() > 1) {  // one launch pair for all parts, each with its own arrays
    std::vector<slimt_hip_batch> calls(slot.parts.size());
    for (size_t j = 0; j < slot.parts.size(); ++j) {
      const Slot::Part &p = slot.parts[j];
      slimt_hip_batch &c = calls[j];
      c.src_ids = ids + p.ids_at;
      c.lengths = lengths + p.first;
      c.B = p.B;
      c.S = p.S;
      c.shortlist = sl;
      c.n_shortlist = n_sl;
      c.out_ids = out_ids + p.out_at;
      c.out_len = out_len + p.first;
      c.align = align ? align + p.align_at : nullptr;
    }
    if (generator)  // every part's own shortlist, generated inside the one encoder launch
      slot.worker->forward_many_async_generated(generator, calls.data(), calls.size(), launch_S, config_.tgt_length_limit_factor);
    else
      slot.worker->forward_many_async(calls.data(), calls.size(), launch_S, config_.tgt_length_limit_factor);
    return;
  }
  slot.worker->forward_async(ids, lengths, B, p0.S, sl, n_sl, config_.tgt_length_limit_factor, out_ids, out_len, align);
}

void Service::finish(Slot &slot) {
  // the batch stays in the slot until its results are in hand: if the wait fails, the caller
  // fails slot.batch with the error (its requests then see the HIP error, not a broken promise)
  Lap lap;
  slot.worker->wait();
  lap.to(ns_wait_);
  std::vector<Histories> histories;
  histories.reserve(slot.parts.size());
  for (const Slot::Part &p : slot.parts)  // (rows cut by the collecting thread: translate())
    histories.push_back(collect(slot.out_ids.get() + p.out_at, slot.out_len.get() + p.first,
                                config_.alignments ? slot.align.get() + p.align_at : nullptr, slot.lengths.get() + p.first, p.B,
                                p.S, p.T, /*flat=*/true));
  lap.to(ns_collect_);
  std::vector<Unit> batch = std::move(slot.batch);
  slot.batch.clear();
  for (size_t j = 0; j < slot.parts.size(); ++j) {
    const Slot::Part &p = slot.parts[j];
    for (size_t b = 0; b < p.B; ++b) {
      histories[j][b]->batch = slot.serial + j;
      batch[p.first + b].owner->deliver(batch[p.first + b].index, std::move(histories[j][b]));
    }
  }
  slot.parts.clear();
  lap.to(ns_deliver_);
}

// A worker that cannot run leaves: the others keep serving. Only when the LAST one has gone do
// the waiting (and all later) requests fail, with that worker's error.
void Service::retire(const std::exception_ptr &error) {
  std::vector<Unit> orphans;
  {
    std::lock_guard<std::mutex> lock(mutex_);
    if (--live_workers_ > 0) return;
    dead_error_ = error;
    for (std::vector<Unit> batch = queue_.take(); !batch.empty(); batch = queue_.take())
      orphans.insert(orphans.end(), std::make_move_iterator(batch.begin()), std::make_move_iterator(batch.end()));
  }
  for (Unit &u : orphans) u.owner->fail(error);
}

void Service::work(const Model *model, slimt_hip_shortlist *generator) {
  Slot slots[2];
  size_t cur = 0;
  auto fail_batch = [](std::vector<Unit> &batch, const std::exception_ptr &error) {
    for (Unit &u : batch) u.owner->fail(error);
    batch.clear();
  };
  try {
    if (config_.fail_worker_setup) config_.fail_worker_setup(model);  // tests: a worker that cannot be built
    for (Slot &s : slots) {
      // (B + 1) * S <= max_words: at most max_words - 1 rows, at most max_words padded tokens; a merged launch: up to
      // merge_words of its 32-aligned rows x length (next_batch)
      const bool merging = config_.merge_batches > 1 && config_.merge_words > config_.max_words;
      const size_t words = merging ? config_.merge_words : config_.max_words;
      s.worker = std::make_unique<Worker>(*model, words, longest_, words);
      // every staging array at its largest, once: growing one later frees and allocates pinned
      // memory, and hipHostFree waits for the whole device (B S <= max_words, T <= factor S + 1)
      const size_t rows = words;
      const size_t out_tokens = static_cast<size_t>(config_.tgt_length_limit_factor * static_cast<float>(rows)) + rows + 1;
      s.ids.ensure(rows);
      s.lengths.ensure(rows);
      s.out_len.ensure(rows);
      s.out_ids.ensure(out_tokens);
      if (config_.alignments) s.align.ensure(out_tokens * longest_);
      if (config_.shortlist && !generator) {
        uint32_t *sl = s.shortlist.ensure(config_.shortlist->size());
        std::copy(config_.shortlist->begin(), config_.shortlist->end(), sl);
      }
    }
  } catch (...) {
    retire(std::current_exception());
    return;
  }
  for (;;) {
    Slot &mine = slots[cur], &other = slots[cur ^ 1];
    // with a batch in flight on the other slot, only take work that is already there
    const bool merging = config_.merge_batches > 1 && config_.merge_words > config_.max_words;
    std::vector<size_t> parts;
    std::vector<Unit> batch = next_batch(other.batch.empty(), merging ? &parts : nullptr);
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
      launch(mine, batch, generator, parts);
    } catch (...) {
      const std::exception_ptr error = std::current_exception();
      fail_batch(batch, error);
      fail_batch(mine.batch, error);
      // a launch that failed half way (a HIP error behind the argument checks) may have queued work that still
      // reads this slot's pinned ids / lengths or writes its outputs: drain the slot's stream before the next
      // batch refills them (whatever the drain itself reports; the batch has failed already)
      try {
        mine.worker->wait();
      } catch (...) {
      }
    }
    if (!other.batch.empty()) {
      try {
        finish(other);  // the previous batch, while `mine` runs on the GPU
      } catch (...) {
        fail_batch(other.batch, std::current_exception());
      }
    }
    if (!mine.batch.empty()) cur ^= 1;
  }
}

}  // namespace slimt
