// Test driver for the C++ host mirror (run by tests/test_gpu_host_cpp.py).
//   host_test <model.bin> <case.bin> <out.bin>
// case.bin: u32 {enc_layers, dec_layers, heads, B, S, n_shortlist, M, K, N},
//           f32 limit_factor, f32 a_quant, f32 b_quant, then u32 ids[B*S],
//           u32 lengths[B], u32 shortlist[n_sl], f32 x[M*K], i8 W[N*K],
//           f32 bias[N], u32 n_idx, u32 idx[n_idx]
// out.bin : per sentence u32 n, u32 tokens[n], f32 align[n][len];
//           then f32 affine[M*N], f32 dot[M*N], f32 select[M*n_idx]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <future>
#include <iostream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "Service.hh"
#include "Io.hh"
#include "Model.hh"
#include "Shortlist.hh"
#include "Transformer.hh"
#include "QMM.hh"

namespace {
std::vector<char> slurp(const char *path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    std::exit(2);
  }
  return std::vector<char>(std::istreambuf_iterator<char>(f), {});
}
struct Cur {
  const char *p;
  template <class T>
  T get() {
    T v;
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  template <class T>
  std::vector<T> vec(size_t n) {
    std::vector<T> v(n);
    std::memcpy(v.data(), p, n * sizeof(T));
    p += n * sizeof(T);
    return v;
  }
};
template <class T>
void put(std::ofstream &o, const T *p, size_t n) {
  o.write(reinterpret_cast<const char *>(p), static_cast<std::streamsize>(n * sizeof(T)));
}
}  // namespace

// Batching modes:
//   host_test --batcher case.bin out.bin            LengthQueue alone (no GPU work): push every
//       sentence, then take() until empty; writes the batches.
//   host_test --async model.bin case.bin out.bin    Service: `workers` double-buffered workers on GPU 0.
// case.bin: u32 {enc_layers, dec_layers, heads, max_words, wrap_length, workers, n_requests},
//           f32 limit_factor, then per request u32 n_segments, per segment u32 len + u32 tokens[len].
//   optional trailing u32 n_shortlist + u32 ids[n] (--async only; absent = full vocabulary)
static int batching_main(int argc, char **argv) {
  using namespace slimt;
  const bool async = std::string(argv[1]) == "--async";
  if (argc != (async ? 5 : 4)) return 2;
  std::vector<char> cs = slurp(argv[async ? 3 : 2]);
  Cur c{cs.data()};
  const uint32_t Le = c.get<uint32_t>(), Ld = c.get<uint32_t>(), H = c.get<uint32_t>();
  const uint32_t max_words = c.get<uint32_t>(), wrap = c.get<uint32_t>(), workers = c.get<uint32_t>();
  const uint32_t n_req = c.get<uint32_t>();
  const float limit = c.get<float>();
  std::vector<std::vector<Words>> requests;
  for (uint32_t r = 0; r < n_req; ++r) {
    const uint32_t n_seg = c.get<uint32_t>();
    std::vector<Words> segs;
    for (uint32_t i = 0; i < n_seg; ++i) {
      const uint32_t len = c.get<uint32_t>();
      segs.push_back(c.vec<uint32_t>(len));
    }
    requests.push_back(std::move(segs));
  }
  std::ofstream out(argv[async ? 4 : 3], std::ios::binary);
  try {
    if (!async) {
      // the longest sentence the reference's batcher provisions for: the wrap length plus the
      // slack for sentences that overflow it (slimt/Batcher.cc:77-92)
      const size_t longest = wrap + static_cast<size_t>(static_cast<float>(wrap) * limit - static_cast<float>(wrap));
      LengthQueue queue(max_words, longest);
      for (uint32_t r = 0; r < n_req; ++r) {
        auto owner = std::make_shared<Pending>(requests[r]);
        for (uint32_t i = 0; i < requests[r].size(); ++i) {
          Unit u;
          u.order = (static_cast<uint64_t>(r) << 24) | i;
          u.owner = owner;
          u.index = i;
          u.length = static_cast<uint32_t>(requests[r][i].size());
          queue.push(std::move(u));
        }
      }
      // (peek_length -- what a worker asks before it merges the NEXT batch into its launch -- must name the padded length
      // of exactly the batch take() then forms)
      size_t peeked = queue.peek_length();
      for (std::vector<Unit> b = queue.take(); !b.empty(); peeked = queue.peek_length(), b = queue.take()) {
        const uint32_t n = static_cast<uint32_t>(b.size()), ml = b.back().length;
        if (peeked != ml) {
          std::fprintf(stderr, "peek_length said %zu, take() formed a batch padded to %u\n", peeked, ml);
          return 1;
        }
        put(out, &n, 1);
        put(out, &ml, 1);
        for (const Unit &u : b) {
          const uint32_t rid = static_cast<uint32_t>(u.order >> 24), idx = u.index;
          put(out, &rid, 1);
          put(out, &idx, 1);
        }
      }
      if (peeked != 0) {
        std::fprintf(stderr, "peek_length said %zu on an empty queue\n", peeked);
        return 1;
      }
      return 0;
    }
    std::vector<char> bin = slurp(argv[2]);
    Model::Config cfg;
    cfg.encoder_layers = Le;
    cfg.decoder_layers = Ld;
    cfg.num_heads = H;
    Model model(cfg, bin.data(), bin.size());
    ServiceConfig sc;
    sc.max_words = max_words;
    sc.wrap_length = wrap;
    sc.tgt_length_limit_factor = limit;
    sc.workers_per_device = workers;
    if (c.p < cs.data() + cs.size()) {
      const uint32_t n_sl = c.get<uint32_t>();
      if (n_sl) sc.shortlist = c.vec<uint32_t>(n_sl);
    }
    if (const char *e = std::getenv("SLIMT_SERVICE_NO_ALIGN")) sc.alignments = e[0] != '1';
    if (const char *e = std::getenv("SLIMT_SERVICE_MERGE")) sc.merge_batches = static_cast<size_t>(std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("SLIMT_SERVICE_MERGE_WORDS")) sc.merge_words = static_cast<size_t>(std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("SLIMT_SERVICE_FLAT_ALIGN")) sc.flat_alignments = e[0] == '1';
    // SLIMT_SERVICE_LEXICAL=<binary shortlist file>: every batch gets its own lexical shortlist,
    // generated on the device (ServiceConfig::lexical_shortlist); the vocabulary sizes are the model's
    std::vector<char> lexical;
    if (const char *e = std::getenv("SLIMT_SERVICE_LEXICAL")) {
      lexical = slurp(e);
      int32_t vocab = 0;
      if (slimt_hip_model_info(model.handle(), nullptr, nullptr, &vocab, nullptr)) throw std::runtime_error("model_info");
      sc.lexical_shortlist = View{lexical.data(), lexical.size()};
      sc.source_vocab = sc.target_vocab = static_cast<size_t>(vocab);
      sc.shortlist_check = true;
    }
    // SLIMT_SERVICE_REPLICAS=n: n model replicas, all on device 0 (exercises the replica / worker
    // assignment without an 8-GPU node); SLIMT_SERVICE_FAIL_WORKERS=k: the first k workers to start
    // fail their set-up (the others must serve everything)
    const int n_replicas = std::getenv("SLIMT_SERVICE_REPLICAS") ? std::atoi(std::getenv("SLIMT_SERVICE_REPLICAS")) : 1;
    std::vector<std::unique_ptr<Model>> extra;
    std::vector<const Model *> replicas{&model};
    for (int r = 1; r < n_replicas; ++r) {
      extra.push_back(std::make_unique<Model>(cfg, bin.data(), bin.size()));
      replicas.push_back(extra.back().get());
    }
    std::atomic<int> to_fail{std::getenv("SLIMT_SERVICE_FAIL_WORKERS") ? std::atoi(std::getenv("SLIMT_SERVICE_FAIL_WORKERS")) : 0};
    if (to_fail.load() > 0)
      sc.fail_worker_setup = [&to_fail](const Model *) {
        if (to_fail.fetch_sub(1) > 0) throw std::runtime_error("injected worker set-up failure");
      };
    const bool dump_full = std::getenv("SLIMT_SERVICE_DUMP_FULL") != nullptr;
    std::vector<std::future<Histories>> futures;
    std::vector<Histories> results;
    {
      Service service(sc, replicas);
      // first request alone and untimed: worker start-up (contexts, pinned buffers)
      const auto t0 = std::chrono::steady_clock::now();
      for (auto &r : requests) futures.push_back(service.translate(r));
      if (std::getenv("SLIMT_SERVICE_DISCARD")) {
        // a client that consumes results as they arrive (benchmarks): count and drop, so that the
        // allocator recycles the memory instead of faulting in 1.5 MB of fresh pages per batch
        size_t tokens = 0;
        for (auto &f : futures)
          for (const auto &h : f.get()) tokens += h->target.size();
        std::fprintf(stderr, "async-tokens: %zu\n", tokens);
      } else {
        for (auto &f : futures) results.push_back(f.get());
      }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::fprintf(stderr, "async: %zu requests translated in %.3f ms\n", requests.size(), ms);
      if (std::getenv("SLIMT_SERVICE_REPEAT")) {  // steady state: the same requests again, workers warm
        std::vector<std::future<Histories>> again;
        service.stats_reset();
        const auto t1 = std::chrono::steady_clock::now();
        if (std::getenv("SLIMT_SERVICE_DISCARD")) {
          // clients that consume (and free) their results as they arrive: request i belongs to client
          // i mod C; one client alone would be the bottleneck once alignment rows come back (45 small
          // vectors per sentence, Types.hh:34-36)
          const size_t C = std::getenv("SLIMT_SERVICE_CLIENTS") ? std::max(1, std::atoi(std::getenv("SLIMT_SERVICE_CLIENTS"))) : 4;
          std::atomic<size_t> warm_tokens{0};
          std::vector<std::thread> clients;
          for (size_t c = 0; c < C; ++c)
            clients.emplace_back([&, c]() {
              // steady state: at most `window` requests outstanding per client; the oldest is consumed
              // (and freed) before the next one is submitted; `rounds` passes over the client's requests
              const size_t window = std::getenv("SLIMT_SERVICE_WINDOW") ? std::max(1, std::atoi(std::getenv("SLIMT_SERVICE_WINDOW"))) : 8;
              const size_t rounds = std::getenv("SLIMT_SERVICE_ROUNDS") ? std::max(1, std::atoi(std::getenv("SLIMT_SERVICE_ROUNDS"))) : 3;
              std::deque<std::future<Histories>> mine;
              size_t tokens = 0;
              auto consume = [&]() {
                for (const auto &h : mine.front().get()) tokens += h->target.size();
                mine.pop_front();
              };
              for (size_t r = 0; r < rounds; ++r)
                for (size_t i = c; i < requests.size(); i += C) {
                  if (mine.size() >= window) consume();
                  mine.push_back(service.translate(requests[i]));
                }
              while (!mine.empty()) consume();
              warm_tokens += tokens;
            });
          for (auto &t : clients) t.join();
          std::fprintf(stderr, "async-warm-tokens: %zu\n", warm_tokens.load());
        } else {
          for (auto &r : requests) again.push_back(service.translate(r));
          for (auto &f : again) f.wait();
        }
        const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        std::fprintf(stderr, "async-warm: %zu requests translated in %.3f ms\n", requests.size(), ms2);
      }
    }  // drains and joins the workers
    for (const Histories &hs : results) {
      for (const auto &h : hs) {
        const uint32_t S = static_cast<uint32_t>(h->padded_length), n = static_cast<uint32_t>(h->target.size());
        put(out, &S, 1);
        put(out, &n, 1);
        if (dump_full) {  // + the batch it travelled in and its alignment rows
          const uint32_t batch = static_cast<uint32_t>(h->batch);
          const uint32_t rows = static_cast<uint32_t>(h->alignment.size());
          const uint32_t len = rows ? static_cast<uint32_t>(h->alignment[0].size()) : 0;
          put(out, &batch, 1);
          put(out, &rows, 1);
          put(out, &len, 1);
        }
        put(out, h->target.data(), n);
        if (dump_full)
          for (const Distribution &row : h->alignment) put(out, row.data(), row.size());
      }
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "host_test: %s\n", e.what());
    return 1;
  }
  return 0;
}

// host_test --service-errors model.bin Le Ld H: bad requests are refused or reported, never fatal.
static int service_errors_main(int argc, char **argv) {
  using namespace slimt;
  if (argc != 6) return 2;
  try {
    std::vector<char> bin = slurp(argv[2]);
    Model::Config cfg;
    cfg.encoder_layers = static_cast<size_t>(std::atoi(argv[3]));
    cfg.decoder_layers = static_cast<size_t>(std::atoi(argv[4]));
    cfg.num_heads = static_cast<size_t>(std::atoi(argv[5]));
    Model model(cfg, bin.data(), bin.size());
    int32_t vocab = 0;
    if (slimt_hip_model_info(model.handle(), nullptr, nullptr, &vocab, nullptr)) throw std::runtime_error("model_info");
    ServiceConfig sc;
    sc.max_words = 96;
    sc.wrap_length = 24;
    sc.workers_per_device = 2;
    Service service(sc, {&model});
    try {
      service.translate({Words{}});
      std::printf("accepted empty\n");
    } catch (const std::invalid_argument &) {
      std::printf("rejected empty\n");
    }
    try {
      service.translate({Words(25, 1)});
      std::printf("accepted overlong\n");
    } catch (const std::invalid_argument &) {
      std::printf("rejected overlong\n");
    }
    // a token id outside the vocabulary: the engine refuses the batch on the worker; the
    // error must come back through the future (and the innocent request in the same batch too)
    auto bad = service.translate({Words{1, 2, static_cast<Word>(vocab), 0}});
    try {
      bad.get();
      std::printf("worker failure lost\n");
    } catch (const std::runtime_error &e) {
      std::printf("worker failure reported: %s\n", e.what());
    }
    auto good = service.translate({Words{5, 6, 0}, Words{7, 0}, Words{9, 10, 11, 12, 0}});
    Histories hs = good.get();
    size_t ok = 0;
    for (const auto &h : hs) ok += (h && !h->target.empty()) ? 1 : 0;
    std::printf("survived: %zu sentences\n", ok);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "host_test: %s\n", e.what());
    return 1;
  }
  return 0;
}

// host_test --transformer model.bin case.bin out.bin
// The class-level mirror (host/Transformer.hh) driven the way slimt::Model::forward and
// Model::decode drive the reference's classes (Model.cc:111-204): index_select +
// transform_embedding, Encoder::forward, Decoder::start_states, then Decoder::step +
// greedy_sample* per step with the EOS bookkeeping. case.bin as in the default mode.
// out.bin: f32 encoder_out[B*S*D]; per sentence u32 n, u32 tokens[n], f32 align[n][len].
static int transformer_main(int argc, char **argv) {
  using namespace slimt;
  if (argc != 5) return 2;
  std::vector<char> bin = slurp(argv[2]), cs = slurp(argv[3]);
  Cur c{cs.data()};
  const uint32_t Le = c.get<uint32_t>(), Ld = c.get<uint32_t>(), H = c.get<uint32_t>();
  const uint32_t B = c.get<uint32_t>(), S = c.get<uint32_t>(), n_sl = c.get<uint32_t>();
  c.get<uint32_t>(); c.get<uint32_t>(); c.get<uint32_t>();  // M, K, N of the qmm part: unused here
  const float limit = c.get<float>();
  c.get<float>(); c.get<float>();
  auto ids = c.vec<uint32_t>(size_t(B) * S);
  auto lens = c.vec<uint32_t>(B);
  auto sl = c.vec<uint32_t>(n_sl);
  std::ofstream out(argv[4], std::ios::binary);
  try {
    Transformer transformer(Le, Ld, H, /*feed_forward_depth=*/0, View{bin.data(), bin.size()});
    const Vocabulary vocabulary(transformer.vocab(), /*eos_id=*/0, /*pad_id=*/0);
    // Input (Input.cc:20-63): indices [B,S] and the additive mask
    Tensor indices(Type::u32, Shape({B, S}), "indices");
    std::memcpy(indices.data<uint32_t>(), ids.data(), ids.size() * sizeof(uint32_t));
    Tensor mask(Type::f32, Shape({B, S}), "mask");
    for (uint32_t b = 0; b < B; ++b)
      for (uint32_t j = 0; j < S; ++j) mask.data<float>()[size_t(b) * S + j] = j < lens[b] ? 0.0F : -99999999.0F;
    // Model::forward (Model.cc:187-204)
    Tensor word_embedding = index_select(transformer.embedding(), indices, "word_embedding");
    transform_embedding(word_embedding);
    Tensor encoder_out = transformer.encoder().forward(word_embedding, mask);
    put(out, encoder_out.data<float>(), encoder_out.size());
    // Model::decode (Model.cc:111-185)
    std::optional<Words> shortlist;
    if (n_sl) shortlist = sl;
    std::vector<bool> complete(B, false);
    std::vector<Words> sentences(B);
    std::vector<std::vector<std::vector<float>>> alignments(B);
    const Decoder &decoder = transformer.decoder();
    std::vector<Tensor> states = decoder.start_states(B);
    Words previous;
    const size_t max_steps = static_cast<size_t>(limit * static_cast<float>(S));
    size_t remaining = B;
    for (size_t i = 0; i == 0 || (i < max_steps && remaining > 0); ++i) {
      auto [logits, attn] = decoder.step(encoder_out, mask, states, previous, shortlist);
      previous = shortlist ? greedy_sample_from_words(logits, vocabulary, *shortlist, B)
                           : greedy_sample(logits, vocabulary, B);
      size_t finished = 0;
      for (uint32_t b = 0; b < B; ++b) {
        if (!complete[b]) {  // update_alignment, then record (Model.cc:84-108,127-137)
          const float *row = attn.data<float>() + size_t(b) * H * S;  // head 0
          alignments[b].emplace_back(row, row + lens[b]);
          complete[b] = previous[b] == vocabulary.eos_id();
          sentences[b].push_back(previous[b]);
        }
        finished += complete[b] ? 1 : 0;
      }
      remaining = B - finished;
    }
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t n = static_cast<uint32_t>(sentences[b].size());
      put(out, &n, 1);
      put(out, sentences[b].data(), n);
      for (const auto &row : alignments[b]) put(out, row.data(), row.size());
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "host_test: %s\n", e.what());
    return 1;
  }
  return 0;
}

// host_test --load model.bin: io::load_items alone (no GPU): prints the item count or the error.
static int load_main(int argc, char **argv) {
  if (argc != 3) return 2;
  std::vector<char> bin = slurp(argv[2]);
  try {
    std::vector<slimt::io::Item> items = slimt::io::load_items(bin.data(), bin.size());
    std::printf("loaded %zu items\n", items.size());
  } catch (const std::exception &e) {
    std::fprintf(stderr, "host_test: %s\n", e.what());
    return 1;
  }
  return 0;
}

// host_test --model-forward model.bin case.bin out.bin [lexical_shortlist.bin]
//   slimt::Model::forward(const Input &) const (slimt/Model.hh:56) called from `threads` threads on ONE
//   const Model, batch i by thread i % threads. case.bin: u32 {enc_layers, dec_layers, heads, threads,
//   batches}, f32 limit_factor, then per batch u32 B, u32 S, u32 ids[B*S], u32 lengths[B].
//   out.bin: per batch, per sentence u32 n, u32 tokens[n], f32 align[n][len]. stderr: contexts built.
static int model_forward_main(int argc, char **argv) {
  using namespace slimt;
  if (argc != 5 && argc != 6) return 2;
  std::vector<char> bin = slurp(argv[2]), cs = slurp(argv[3]);
  std::vector<char> blob;
  if (argc == 6) blob = slurp(argv[5]);
  Cur c{cs.data()};
  const uint32_t Le = c.get<uint32_t>(), Ld = c.get<uint32_t>(), H = c.get<uint32_t>();
  const uint32_t n_threads = c.get<uint32_t>(), n_batches = c.get<uint32_t>();
  const float limit = c.get<float>();
  std::vector<Input> inputs;
  for (uint32_t i = 0; i < n_batches; ++i) {
    const uint32_t B = c.get<uint32_t>(), S = c.get<uint32_t>();
    auto ids = c.vec<uint32_t>(size_t(B) * S);
    auto lens = c.vec<uint32_t>(B);
    inputs.emplace_back(B, S, /*pad_id=*/0, limit);
    for (uint32_t b = 0; b < B; ++b)
      inputs.back().add(Words(ids.begin() + size_t(b) * S, ids.begin() + size_t(b) * S + lens[b]));
  }
  try {
    Model::Config cfg;
    cfg.encoder_layers = Le;
    cfg.decoder_layers = Ld;
    cfg.num_heads = H;
    const Model model(cfg, bin.data(), bin.size(), blob.empty() ? nullptr : blob.data(), blob.size());
    std::vector<Histories> results(n_batches);
    std::vector<std::string> errors(n_threads);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; ++t)
      pool.emplace_back([&, t] {
        try {
          for (uint32_t i = t; i < n_batches; i += n_threads) results[i] = model.forward(inputs[i]);
        } catch (const std::exception &e) {
          errors[t] = e.what();
        }
      });
    for (auto &th : pool) th.join();
    for (const std::string &e : errors)
      if (!e.empty()) throw std::runtime_error(e);
    std::fprintf(stderr, "contexts-built: %zu\n", model.contexts_built());
    std::ofstream out(argv[4], std::ios::binary);
    for (uint32_t i = 0; i < n_batches; ++i)
      for (const auto &h : results[i]) {
        const uint32_t n = static_cast<uint32_t>(h->target.size());
        put(out, &n, 1);
        put(out, h->target.data(), n);
        for (const auto &row : h->alignment) put(out, row.data(), row.size());
      }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "host_test: %s\n", e.what());
    return 1;
  }
  return 0;
}

int main(int argc, char **argv) {
  slimt_hip_request_hw_queues(32);  // before the first HIP call of the process (include/slimt_hip.h)
  if (argc >= 2 && std::string(argv[1]) == "--model-forward") return model_forward_main(argc, argv);
  if (argc >= 2 && (std::string(argv[1]) == "--batcher" || std::string(argv[1]) == "--async"))
    return batching_main(argc, argv);
  if (argc >= 2 && std::string(argv[1]) == "--service-errors") return service_errors_main(argc, argv);
  if (argc >= 2 && std::string(argv[1]) == "--load") return load_main(argc, argv);
  if (argc >= 2 && std::string(argv[1]) == "--transformer") return transformer_main(argc, argv);
  if (argc != 4 && argc != 5) {
    std::fprintf(stderr, "usage: %s model.bin case.bin out.bin [lexical_shortlist.bin]\n", argv[0]);
    return 2;
  }
  using namespace slimt;
  std::vector<char> bin = slurp(argv[1]), cs = slurp(argv[2]);
  Cur c{cs.data()};
  const uint32_t Le = c.get<uint32_t>(), Ld = c.get<uint32_t>(), H = c.get<uint32_t>();
  const uint32_t B = c.get<uint32_t>(), S = c.get<uint32_t>(), n_sl = c.get<uint32_t>();
  const uint32_t M = c.get<uint32_t>(), K = c.get<uint32_t>(), N = c.get<uint32_t>();
  const float limit = c.get<float>(), aq = c.get<float>(), bq = c.get<float>();
  auto ids = c.vec<uint32_t>(size_t(B) * S);
  auto lens = c.vec<uint32_t>(B);
  auto sl = c.vec<uint32_t>(n_sl);
  auto x = c.vec<float>(size_t(M) * K);
  auto W = c.vec<int8_t>(size_t(N) * K);
  auto bias = c.vec<float>(N);
  const uint32_t n_idx = c.get<uint32_t>();
  auto idx = c.vec<uint32_t>(n_idx);
  std::ofstream out(argv[3], std::ios::binary);
  try {
    Model::Config cfg;
    cfg.encoder_layers = Le;
    cfg.decoder_layers = Ld;
    cfg.num_heads = H;
    Model model(cfg, bin.data(), bin.size());
    Worker worker(model, B, S);
    Input input(B, S, /*pad_id=*/0, limit);
    for (uint32_t b = 0; b < B; ++b)
      input.add(Words(ids.begin() + size_t(b) * S, ids.begin() + size_t(b) * S + lens[b]));
    std::optional<Words> shortlist;
    if (n_sl) shortlist = sl;
    if (argc == 5) {  // Model::forward's order: generate the batch's shortlist first (Model.cc:117-120)
      std::vector<char> blob = slurp(argv[4]);
      int32_t vocab = 0;
      if (slimt_hip_model_info(model.handle(), nullptr, nullptr, &vocab, nullptr)) throw std::runtime_error("model_info");
      const size_t V = static_cast<size_t>(vocab);
      ShortlistGenerator generator(View{blob.data(), blob.size()}, V, V);
      Shortlist generated = generator.generate(input.words());
      shortlist = generated.words();
      const uint32_t n = static_cast<uint32_t>(generated.words().size());
      put(out, &n, 1);
      put(out, generated.words().data(), n);
    }
    Histories hs = worker.forward(input, shortlist, true);
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t n = static_cast<uint32_t>(hs[b]->target.size());
      put(out, &n, 1);
      put(out, hs[b]->target.data(), n);
      for (const auto &row : hs[b]->alignment) put(out, row.data(), row.size());
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "host_test: %s\n", e.what());
    return 1;
  }
  // slimt::qmm through the provider facade, reference-style tensors
  Tensor tx(Type::f32, Shape({M, K}), "x");
  std::memcpy(tx.data<float>(), x.data(), x.size() * sizeof(float));
  Tensor tW(Type::i8, Shape({K, N}), "W", sizeof(float));  // + trailing b_quant
  qmm::prepare_weight_quantized_transposed(W.data(), tW.data<int8_t>(), K, N);
  std::memcpy(tW.data<int8_t>() + size_t(K) * N, &bq, sizeof(float));
  Tensor tb(Type::f32, Shape({1, N}), "b");
  std::memcpy(tb.data<float>(), bias.data(), bias.size() * sizeof(float));
  const float b_quant = *reinterpret_cast<const float *>(tW.end<int8_t>());  // Modules.cc:18-22
  Tensor y1 = qmm::affine(tx, tW, tb, aq, b_quant, "y");
  Tensor y2 = qmm::dot(tx, tW, aq, b_quant, "y");
  Tensor y3 = qmm::affine_with_select(tx, tW, tb, aq, b_quant, idx, "logits");
  put(out, y1.data<float>(), y1.size());
  put(out, y2.data<float>(), y2.size());
  put(out, y3.data<float>(), y3.size());
  return 0;
}
