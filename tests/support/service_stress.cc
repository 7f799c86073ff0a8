// CPU stress of host/Service over the fake engine (fake_hip_engine.cc): many client threads, many
// workers on two "replicas", every sentence checked, then the failure modes. Built with
// -fsanitize=thread or address by tests/test_service_sanitizers.py.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <thread>

#include "Service.hh"

using namespace slimt;

static bool check(const Words &src, const Hypothesis &h, float limit, bool with_align) {
  const size_t S = h.padded_length, T = std::max<size_t>(1, (size_t)(limit * (float)S));
  Words want;
  for (size_t t = 0; t + 1 < src.size() && want.size() + 1 < T; ++t) want.push_back(src[src.size() - 2 - t]);
  want.push_back(0);
  if (h.target != want) return false;
  if (!with_align) return h.alignment.empty();
  if (h.alignment.size() != want.size()) return false;
  for (const auto &row : h.alignment) {
    if (row.size() != src.size()) return false;
    float sum = 0;
    for (float v : row) sum += v;
    if (sum != 1.0f) return false;
  }
  return true;
}

int main(int argc, char **argv) {
  const int clients = argc > 1 ? std::atoi(argv[1]) : 4;
  const int requests = argc > 2 ? std::atoi(argv[2]) : 30;
  Model::Config mc;
  slimt_hip_dims dims{1, 1, 2};
  auto fake_model = [&]() {  // the fake engine has no weights: borrowed handles (Model's view constructor)
    slimt_hip_model *h = nullptr;
    if (slimt_hip_model_create(nullptr, 0, &dims, 0, &h)) std::abort();
    return h;
  };
  slimt_hip_model *ha = fake_model(), *hb = fake_model(), *h2 = nullptr, *h3 = nullptr;
  Model a(mc, ha), b(mc, hb);
  std::atomic<int> bad{0}, done{0};
  for (int with_align = 0; with_align < 2; ++with_align) {
    ServiceConfig sc;
    sc.max_words = 64;
    sc.wrap_length = 16;
    sc.workers_per_device = 3;
    sc.alignments = with_align != 0;
    Service service(sc, {&a, &b});
    std::vector<std::thread> ts;
    for (int c = 0; c < clients; ++c)
      ts.emplace_back([&, c]() {
        std::mt19937 rng(100 + c);
        std::vector<std::pair<std::vector<Words>, std::future<Histories>>> inflight;
        auto drain = [&]() {
          auto &p = inflight.front();
          Histories hs = p.second.get();
          for (size_t i = 0; i < hs.size(); ++i)
            if (!check(p.first[i], *hs[i], sc.tgt_length_limit_factor, sc.alignments)) ++bad;
          done += (int)hs.size();
          inflight.erase(inflight.begin());
        };
        for (int r = 0; r < requests; ++r) {
          std::vector<Words> sents(1 + rng() % 9);
          for (auto &s : sents) {
            s.resize(1 + rng() % 16);
            for (auto &w : s) w = 2 + rng() % 500;
            s.back() = 0;
          }
          auto fut = service.translate(sents);
          inflight.emplace_back(std::move(sents), std::move(fut));
          if (inflight.size() > 4) drain();
        }
        while (!inflight.empty()) drain();
      });
    for (auto &t : ts) t.join();
  }
  std::printf("translated %d sentences, %d wrong\n", done.load(), bad.load());
  // refused requests; a failing batch reaches its requests and the service keeps going
  {
    ServiceConfig sc;
    sc.max_words = 64;
    sc.wrap_length = 16;
    sc.workers_per_device = 2;
    Service service(sc, {&a});
    try { service.translate({Words{}}); std::printf("accepted empty\n"); } catch (const std::invalid_argument &) { std::printf("rejected empty\n"); }
    auto poisoned = service.translate({Words{1, 2, 600, 0}});
    try { poisoned.get(); std::printf("bad token accepted\n"); } catch (const std::exception &e) { std::printf("engine failure reported: %s\n", e.what()); }
    Histories ok = service.translate({Words{5, 6, 0}, Words{7, 0}}).get();
    std::printf("survived: %zu sentences\n", ok.size());
  }
  // workers that cannot be built retire; with none left, requests fail instead of hanging
  {
    setenv("FAKE_HIP_FAIL_CTX_AFTER", "4", 1);  // two of four workers get their two contexts
    h2 = fake_model();
    Model m2(mc, h2);
    ServiceConfig sc;
    sc.max_words = 64;
    sc.wrap_length = 16;
    sc.workers_per_device = 4;
    Service service(sc, {&m2});
    size_t n = 0;
    for (int r = 0; r < 20; ++r) n += service.translate({Words{3, 4, 5, 0}, Words{9, 0}}).get().size();
    std::printf("half the workers retired: %zu sentences translated\n", n);
    setenv("FAKE_HIP_FAIL_CTX_AFTER", "0", 1);
    h3 = fake_model();
    Model m3(mc, h3);
    Service dead(sc, {&m3});
    try { dead.translate({Words{3, 0}}).get(); std::printf("dead service answered\n"); } catch (const std::exception &e) { std::printf("dead service: %s\n", e.what()); }
    unsetenv("FAKE_HIP_FAIL_CTX_AFTER");
  }
  // Model::forward(const Input &) const from several threads on one Model: the context pool
  {
    std::vector<std::thread> ts;
    std::atomic<int> fwd{0};
    for (int c = 0; c < clients; ++c)
      ts.emplace_back([&, c]() {
        std::mt19937 rng(700 + c);
        for (int r = 0; r < requests; ++r) {
          const size_t B = 1 + rng() % 6, S = 2 + rng() % 15;
          Input in(B, S, 0, 1.5f);
          std::vector<Words> sents(B);
          for (auto &s : sents) {
            s.resize(1 + rng() % S);
            for (auto &w : s) w = 2 + rng() % 500;
            s.back() = 0;
            in.add(s);
          }
          Histories hs = static_cast<const Model &>(a).forward(in);
          for (size_t i = 0; i < hs.size(); ++i)
            if (!check(sents[i], *hs[i], 1.5f, true)) ++bad;
          fwd += (int)hs.size();
        }
      });
    for (auto &t : ts) t.join();
    std::printf("Model::forward: %d sentences on %zu pooled contexts\n", fwd.load(), a.contexts_built());
  }
  for (slimt_hip_model *h : {ha, hb, h2, h3}) slimt_hip_model_destroy(h);
  return bad.load() ? 1 : 0;
}
