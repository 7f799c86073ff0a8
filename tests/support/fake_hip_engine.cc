// TEST DOUBLE of the engine-level C ABI (include/slimt_hip.h), CPU only: just enough for
// host/Service.{hh,cc} to run WITHOUT a GPU, so that its threading (queue, double-buffered workers,
// retiring workers, futures) can be exercised under ThreadSanitizer / AddressSanitizer in the CPU
// test suite. Not the product and never linked into it: the "translation" of a sentence is its own
// tokens reversed (EOS kept last), capped at max(1, floor(limit_factor * S)); an alignment row t puts
// weight 1 on source position (len - 2 - t) (or the EOS column). A call completes on a helper thread
// after a short sleep, like a kernel would on a stream.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>
#include <vector>

#include "slimt_hip.h"

struct slimt_hip_model { int device = 0; int heads = 2; std::atomic<int> contexts{0}; };
struct slimt_hip_ctx { slimt_hip_model *model; size_t max_B, max_S, max_M; std::future<void> pending; };
struct slimt_hip_shortlist { int device = 0; };

static thread_local char g_err[256] = "";
static int fail(const char *m) { std::strncpy(g_err, m, sizeof(g_err) - 1); return -1; }
static std::atomic<long> g_host_allocs{0};

extern "C" {
const char *slimt_hip_last_error(void) { return g_err; }
int slimt_hip_model_create(const slimt_hip_param *, size_t, const slimt_hip_dims *dims, int device, slimt_hip_model **out) {
  auto *m = new slimt_hip_model();
  m->device = device;
  if (dims) m->heads = dims->num_heads;
  *out = m;
  return 0;
}
int slimt_hip_model_destroy(slimt_hip_model *m) { delete m; return 0; }
int slimt_hip_model_device(const slimt_hip_model *m) { return m ? m->device : -1; }
int slimt_hip_hw_queues(void) { return 32; }
int slimt_hip_request_hw_queues(int) { return 0; }
int slimt_hip_model_info(const slimt_hip_model *m, int32_t *d, int32_t *f, int32_t *v, int32_t *h) {
  if (!m) return fail("model is NULL");
  if (d) *d = 64; if (f) *f = 128; if (v) *v = 512; if (h) *h = m->heads;
  return 0;
}
int slimt_hip_ctx_create_budget(slimt_hip_model *model, size_t max_batch, size_t max_len, size_t max_tokens, void *,
                                slimt_hip_ctx **out) {
  if (!model) return fail("model is NULL");
  if (std::getenv("FAKE_HIP_FAIL_CTX_AFTER")) {  // the Nth context and every later one cannot be built
    if (model->contexts.fetch_add(1) >= std::atoi(std::getenv("FAKE_HIP_FAIL_CTX_AFTER"))) return fail("fake: out of device memory");
  }
  *out = new slimt_hip_ctx{model, max_batch, max_len, max_tokens, {}};
  return 0;
}
int slimt_hip_ctx_destroy(slimt_hip_ctx *c) {
  if (c && c->pending.valid()) c->pending.wait();
  delete c;
  return 0;
}
int slimt_hip_ctx_synchronize(slimt_hip_ctx *c) {
  if (!c) return fail("ctx is NULL");
  if (c->pending.valid()) c->pending.get();
  return 0;
}
int slimt_hip_host_alloc(size_t bytes, void **out) {
  *out = std::malloc(bytes ? bytes : 1);
  g_host_allocs += 1;
  return *out ? 0 : fail("malloc");
}
int slimt_hip_host_free(void *p) { std::free(p); return 0; }

// one batch's "translation": every sentence's tokens reversed (+ EOS), its alignment rows one-hot on the token they came from
static void fake_translate(const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S, size_t T, uint32_t eos,
                           uint32_t *out_ids, uint32_t *out_len, float *align) {
  for (size_t b = 0; b < B; ++b) {
    const size_t len = lengths[b];
    size_t n = 0;
    for (size_t t = 0; t + 1 < len && n + 1 < T; ++t) out_ids[b * T + n++] = ids[b * S + (len - 2 - t)];
    out_ids[b * T + n++] = eos;
    for (size_t t = n; t < T; ++t) out_ids[b * T + t] = 0;
    out_len[b] = (uint32_t)n;
    if (align) {
      std::fill(align + b * T * S, align + (b + 1) * T * S, 0.0f);
      for (size_t t = 0; t < n; ++t) align[(b * T + t) * S + (t + 1 < n && len >= 2 + t ? len - 2 - t : len - 1)] = 1.0f;
    }
  }
}

static int check_one(slimt_hip_ctx *c, const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S, const uint32_t *out_ids,
                     const uint32_t *out_len) {
  if (!c || !ids || !lengths || !out_ids || !out_len) return fail("null argument");
  if (B == 0 || S == 0 || B > c->max_B || S > c->max_S || B * S > c->max_M) return fail("batch exceeds the context's workspace");
  for (size_t b = 0; b < B; ++b)
    for (size_t j = 0; j < lengths[b]; ++j)
      if (ids[b * S + j] >= 512) return fail("token id out of range");
  return 0;
}

static int run(slimt_hip_ctx *c, const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S, float limit, uint32_t eos,
               uint32_t *out_ids, uint32_t *out_len, float *align, bool wait) {
  if (int rc = check_one(c, ids, lengths, B, S, out_ids, out_len)) return rc;
  const size_t T = std::max<size_t>(1, (size_t)(limit * (float)S));
  c->pending = std::async(std::launch::async, [=]() {
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 13 * (B % 7)));
    fake_translate(ids, lengths, B, S, T, eos, out_ids, out_len, align);
  });
  if (wait) c->pending.get();
  return 0;
}

// several batches in one "launch pair" (include/slimt_hip.h, slimt_hip_translate_many_async): each with its own padded
// length, arrays and results, all of them completed by the one asynchronous task
static int run_many(slimt_hip_ctx *c, const slimt_hip_batch *batches, size_t n, size_t S, float limit, uint32_t eos) {
  if (!c || !batches || n == 0) return fail("null argument");
  std::vector<slimt_hip_batch> copy(batches, batches + n);
  size_t rows = 0;
  for (slimt_hip_batch &b : copy) {
    if (b.S == 0) b.S = S;
    if (b.S > S) return fail("a batch is padded to more tokens than the launch");
    if (int rc = check_one(c, b.src_ids, b.lengths, b.B, b.S, b.out_ids, b.out_len)) return rc;
    rows += b.B;
  }
  if (rows > c->max_B || rows * S > c->max_M) return fail("merged batches exceed the context's workspace");
  c->pending = std::async(std::launch::async, [copy, limit, eos]() {
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 13 * (copy.size() % 7)));
    for (const slimt_hip_batch &b : copy)
      fake_translate(b.src_ids, b.lengths, b.B, b.S, std::max<size_t>(1, (size_t)(limit * (float)b.S)), eos, b.out_ids, b.out_len,
                     b.align);
  });
  return 0;
}
int slimt_hip_translate_many_async(slimt_hip_ctx *c, const slimt_hip_batch *batches, size_t n, size_t S, float limit, uint32_t eos) {
  return run_many(c, batches, n, S, limit, eos);
}
int slimt_hip_translate_many_async_generated(slimt_hip_ctx *c, slimt_hip_shortlist *sl, const slimt_hip_batch *batches, size_t n,
                                             size_t S, float limit, uint32_t eos) {
  if (!sl) return fail("shortlist is NULL");
  return run_many(c, batches, n, S, limit, eos);
}
int slimt_hip_translate(slimt_hip_ctx *c, const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S, const uint32_t *,
                        size_t, float limit, uint32_t eos, uint32_t *out_ids, uint32_t *out_len, float *align) {
  return run(c, ids, lengths, B, S, limit, eos, out_ids, out_len, align, true);
}
int slimt_hip_translate_async(slimt_hip_ctx *c, const uint32_t *ids, const uint32_t *lengths, size_t B, size_t S,
                              const uint32_t *, size_t, float limit, uint32_t eos, uint32_t *out_ids, uint32_t *out_len,
                              float *align) {
  return run(c, ids, lengths, B, S, limit, eos, out_ids, out_len, align, false);
}
int slimt_hip_shortlist_create(const void *, size_t, size_t, size_t, int, int, int device, slimt_hip_shortlist **out) {
  *out = new slimt_hip_shortlist{device};
  return 0;
}
int slimt_hip_shortlist_destroy(slimt_hip_shortlist *s) { delete s; return 0; }
int slimt_hip_shortlist_generate(slimt_hip_shortlist *, const uint32_t *, const uint32_t *, size_t, size_t, uint32_t *out, size_t *n) {
  for (uint32_t i = 0; i < 8; ++i) out[i] = i;
  *n = 8;
  return 0;
}
int slimt_hip_translate_generated(slimt_hip_ctx *c, slimt_hip_shortlist *sl, const uint32_t *ids, const uint32_t *lengths, size_t B,
                                  size_t S, float limit, uint32_t eos, uint32_t *out_ids, uint32_t *out_len, float *align) {
  if (!sl) return fail("shortlist is NULL");
  return run(c, ids, lengths, B, S, limit, eos, out_ids, out_len, align, true);
}
int slimt_hip_translate_async_generated(slimt_hip_ctx *c, slimt_hip_shortlist *sl, const uint32_t *ids, const uint32_t *lengths,
                                        size_t B, size_t S, float limit, uint32_t eos, uint32_t *out_ids, uint32_t *out_len,
                                        float *align) {
  if (!sl) return fail("shortlist is NULL");
  return run(c, ids, lengths, B, S, limit, eos, out_ids, out_len, align, false);
}
}
