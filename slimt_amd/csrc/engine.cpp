// C ABI (include/slimt_hip.h) + host-side engine of the MI355X slimt backend.
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

using namespace slimt_hip;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code ? code : -1;
}

// Set on this library's first call into the HIP runtime (every path to the device goes through one of
// DevBuf::reserve, slimt_hip_device_count, slimt_hip_set_device, slimt_hip_host_alloc): the runtime reads
// GPU_MAX_HW_QUEUES once, when it initialises, so slimt_hip_request_hw_queues is of no use afterwards.
static std::atomic<bool> g_hip_called{false};

#define HIPCHK(expr)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail((int)e_, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
  } while (0)

#define RCCHK(expr)        \
  do {                     \
    int rc_ = (expr);      \
    if (rc_) return rc_;   \
  } while (0)

// The persistent decoder runs the hoisted cross-attention order over a cache of accumulators; the reference's literal
// sequence (K/V cache format 3) exists in the stage-wise attention kernel only (decode_kernels.hip, DQAttnArgs::literal)
static bool fused_decoder_allowed(const slimt_hip_ctx *c) { return c->decode_mode != 1 && c->model->kv_format != 3; }

// The narrow K/V form for this model's next batch (engine.h, kv_auto_wide): format 0, and the sentences so far mostly fit.
// Called under the model's submit / gate discipline by encode_device; the counter lags the device by a few batches.
static bool kv_narrow_wanted(slimt_hip_model *m, unsigned long long **count_dev) {
  *count_dev = nullptr;
  if (m->kv_format != 0 || m->kv_auto_wide.load(std::memory_order_relaxed)) return false;
  {
    std::lock_guard<std::mutex> lock(m->gate_mu);
    if (!m->kv_wide_count) {
      void *p = nullptr;
      if (hipHostMalloc(&p, 256, hipHostMallocDefault) != hipSuccess) {  // [0] 24-bit, [1 + 4 gen + layer] not-16-bit (engine.h)
        (void)hipGetLastError();
        return true;  // (no counter: narrow without the watch)
      }
      std::memset(p, 0, 256);
      m->kv_wide_count = static_cast<unsigned long long *>(p);
    }
  }
  const unsigned long long wide = *static_cast<volatile unsigned long long *>(m->kv_wide_count);
  const unsigned long long total = m->kv_layers_submitted.load(std::memory_order_relaxed);
  // one sentence-layer in 32 is enough to make most 16-sentence workgroups wait for a fallback call every step
  // (SLIMT_KV_WATCH=0: never switch -- measurements of the fallback itself)
  static const bool watch = !(std::getenv("SLIMT_KV_WATCH") && std::getenv("SLIMT_KV_WATCH")[0] == '0');
  if (watch && total >= 1024 && wide * 32 > total) {
    m->kv_auto_wide.store(true, std::memory_order_relaxed);
    return false;
  }
  void *dev = nullptr;
  if (hipHostGetDevicePointer(&dev, m->kv_wide_count, 0) == hipSuccess) *count_dev = static_cast<unsigned long long *>(dev);
  else (void)hipGetLastError();
  return true;
}

static bool kv_tight_enabled() {
  static const bool enabled = !(std::getenv("SLIMT_KV_TIGHT") && std::getenv("SLIMT_KV_TIGHT")[0] == '0');
  return enabled;
}

// Where the tight form has a writer and a reader: D = 256 / F = 1536 (S <= 128; the 32-sentence tiling: S <= 32) and D = 512 /
// F = 2048 (S <= 32), not clusters, and an encoder with a writer for it (`writer`: kv_tight_writer).
static bool kv_tight_shape(const slimt_hip_ctx *c, int S, bool writer) {
  const slimt_hip_model *m = c->model;
  if (!kv_tight_enabled() || !writer || m->kv_tight_limit <= 0 || m->kv_format != 0) return false;
  if (S > 32)  // longer sentences: D = 256 only (33..64 tokens: one per 64-row workgroup; 65..128: the per-sentence encoder)
    return S <= 128 && m->D == 256 && c->decode_mode != 3 && c->decode_mode != 6 && c->decode_mode != 1 &&
           fused_decode_tight_mid_supported(m->D, m->F, m->H, m->Ld, S > 64 ? 2 : 1);
  // (the 32-sentence tiling -- mode 3, or mode 0 with a large output layer, which only the decoder launch knows: what this
  // context's last one saw stands in for it -- has the reader too where its LDS allows; else a batch that meets it is
  // decoded by the 16-sentence tiling)
  const bool rows32 = m->D == 256 && (c->decode_mode == 3 || (c->decode_mode == 0 && c->expect_large_output));
  if (rows32) return fused_decode_tight_rows32_supported(m->D, m->F, m->H, m->Ld);
  if (!(c->decode_mode == 0 || c->decode_mode == 2 || c->decode_mode == 3 || c->decode_mode == 4 || c->decode_mode == 5)) return false;
  return fused_decode_tight_supported(m->D, m->F, m->H, m->Ld);
}

static bool kv_centres_ready(slimt_hip_model *m) {
  const int state = m->kv_centre_state.load(std::memory_order_acquire);
  if (state == 2) return true;
  if (state == 1 && hipEventQuery(m->kv_centre_ev) == hipSuccess) {
    m->kv_centre_state.store(2, std::memory_order_release);
    return true;
  }
  (void)hipGetLastError();  // (hipErrorNotReady)
  return false;
}

// The tight form for this batch's decoder layers (engine.h, kv_tight_off): a mask of the layers that try it. Only where the
// decoder launch that follows will be one with the reader inlined (kv_tight_shape), and once the centres are there.
static unsigned kv_tight_wanted(slimt_hip_ctx *c, int S, bool writer, unsigned long long **count_dev) {
  slimt_hip_model *m = c->model;
  *count_dev = nullptr;
  if (!kv_tight_shape(c, S, writer) || !m->kv_wide_count || !kv_centres_ready(m)) return 0;
  static const bool watch = !(std::getenv("SLIMT_KV_WATCH") && std::getenv("SLIMT_KV_WATCH")[0] == '0');
  unsigned off = m->kv_tight_off.load(std::memory_order_relaxed);
  const int gen = m->kv_gen.load(std::memory_order_relaxed);
  unsigned tripped = 0;
  for (int l = 0; l < m->Ld && watch; ++l) {
    const unsigned long long missed = static_cast<volatile unsigned long long *>(m->kv_wide_count)[1 + 4 * gen + l];
    const unsigned long long total = m->kv_tight_submitted[l].load(std::memory_order_relaxed);
    // (as for the narrow form: the fallback is an out-of-line call its whole workgroup waits for)
    if (!((off >> l) & 1u) && total >= 1024 && missed * 32 > total) tripped |= 1u << l;
  }
  if (tripped) {
    if (gen < m->kv_recal_max && gen + 1 < slimt_hip_model::kKvGens) {
      // these centres do not fit the traffic: a new generation, calibrated from the next suitable batch (translate_device);
      // until it is ready nobody tries the form, and this generation's buffer stays as it is for the batches in flight
      for (auto &n : m->kv_tight_submitted) n.store(0, std::memory_order_relaxed);
      m->kv_gen.store(gen + 1, std::memory_order_relaxed);
      m->kv_centre_state.store(0, std::memory_order_release);
      m->kv_centre_claimed.store(false, std::memory_order_release);
      return 0;
    }
    off |= tripped;
    m->kv_tight_off.fetch_or(tripped, std::memory_order_relaxed);
  }
  const unsigned layers = ((1u << m->Ld) - 1u) & ~off;
  if (!layers) return 0;
  void *dev = nullptr;
  if (hipHostGetDevicePointer(&dev, m->kv_wide_count, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  *count_dev = static_cast<unsigned long long *>(dev) + 1 + 4 * gen;
  c->kv_gen = gen;
  return layers;
}

static void kv_watch_restart(slimt_hip_model *model) {
  model->kv_auto_wide.store(false, std::memory_order_relaxed);
  model->kv_layers_submitted.store(0, std::memory_order_relaxed);
  model->kv_tight_off.store(0, std::memory_order_relaxed);
  for (auto &n : model->kv_tight_submitted) n.store(0, std::memory_order_relaxed);
  if (model->kv_wide_count)
    for (int i = 0; i < 32; ++i) static_cast<volatile unsigned long long *>(model->kv_wide_count)[i] = 0;
}

hipError_t DevBuf::reserve(size_t n) {
  if (n <= bytes && p) return hipSuccess;
  g_hip_called.store(true, std::memory_order_relaxed);
  if (p) {
    hipError_t e = hipFree(p);
    if (e != hipSuccess) return e;
    p = nullptr;
    bytes = 0;
  }
  if (n == 0) n = 16;
  hipError_t e = hipMalloc(&p, n);
  if (e == hipSuccess) bytes = n;
  return e;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

namespace {
// SLIMT_HOST_TIMING=1: where a translate call's host time goes (printed when a model is destroyed)
struct HostTiming {
  bool on = std::getenv("SLIMT_HOST_TIMING") != nullptr;
  std::atomic<uint64_t> ns[8] = {};
  std::atomic<uint64_t> calls{0};
} g_timing;
const char *kTimingNames[8] = {"validate", "encode: prepare + launch", "submit + gate locks", "gate: wait-event call",
                               "decoder launch call", "gate: event record", "pinned-pointer views", "other"};
struct StageClock {
  std::chrono::steady_clock::time_point t;
  StageClock() { if (g_timing.on) t = std::chrono::steady_clock::now(); }
  void lap(int stage) {
    if (!g_timing.on) return;
    const auto now = std::chrono::steady_clock::now();
    g_timing.ns[stage] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count();
    t = now;
  }
};
void timing_report() {
  if (!g_timing.on || g_timing.calls == 0) return;
  std::fprintf(stderr, "host-timing: %llu translate calls; us per call:", (unsigned long long)g_timing.calls.load());
  for (int i = 0; i < 8; ++i)
    if (g_timing.ns[i]) std::fprintf(stderr, " [%s %.1f]", kTimingNames[i], 1e-3 * g_timing.ns[i] / g_timing.calls);
  std::fprintf(stderr, "\n");
}
}  // namespace

// Nothing here runs when the library is loaded, and nothing but this function writes the environment.
extern "C" int slimt_hip_request_hw_queues(int n) {
  if (n < 1 || n > 1024) return fail(-1, "hardware queues %d not in 1..1024", n);
  if (g_hip_called.load(std::memory_order_relaxed)) return 1;
  char value[16];
  snprintf(value, sizeof value, "%d", n);
  if (setenv("GPU_MAX_HW_QUEUES", value, /*overwrite=*/0)) return fail(-1, "setenv failed");
  return 0;
}

extern "C" int slimt_hip_hw_queues(void) {
  const char *v = getenv("GPU_MAX_HW_QUEUES");
  return v ? atoi(v) : 0;
}

namespace slimt_hip {
hipError_t set_dynamic_lds_once(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> largest;
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  int &have = largest[{device, kernel}];
  if (bytes <= have) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}
}  // namespace slimt_hip

extern "C" int slimt_hip_abi_version(void) { return SLIMT_HIP_ABI_VERSION; }
extern "C" const char *slimt_hip_last_error(void) { return g_err; }

extern "C" int slimt_hip_device_count(int *count) {
  if (!count) return fail(-1, "count is NULL");
  int n = 0;
  g_hip_called.store(true, std::memory_order_relaxed);
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail((int)e, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return 0;
}

extern "C" int slimt_hip_set_device(int device) {
  g_hip_called.store(true, std::memory_order_relaxed);
  HIPCHK(hipSetDevice(device));
  return 0;
}

// ---------------------------------------------------------------------------
// host-side weight preparation (load time; slimt/Io.cc:215,234)
// ---------------------------------------------------------------------------
extern "C" int slimt_hip_prepare_weight_transposed(const float *weights, int8_t *prepared,
                                                   float quantization_multiplier, size_t cols,
                                                   size_t rows) {
  if (!weights || !prepared) return fail(-1, "null argument");
  const size_t n = rows * cols;
  for (size_t i = 0; i < n; ++i) {
    float v = rintf(weights[i] * quantization_multiplier);
    if (v != v) v = -127.0f;  // NaN: what intgemm's convert + saturating packs + max yield
    v = v < -127.0f ? -127.0f : v;
    v = v > 127.0f ? 127.0f : v;
    prepared[i] = (int8_t)v;
  }
  return 0;
}

extern "C" int slimt_hip_prepare_weight_quantized_transposed(const int8_t *input, int8_t *output,
                                                             size_t rows, size_t cols) {
  if (!input || !output) return fail(-1, "null argument");
  if (input != output) memmove(output, input, rows * cols);
  return 0;
}

// ---------------------------------------------------------------------------
// op level
// ---------------------------------------------------------------------------
// process-wide diagnostic switch (slimt_hip_debug_occupancy_trace)
static slimt_hip::OccTrace g_occ_trace;


namespace {

struct TmpAffine {  // device temporaries of one stateless qmm call
  DevBuf x, W, bias, idx, y;
  AffineW aw;
  ~TmpAffine() {
    x.release(); W.release(); bias.release(); idx.release(); y.release();
    aw.Wp.release(); aw.colsum.release(); aw.pb.release();
  }
};

// buffers + descriptor of a prepared weight; the packing itself is a PackArgs job
int prepare_affine_meta(AffineW &aw, const int8_t *dW, int K, int N, const uint32_t *d_idx,
                        const float *d_bias, float a_quant, float b_quant, PackArgs &job, int n_src = 0) {
  const int n_tiles = (N + 15) / 16;
  HIPCHK(aw.Wp.reserve(packed_weight_bytes(K, N)));
  HIPCHK(aw.colsum.reserve(colsum_alloc_bytes(N)));
  HIPCHK(aw.pb.reserve((size_t)n_tiles * 16 * sizeof(float)));
  job.W = dW; job.K = K; job.N = N; job.idx = d_idx; job.bias = d_bias;
  job.n_src = n_src;
  job.mult = pack_mult(a_quant, b_quant);
  job.Wp = aw.Wp.p; job.colsum = aw.colsum.as<int>(); job.pb = aw.pb.as<float>();
  aw.w.Wp = aw.Wp.p;
  aw.w.colsum = aw.colsum.as<int>();
  aw.w.cp4 = aw.colsum.as<int>() + epi_pair_offset_ints(N);
  aw.w.pb = aw.pb.as<float>();
  aw.w.u = 1.0f / (a_quant * b_quant);  // Intgemm.inl.cc:146
  aw.w.a_quant = a_quant;
  aw.w.b_quant = b_quant;
  aw.w.K = K;
  aw.w.N = N;
  aw.w.n_tiles = n_tiles;
  return 0;
}

int prepare_affine(AffineW &aw, const int8_t *dW, int K, int N, const uint32_t *d_idx,
                   const float *d_bias, float a_quant, float b_quant, hipStream_t st) {
  const int n_tiles = (N + 15) / 16;
  HIPCHK(aw.Wp.reserve(packed_weight_bytes(K, N)));
  HIPCHK(aw.colsum.reserve(colsum_alloc_bytes(N)));
  HIPCHK(aw.pb.reserve((size_t)n_tiles * 16 * sizeof(float)));
  HIPCHK(launch_pack_weight(dW, K, N, d_idx, d_bias, a_quant, b_quant, aw.Wp.p,
                            aw.colsum.as<int>(), aw.pb.as<float>(), st));
  aw.w.Wp = aw.Wp.p;
  aw.w.colsum = aw.colsum.as<int>();
  aw.w.cp4 = aw.colsum.as<int>() + epi_pair_offset_ints(N);
  aw.w.pb = aw.pb.as<float>();
  aw.w.u = 1.0f / (a_quant * b_quant);  // Intgemm.inl.cc:146
  aw.w.a_quant = a_quant;
  aw.w.b_quant = b_quant;
  aw.w.K = K;
  aw.w.N = N;
  aw.w.n_tiles = n_tiles;
  return 0;
}

int affine_common(const float *x, size_t M, size_t K, const int8_t *W_nk, size_t N,
                  const float *bias, float a_quant, float b_quant, const uint32_t *idx,
                  size_t n_idx, float *y, int32_t *acc) {
  if (!x || !W_nk || (!y && !acc)) return fail(-1, "null argument");
  if (K == 0 || K % 64 != 0 || K > 4096) return fail(-1, "K=%zu must be a multiple of 64, <= 4096", K);
  if (M == 0 || N == 0) return fail(-1, "empty operand");
  if (idx && n_idx == 0) return fail(-1, "empty index list");
  hipStream_t st = nullptr;
  TmpAffine t;
  const size_t Nout = idx ? n_idx : N;
  HIPCHK(t.x.reserve(M * K * sizeof(float)));
  HIPCHK(t.W.reserve(N * K));
  HIPCHK(hipMemcpyAsync(t.x.p, x, M * K * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.W.p, W_nk, N * K, hipMemcpyHostToDevice, st));
  if (bias) {
    HIPCHK(t.bias.reserve(N * sizeof(float)));
    HIPCHK(hipMemcpyAsync(t.bias.p, bias, N * sizeof(float), hipMemcpyHostToDevice, st));
  }
  if (idx) {
    for (size_t i = 0; i < n_idx; ++i)
      if (idx[i] >= N) return fail(-1, "index %u out of range (N=%zu)", idx[i], N);
    HIPCHK(t.idx.reserve(n_idx * sizeof(uint32_t)));
    HIPCHK(hipMemcpyAsync(t.idx.p, idx, n_idx * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  }
  RCCHK(prepare_affine(t.aw, t.W.as<int8_t>(), (int)K, (int)Nout, idx ? t.idx.as<uint32_t>() : nullptr,
                       bias ? t.bias.as<float>() : nullptr, a_quant, b_quant, st));
  GemmArgs g;
  g.x_f32 = t.x.as<float>();
  g.lda = (int)K;
  g.M = (int)M;
  g.w = t.aw.w;
  if (acc) {
    HIPCHK(t.y.reserve(M * Nout * sizeof(int32_t)));
    g.acc_out = t.y.as<int32_t>();
    HIPCHK(launch_gemm(g, EPI_ACC, M >= 64 ? 64 : 16, st));
    HIPCHK(hipMemcpyAsync(acc, t.y.p, M * Nout * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  } else {
    HIPCHK(t.y.reserve(M * Nout * sizeof(float)));
    g.y = t.y.as<float>();
    g.ldy = (int)Nout;
    // many rows: the 128-row tiling (rows_per_block 0 asks for it where it applies)
    HIPCHK(launch_gemm(g, EPI_PLAIN, M >= 1024 ? 0 : (M >= 64 ? 64 : 16), st));
    HIPCHK(hipMemcpyAsync(y, t.y.p, M * Nout * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

}  // namespace

extern "C" int slimt_hip_affine(const float *x, size_t M, size_t K, const int8_t *W_nk, size_t N,
                                const float *bias, float a_quant, float b_quant, float *y) {
  return affine_common(x, M, K, W_nk, N, bias, a_quant, b_quant, nullptr, 0, y, nullptr);
}

extern "C" int slimt_hip_affine_select(const float *x, size_t M, size_t K, const int8_t *W_nk,
                                       size_t N, const float *bias, float a_quant, float b_quant,
                                       const uint32_t *idx, size_t n_idx, float *y) {
  if (!idx) return fail(-1, "idx is NULL");
  return affine_common(x, M, K, W_nk, N, bias, a_quant, b_quant, idx, n_idx, y, nullptr);
}

extern "C" int slimt_hip_affine_acc_i32(const float *x, size_t M, size_t K, const int8_t *W_nk,
                                        size_t N, float a_quant, int32_t *accS) {
  return affine_common(x, M, K, W_nk, N, nullptr, a_quant, 1.0f, nullptr, 0, nullptr, accS);
}

namespace {
struct Tmp3 {
  DevBuf a, b, c, d, e, f;
  ~Tmp3() { a.release(); b.release(); c.release(); d.release(); e.release(); f.release(); }
};
}  // namespace

extern "C" int slimt_hip_layer_norm(const float *x, const float *scale, const float *bias,
                                    float eps, size_t rows, size_t cols, float *y) {
  if (!x || !scale || !bias || !y || !rows || !cols) return fail(-1, "bad argument");
  Tmp3 t;
  hipStream_t st = nullptr;
  HIPCHK(t.a.reserve(rows * cols * 4));
  HIPCHK(t.b.reserve(cols * 4));
  HIPCHK(t.c.reserve(cols * 4));
  HIPCHK(t.d.reserve(rows * cols * 4));
  HIPCHK(hipMemcpyAsync(t.a.p, x, rows * cols * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.b.p, scale, cols * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.c.p, bias, cols * 4, hipMemcpyHostToDevice, st));
  HIPCHK(launch_layer_norm(t.a.as<float>(), t.b.as<float>(), t.c.as<float>(), eps, (int)rows,
                           (int)cols, t.d.as<float>(), st));
  HIPCHK(hipMemcpyAsync(y, t.d.p, rows * cols * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int slimt_hip_softmax(const float *x, size_t rows, size_t cols, float *y) {
  if (!x || !y || !rows || !cols) return fail(-1, "bad argument");
  Tmp3 t;
  hipStream_t st = nullptr;
  HIPCHK(t.a.reserve(rows * cols * 4));
  HIPCHK(t.b.reserve(rows * cols * 4));
  HIPCHK(hipMemcpyAsync(t.a.p, x, rows * cols * 4, hipMemcpyHostToDevice, st));
  HIPCHK(launch_softmax(t.a.as<float>(), (int)rows, (int)cols, t.b.as<float>(), st));
  HIPCHK(hipMemcpyAsync(y, t.b.p, rows * cols * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int slimt_hip_highway(const float *x, const float *y, const float *g, size_t n,
                                 float *out) {
  if (!x || !y || !g || !out || !n) return fail(-1, "bad argument");
  Tmp3 t;
  hipStream_t st = nullptr;
  HIPCHK(t.a.reserve(n * 4));
  HIPCHK(t.b.reserve(n * 4));
  HIPCHK(t.c.reserve(n * 4));
  HIPCHK(t.d.reserve(n * 4));
  HIPCHK(hipMemcpyAsync(t.a.p, x, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.b.p, y, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.c.p, g, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(launch_highway(t.a.as<float>(), t.b.as<float>(), t.c.as<float>(), n, t.d.as<float>(), st));
  HIPCHK(hipMemcpyAsync(out, t.d.p, n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int slimt_hip_sdpa(const float *q, const float *k, const float *v, const float *mask,
                              size_t B, size_t H, size_t Tq, size_t S, size_t dh, float *out,
                              float *attn) {
  if (!q || !k || !v || !mask || !out) return fail(-1, "null argument");
  if (S < 1 || S > 128 || dh < 1 || dh > 64) return fail(-1, "unsupported S=%zu dh=%zu", S, dh);
  Tmp3 t;
  DevBuf qj, kj, vj, oj, ob;
  struct G { DevBuf *b[5]; ~G() { for (auto *x : b) x->release(); } } guard{{&qj, &kj, &vj, &oj, &ob}};
  hipStream_t st = nullptr;
  const size_t D = H * dh;
  const size_t nq = B * Tq * D * 4, nk = B * S * D * 4;
  HIPCHK(t.a.reserve(nq)); HIPCHK(t.b.reserve(nk)); HIPCHK(t.c.reserve(nk));
  HIPCHK(t.d.reserve(B * S * 4));
  HIPCHK(qj.reserve(nq)); HIPCHK(kj.reserve(nk)); HIPCHK(vj.reserve(nk));
  HIPCHK(oj.reserve(nq)); HIPCHK(ob.reserve(nq));
  HIPCHK(hipMemcpyAsync(t.a.p, q, nq, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.b.p, k, nk, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.c.p, v, nk, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.d.p, mask, B * S * 4, hipMemcpyHostToDevice, st));
  // [B,H,T,dh] -> joined [B,T,H*dh] (the engine's native layout)
  HIPCHK(launch_transpose_heads(t.a.as<float>(), (int)B, (int)H, (int)Tq, (int)dh, qj.as<float>(), st));
  HIPCHK(launch_transpose_heads(t.b.as<float>(), (int)B, (int)H, (int)S, (int)dh, kj.as<float>(), st));
  HIPCHK(launch_transpose_heads(t.c.as<float>(), (int)B, (int)H, (int)S, (int)dh, vj.as<float>(), st));
  if (attn) HIPCHK(t.e.reserve(B * H * Tq * S * 4));
  AttnArgs a;
  a.q = qj.as<float>(); a.k = kj.as<float>(); a.v = vj.as<float>();
  a.ldq = a.ldk = a.ldv = a.ldo = (int)D;
  a.mask = t.d.as<float>();
  a.B = (int)B; a.H = (int)H; a.Tq = (int)Tq; a.S = (int)S; a.dh = (int)dh;
  a.alpha = 1.0f / std::sqrt((float)dh);  // Modules.cc:43
  a.out = oj.as<float>();
  a.attn = attn ? t.e.as<float>() : nullptr;
  HIPCHK(launch_attention(a, st));
  HIPCHK(launch_transpose_heads(oj.as<float>(), (int)B, (int)Tq, (int)H, (int)dh, ob.as<float>(), st));
  HIPCHK(hipMemcpyAsync(out, ob.p, nq, hipMemcpyDeviceToHost, st));
  if (attn) HIPCHK(hipMemcpyAsync(attn, t.e.p, B * H * Tq * S * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
namespace {

struct ParamTable {
  std::map<std::string, const slimt_hip_param *> by_name;
  const slimt_hip_param *get(const std::string &n) const {
    auto it = by_name.find(n);
    return it == by_name.end() ? nullptr : it->second;
  }
};

int upload_f32(DevBuf &dst, const ParamTable &t, const std::string &name, size_t expect) {
  const slimt_hip_param *p = t.get(name);
  if (!p) return fail(-1, "missing parameter %s", name.c_str());
  if (p->type != 0 || (size_t)p->rows * p->cols != expect)
    return fail(-1, "parameter %s: expected f32 with %zu elements", name.c_str(), expect);
  HIPCHK(dst.reserve(expect * 4));
  HIPCHK(hipMemcpy(dst.p, p->data, expect * 4, hipMemcpyHostToDevice));
  return 0;
}

int scalar_f32(const ParamTable &t, const std::string &name, float *out) {
  const slimt_hip_param *p = t.get(name);
  if (!p) return fail(-1, "missing parameter %s", name.c_str());
  if (p->type != 0) return fail(-1, "parameter %s must be f32", name.c_str());
  *out = *reinterpret_cast<const float *>(p->data);
  return 0;
}

// Affine{W,b,quant} / Linear{W,quant} binding (Modules.cc:145-180,361-400)
int load_affine(AffineW &aw, const ParamTable &t, const std::string &wname,
                const std::string &bname /* may be empty */, int K, int N, DevBuf &scratch) {
  const slimt_hip_param *W = t.get(wname);
  if (!W) return fail(-1, "missing parameter %s", wname.c_str());
  if (W->type != 1 || W->rows != K || W->cols != N)
    return fail(-1, "parameter %s: expected intgemm8 [%d,%d], got type %d [%d,%d]", wname.c_str(),
                K, N, W->type, W->rows, W->cols);
  float a_quant = 0.f, b_quant = 0.f;
  RCCHK(scalar_f32(t, wname + "_QuantMultA", &a_quant));
  // b_quant sits right after the int8 payload (Modules.cc:18-22)
  memcpy(&b_quant, reinterpret_cast<const int8_t *>(W->data) + (size_t)K * N, sizeof(float));
  DevBuf bias;
  if (!bname.empty()) {
    int rc = upload_f32(bias, t, bname, (size_t)N);
    if (rc) return rc;
  }
  HIPCHK(scratch.reserve((size_t)K * N));
  // qmm::prepare_weight_quantized_transposed (Io.cc:234) is the identity for
  // this backend: the file layout [N][K] is the canonical layout.
  HIPCHK(hipMemcpy(scratch.p, W->data, (size_t)K * N, hipMemcpyHostToDevice));
  int rc = prepare_affine(aw, scratch.as<int8_t>(), K, N, nullptr,
                          bname.empty() ? nullptr : bias.as<float>(), a_quant, b_quant, nullptr);
  hipError_t e = hipDeviceSynchronize();
  bias.release();
  if (rc) return rc;
  HIPCHK(e);
  return 0;
}

int load_ln(LnW &ln, const ParamTable &t, const std::string &prefix, int D) {
  RCCHK(upload_f32(ln.scale, t, prefix + "_ln_scale", (size_t)D));
  RCCHK(upload_f32(ln.bias, t, prefix + "_ln_bias", (size_t)D));
  return 0;
}

int load_attn(AttnW &a, const ParamTable &t, const std::string &prefix, int D, DevBuf &scratch) {
  RCCHK(load_affine(a.q, t, prefix + "Wq", prefix + "bq", D, D, scratch));
  RCCHK(load_affine(a.k, t, prefix + "Wk", prefix + "bk", D, D, scratch));
  RCCHK(load_affine(a.v, t, prefix + "Wv", prefix + "bv", D, D, scratch));
  RCCHK(load_affine(a.o, t, prefix + "Wo", prefix + "bo", D, D, scratch));
  RCCHK(load_ln(a.ln, t, prefix + "Wo", D));
  return 0;
}

void free_affine(AffineW &a) {
  a.Wp.release();
  a.colsum.release();
  a.pb.release();
}
void free_ln(LnW &l) {
  l.scale.release();
  l.bias.release();
}
void free_attn(AttnW &a) {
  free_affine(a.q); free_affine(a.k); free_affine(a.v); free_affine(a.o);
  free_ln(a.ln);
}

void model_free(slimt_hip_model *m) {
  for (auto &L : m->enc) {
    free_attn(L.attn); free_affine(L.ffn1); free_affine(L.ffn2); free_ln(L.ffn_ln);
  }
  for (auto &L : m->dec) {
    free_affine(L.rnn_f); free_affine(L.rnn_w); free_ln(L.rnn_ln);
    free_attn(L.attn); free_affine(L.ffn1); free_affine(L.ffn2); free_ln(L.ffn_ln);
  }
  m->wemb.release(); m->out_raw.release(); m->out_bias.release();
  free_affine(m->out_full);
  for (hipEvent_t ev : m->gate_ev) (void)hipEventDestroy(ev);
  m->gate_ev.clear();
}

int model_build(slimt_hip_model *m, const slimt_hip_param *params, size_t n,
                const slimt_hip_dims *dims) {
  ParamTable t;
  for (size_t i = 0; i < n; ++i) {
    const slimt_hip_param &p = params[i];
    if (!p.name || !p.data) continue;
    if (p.type != 0 && p.type != 1) return fail(-1, "parameter %s: unknown type %d", p.name, p.type);
    if (p.rows <= 0 || p.cols <= 0)
      return fail(-1, "parameter %s: bad shape [%d,%d]", p.name, p.rows, p.cols);
    // a truncated or malformed file must not make the uploads below read past its mapping
    const uint64_t need = (uint64_t)p.rows * (uint64_t)p.cols * (p.type == 0 ? 4u : 1u) + (p.type == 1 ? 4u : 0u);
    if (p.bytes && p.bytes < need)
      return fail(-1, "parameter %s: %llu bytes, its shape [%d,%d] needs %llu", p.name,
                  (unsigned long long)p.bytes, p.rows, p.cols, (unsigned long long)need);
    t.by_name[p.name] = &p;
  }
  const slimt_hip_param *wemb = t.get("Wemb");
  if (!wemb) return fail(-1, "missing parameter Wemb");
  if (wemb->type != 1) return fail(-1, "Wemb must be intgemm8");
  m->V = wemb->rows;
  m->D = wemb->cols;
  m->H = dims->num_heads;
  m->Le = dims->encoder_layers;
  m->Ld = dims->decoder_layers;
  const int D = m->D, V = m->V;
  if (D % 64 != 0 || D > 512) return fail(-1, "unsupported embedding size %d", D);
  if (m->H <= 0 || D % m->H != 0 || D / m->H > 64) return fail(-1, "unsupported head count %d", m->H);
  const slimt_hip_param *w1 = t.get("encoder_l1_ffn_W1");
  if (!w1) return fail(-1, "missing parameter encoder_l1_ffn_W1");
  m->F = w1->cols;
  if (m->F % 64 != 0 || m->F > 4096) return fail(-1, "unsupported ffn size %d", m->F);
  const int F = m->F;
  const size_t VD = (size_t)V * D;

  // Wemb (Io.cc:182-224): keep the int8 table for lookups (E = q * (1/m)),
  // and re-quantise the dequantised table for the tied output layer exactly
  // as io::load_items does through qmm::prepare_weight_transposed.
  const int8_t *wq = reinterpret_cast<const int8_t *>(wemb->data);
  memcpy(&m->wemb_mult, wq + VD, sizeof(float));
  HIPCHK(m->wemb.reserve(VD));
  HIPCHK(hipMemcpy(m->wemb.p, wq, VD, hipMemcpyHostToDevice));
  {
    std::vector<float> E(VD);
    const float inv = 1 / m->wemb_mult;  // Io.cc:280-281
    for (size_t i = 0; i < VD; ++i) E[i] = static_cast<float>(wq[i]) * inv;
    std::vector<int8_t> prepared(VD);
    RCCHK(slimt_hip_prepare_weight_transposed(E.data(), prepared.data(), m->wemb_mult, (size_t)D,
                                              (size_t)V));
    HIPCHK(m->out_raw.reserve(VD));
    HIPCHK(hipMemcpy(m->out_raw.p, prepared.data(), VD, hipMemcpyHostToDevice));
  }
  RCCHK(scalar_f32(t, "none_QuantMultA", &m->out_a_quant));
  RCCHK(upload_f32(m->out_bias, t, "decoder_ff_logit_out_b", (size_t)V));
  RCCHK(prepare_affine(m->out_full, m->out_raw.as<int8_t>(), D, V, nullptr,
                       m->out_bias.as<float>(), m->out_a_quant, m->wemb_mult, nullptr));
  HIPCHK(hipDeviceSynchronize());

  DevBuf scratch;
  struct G { DevBuf *b; ~G() { b->release(); } } guard{&scratch};
  m->enc.resize((size_t)m->Le);
  m->dec.resize((size_t)m->Ld);
  for (int i = 0; i < m->Le; ++i) {
    const std::string L = "encoder_l" + std::to_string(i + 1);
    auto &E = m->enc[(size_t)i];
    RCCHK(load_attn(E.attn, t, L + "_self_", D, scratch));
    RCCHK(load_affine(E.ffn1, t, L + "_ffn_W1", L + "_ffn_b1", D, F, scratch));
    RCCHK(load_affine(E.ffn2, t, L + "_ffn_W2", L + "_ffn_b2", F, D, scratch));
    RCCHK(load_ln(E.ffn_ln, t, L + "_ffn_ffn", D));
  }
  for (int i = 0; i < m->Ld; ++i) {
    const std::string L = "decoder_l" + std::to_string(i + 1);
    auto &Dl = m->dec[(size_t)i];
    RCCHK(load_affine(Dl.rnn_w, t, L + "_rnn_W", "", D, D, scratch));
    RCCHK(load_affine(Dl.rnn_f, t, L + "_rnn_Wf", L + "_rnn_bf", D, D, scratch));
    RCCHK(load_ln(Dl.rnn_ln, t, L + "_rnn_ffn", D));
    RCCHK(load_attn(Dl.attn, t, L + "_context_", D, scratch));
    RCCHK(load_affine(Dl.ffn1, t, L + "_ffn_W1", L + "_ffn_b1", D, F, scratch));
    RCCHK(load_affine(Dl.ffn2, t, L + "_ffn_W2", L + "_ffn_b2", F, D, scratch));
    RCCHK(load_ln(Dl.ffn_ln, t, L + "_ffn_ffn", D));
  }
  return 0;
}

}  // namespace

extern "C" int slimt_hip_model_create(const slimt_hip_param *params, size_t n_params,
                                      const slimt_hip_dims *dims, int device,
                                      slimt_hip_model **out) {
  if (!params || !dims || !out) return fail(-1, "null argument");
  *out = nullptr;
  int count = 0;
  RCCHK(slimt_hip_device_count(&count));
  if (count <= 0) return fail(-1, "no HIP device available (this backend has no CPU fallback)");
  if (device < 0 || device >= count) return fail(-1, "device %d out of range (%d devices)", device, count);
  HIPCHK(hipSetDevice(device));
  auto *m = new slimt_hip_model();
  m->device = device;
  {  // default decoder admission: three quarters of the CUs (measured optimum)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->decoder_budget = 7 * prop.multiProcessorCount / 8;
  }
  int rc = model_build(m, params, n_params, dims);
  if (rc) {
    model_free(m);
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}

// The Marian .bin container (slimt/Io.cc:114-161): version, item count, headers, names, shapes,
// padding to a 256-byte boundary, payloads. Only views are taken; every read is bounds-checked.
extern "C" int slimt_hip_model_create_from_bin(const void *bin, size_t size, const slimt_hip_dims *dims,
                                               int device, slimt_hip_model **out) {
  if (!bin || !dims || !out) return fail(-1, "null argument");
  *out = nullptr;
  const char *p = static_cast<const char *>(bin), *const end = p + size;
  auto take = [&](size_t n) -> const char * {
    if ((size_t)(end - p) < n) return nullptr;
    const char *q = p;
    p += n;
    return q;
  };
  auto u64 = [&](uint64_t *v) -> bool {
    const char *q = take(8);
    if (q) memcpy(v, q, 8);
    return q != nullptr;
  };
  uint64_t version = 0, n = 0;
  if (!u64(&version) || !u64(&n)) return fail(-1, "truncated Marian .bin");
  if (version != 1) return fail(-1, "Marian .bin version %llu != 1", (unsigned long long)version);
  if (n > (1u << 20)) return fail(-1, "implausible item count %llu", (unsigned long long)n);
  struct Header { uint64_t name_length, type, shape_length, data_length; };  // slimt/Io.hh:24-29
  std::vector<Header> headers((size_t)n);
  for (Header &h : headers) {
    const char *q = take(sizeof(Header));
    if (!q) return fail(-1, "truncated Marian .bin");
    memcpy(&h, q, sizeof(Header));
  }
  std::vector<std::string> names((size_t)n);
  for (size_t i = 0; i < n; ++i) {
    const char *q = take(headers[i].name_length);
    if (!q) return fail(-1, "truncated Marian .bin");
    names[i].assign(q, headers[i].name_length ? headers[i].name_length - 1 : 0);
  }
  std::vector<slimt_hip_param> params;
  params.reserve((size_t)n);
  std::vector<size_t> which;
  for (size_t i = 0; i < n; ++i) {
    if (headers[i].shape_length > 8) return fail(-1, "item %s: %llu dimensions", names[i].c_str(), (unsigned long long)headers[i].shape_length);
    int32_t rows = 1, cols = 1;
    for (uint64_t d = 0; d < headers[i].shape_length; ++d) {
      const char *q = take(4);
      if (!q) return fail(-1, "truncated Marian .bin");
      int32_t v;
      memcpy(&v, q, 4);
      if (d + 2 == headers[i].shape_length) rows = v;
      if (d + 1 == headers[i].shape_length) cols = v;
    }
    const int type = headers[i].type == 0x0404 ? 0 : headers[i].type == 0x4101 ? 1 : -1;  // f32 / intgemm8 (Io.cc:37-84)
    if (type < 0) continue;  // e.g. special:model.yml
    slimt_hip_param q;
    q.name = names[i].c_str();
    q.type = type;
    q.rows = rows;
    q.cols = cols;
    q.data = nullptr;
    q.bytes = headers[i].data_length;
    params.push_back(q);
    which.push_back(i);
  }
  uint64_t pad = 0;
  if (!u64(&pad) || !take(pad)) return fail(-1, "truncated Marian .bin");  // Io.cc:151-153
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) {
    const char *q = take(headers[i].data_length);
    if (!q) return fail(-1, "truncated Marian .bin (payload of %s)", names[i].c_str());
    if (k < which.size() && which[k] == i) params[k++].data = q;
  }
  return slimt_hip_model_create(params.data(), params.size(), dims, device, out);  // checks shapes against bytes
}

extern "C" int slimt_hip_model_destroy(slimt_hip_model *model) {
  if (!model) return 0;
  timing_report();
  (void)hipSetDevice(model->device);
  model_free(model);
  if (model->kv_wide_count) (void)hipHostFree(model->kv_wide_count);
  for (auto &b : model->kv_centre) b.release();
  model->kv_centre_sums.release();
  if (model->kv_centre_ev) (void)hipEventDestroy(model->kv_centre_ev);
  delete model;
  return 0;
}

extern "C" int slimt_hip_model_set_decoder_budget(slimt_hip_model *model, int workgroups) {
  if (!model) return fail(-1, "model is NULL");
  if (workgroups < 0) return fail(-1, "decoder budget %d < 0", workgroups);
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->decoder_budget = workgroups;
  return 0;
}

extern "C" int slimt_hip_model_set_kv_cache_policy(slimt_hip_model *model, int policy) {
  if (!model) return fail(-1, "model is NULL");
  if (policy < 0 || policy > 2) return fail(-1, "K/V cache policy %d not in 0..2", policy);
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->kv_policy = policy;
  return 0;
}

extern "C" int slimt_hip_model_set_xcd_affinity(slimt_hip_model *model, int xcds) {
  if (!model) return fail(-1, "model is NULL");
  if (xcds != 0 && xcds != 1 && xcds != 2 && xcds != 4) return fail(-1, "XCD affinity %d not in {0, 1, 2, 4}", xcds);
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->xcd_affinity = xcds;
  return 0;
}

extern "C" int slimt_hip_model_set_adaptive_decoder_rows(slimt_hip_model *model, int on) {
  if (!model) return fail(-1, "model is NULL");
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->adaptive_rows = on != 0;
  return 0;
}

extern "C" int slimt_hip_model_set_kv_cache_format(slimt_hip_model *model, int format) {
  if (!model) return fail(-1, "model is NULL");
  if (format < 0 || format > 3) return fail(-1, "K/V cache format %d not in 0..3", format);
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->kv_format = format;
  kv_watch_restart(model);  // choosing a format starts format 0's watches afresh (engine.h, kv_auto_wide)
  return 0;
}

extern "C" int slimt_hip_debug_kv_watch(slimt_hip_model *model, int *switched_to_24_bit, uint64_t *wide, uint64_t *submitted) {
  if (!model) return fail(-1, "model is NULL");
  if (switched_to_24_bit) *switched_to_24_bit = model->kv_auto_wide.load(std::memory_order_relaxed) ? 1 : 0;
  if (wide) *wide = model->kv_wide_count ? *static_cast<volatile unsigned long long *>(model->kv_wide_count) : 0;
  if (submitted) *submitted = model->kv_layers_submitted.load(std::memory_order_relaxed);
  return 0;
}

extern "C" int slimt_hip_debug_cross_attention(slimt_hip_ctx *ctx, int layer, int literal, const float *yq, float *joined,
                                               float *attn) {
  if (!ctx || !yq || !joined) return fail(-1, "null argument");
  const slimt_hip_model *m = ctx->model;
  if (layer < 0 || layer >= m->Ld) return fail(-1, "decoder layer %d not in 0..%d", layer, m->Ld - 1);
  if (!ctx->decode_ready || !ctx->kv_ready)
    return fail(-1, "no f32 K/V cache: call slimt_hip_decode_begin (or _begin_from) for the batch first");
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  const int B = ctx->B, S = ctx->S, D = m->D, M = B * S;
  const DecLayerW &L = m->dec[(size_t)layer];
  HIPCHK(ctx->dout.reserve((size_t)B * D * 4));
  HIPCHK(ctx->dh.reserve((size_t)B * D * 4));
  HIPCHK(ctx->attn_dbg.reserve((size_t)B * m->H * S * 4));
  HIPCHK(hipMemcpyAsync(ctx->dh.p, yq, (size_t)B * D * 4, hipMemcpyHostToDevice, st));
  DQAttnArgs a;
  a.B = B; a.D = D; a.H = m->H; a.S = S;
  a.x.x = ctx->dh.as<float>();  // (unused rows: the queries are given)
  a.wq = L.attn.q.w;
  a.q_given = ctx->dh.as<float>();
  a.literal = literal != 0;
  const float *kv = ctx->kv.as<float>();
  a.k = kv + (size_t)(2 * layer) * M * D;
  a.v = kv + (size_t)(2 * layer + 1) * M * D;
  a.ldv = D;
  a.uk = L.attn.k.w.u;
  a.uv = L.attn.v.w.u;
  a.pbk = L.attn.k.w.pb;
  a.pbv = L.attn.v.w.pb;
  a.lengths = ctx->lengths.as<uint32_t>();
  a.alpha = 1.0f / std::sqrt(static_cast<float>(D / m->H));
  a.out_f32 = ctx->dout.as<float>();
  a.attn = ctx->attn_dbg.as<float>();
  HIPCHK(launch_dqattn(a, st));
  HIPCHK(hipMemcpyAsync(joined, ctx->dout.p, (size_t)B * D * 4, hipMemcpyDeviceToHost, st));
  if (attn) HIPCHK(hipMemcpyAsync(attn, ctx->attn_dbg.p, (size_t)B * m->H * S * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int slimt_hip_debug_break_shortlist_handoff(slimt_hip_ctx *ctx, int broken, unsigned poll_limit) {
  if (!ctx) return fail(-1, "ctx is NULL");
  if (poll_limit == 0 || poll_limit > (1u << 24)) return fail(-1, "poll limit %u not in 1..2^24", poll_limit);
  ctx->gen_break = broken != 0;
  ctx->gen_spin_limit = poll_limit;
  return 0;
}

extern "C" int slimt_hip_debug_kv_narrow_limit(slimt_hip_model *model, int limit) {
  if (!model) return fail(-1, "model is NULL");
  if (limit < 1 || limit > (1 << 19)) return fail(-1, "narrow-form limit %d not in 1..2^19 (20 bits hold [-2^19, 2^19))", limit);
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->kv_narrow_limit = limit;
  kv_watch_restart(model);  // (a new limit: the watch starts afresh)
  return 0;
}

extern "C" int slimt_hip_debug_kv_tight_limit(slimt_hip_model *model, int limit) {
  if (!model) return fail(-1, "model is NULL");
  if (limit < 0 || limit > (1 << 15)) return fail(-1, "tight-form limit %d not in 0..2^15 (int16 holds [-2^15, 2^15); 0 = never tried)", limit);
  std::lock_guard<std::mutex> lock(model->gate_mu);
  model->kv_tight_limit = limit;
  kv_watch_restart(model);
  return 0;
}

extern "C" int slimt_hip_debug_kv_centres(slimt_hip_model *model, int32_t *out, size_t n, int *ready) {
  if (!model || !ready) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(model->device));
  *ready = kv_centres_ready(model) ? 1 : 0;
  if (!*ready || !out) return 0;
  const size_t have = (size_t)model->Ld * 2 * (size_t)model->D;
  if (n < have) return fail(-1, "kv centres: room for %zu values, the model has %zu", n, have);
  HIPCHK(hipMemcpy(out, model->kv_centre[model->kv_gen.load(std::memory_order_relaxed)].p, have * 4, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int slimt_hip_debug_kv_recalibrations(slimt_hip_model *model, int *generations_started, int max_recalibrations) {
  if (!model) return fail(-1, "model is NULL");
  if (max_recalibrations >= 0) model->kv_recal_max = std::min(max_recalibrations, slimt_hip_model::kKvGens - 1);
  if (generations_started) *generations_started = model->kv_gen.load(std::memory_order_relaxed);
  return 0;
}

extern "C" int slimt_hip_model_set_kv_centres(slimt_hip_model *model, const int32_t *centres, size_t n) {
  if (!model || !centres) return fail(-1, "null argument");
  const size_t have = (size_t)model->Ld * 2 * (size_t)model->D;
  if (n != have) return fail(-1, "kv centres: %zu values given, the model takes Ld * 2 * D = %zu", n, have);
  for (size_t i = 0; i < n; ++i)
    if (centres[i] <= -(1 << 24) || centres[i] >= (1 << 24)) return fail(-1, "kv centre %zu = %d not within (-2^24, 2^24)", i, centres[i]);
  HIPCHK(hipSetDevice(model->device));
  //  serialised against translate calls: they read the generation, its state and its buffer under these locks
  std::lock_guard<std::mutex> submit(model->submit_mu);
  std::lock_guard<std::mutex> gate(model->gate_mu);
  const int gen = model->kv_gen.load(std::memory_order_relaxed);
  const bool was_claimed = model->kv_centre_claimed.exchange(true, std::memory_order_acq_rel);
  if (was_claimed && model->kv_centre_state.load(std::memory_order_acquire) != 2 && !kv_centres_ready(model))
    return fail(-1, "kv centres: a calibration batch is in flight");
  // whatever fails below gives the claim back (a claim without centres would keep every batch from calibrating: the form
  // would silently never be used)
  struct Unclaim {
    slimt_hip_model *m;
    bool armed;
    ~Unclaim() {
      if (armed) m->kv_centre_claimed.store(false, std::memory_order_release);
    }
  } unclaim{model, !was_claimed};
  hipError_t e = model->kv_centre[gen].reserve(have * 4);
  if (e == hipSuccess) e = hipDeviceSynchronize();  // (the caller promises no batch in flight; make the copy safe against a finished one's tail)
  if (e == hipSuccess) e = hipMemcpy(model->kv_centre[gen].p, centres, have * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail((int)e, "kv centres: %s", hipGetErrorString(e));
  unclaim.armed = false;
  model->kv_centre_state.store(2, std::memory_order_release);
  // layers switched off under the centres before, and the counters they were judged by, start over with these
  model->kv_tight_off.store(0, std::memory_order_relaxed);
  for (auto &c : model->kv_tight_submitted) c.store(0, std::memory_order_relaxed);
  if (model->kv_wide_count)
    for (int l = 0; l < 4; ++l) static_cast<volatile unsigned long long *>(model->kv_wide_count)[1 + 4 * gen + l] = 0;
  return 0;
}

extern "C" int slimt_hip_debug_kv_tight_watch(slimt_hip_model *model, unsigned *layers_off, uint64_t *missed, uint64_t *submitted) {
  if (!model) return fail(-1, "model is NULL");
  if (layers_off) *layers_off = model->kv_tight_off.load(std::memory_order_relaxed);
  for (int l = 0; l < 4; ++l) {
    if (missed)
      missed[l] = model->kv_wide_count ? static_cast<volatile unsigned long long *>(
                                              model->kv_wide_count)[1 + 4 * model->kv_gen.load(std::memory_order_relaxed) + l] : 0;
    if (submitted) submitted[l] = model->kv_tight_submitted[l].load(std::memory_order_relaxed);
  }
  return 0;
}

extern "C" int slimt_hip_debug_kv_formats(slimt_hip_ctx *ctx, uint8_t *out, size_t n, size_t *batch) {
  if (!ctx || !out || !batch) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(ctx->model->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *batch = 0;
  if (!ctx->kv_fmt_valid) return 0;  // the last batch's caches are all in one form (24-bit or f32)
  const size_t have = (size_t)ctx->model->Ld * (size_t)ctx->kv_fmt_B;
  if (n < have) return fail(-1, "kv formats: %zu bytes for %zu", n, have);
  HIPCHK(hipMemcpy(out, ctx->kv_fmt.p, have, hipMemcpyDeviceToHost));
  *batch = (size_t)ctx->kv_fmt_B;
  return 0;
}

extern "C" int slimt_hip_model_device(const slimt_hip_model *model) { return model ? model->device : -1; }

extern "C" int slimt_hip_model_info(const slimt_hip_model *model, int32_t *dim_emb,
                                    int32_t *dim_ffn, int32_t *vocab, int32_t *heads) {
  if (!model) return fail(-1, "model is NULL");
  if (dim_emb) *dim_emb = model->D;
  if (dim_ffn) *dim_ffn = model->F;
  if (vocab) *vocab = model->V;
  if (heads) *heads = model->H;
  return 0;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
namespace {

// sinusoidal_signal (TensorOps.cc:245-265): host libm, like the reference.
void sinusoid_table(int S, int D, std::vector<float> &out) {
  out.assign((size_t)S * D, 0.f);
  float num_timescales = static_cast<float>(D) / 2;
  // std::log(10000.0F), written as its correctly rounded f32 value so that no
  // compiler's constant folding can differ from another's.
  const float log_10000 = 9.210340371976184f;
  float log_timescale_increment = log_10000 / (num_timescales - 1.0F);
  for (size_t p = 0; p < (size_t)S; ++p) {
    for (int i = 0; i < num_timescales; ++i) {
      float v = p * std::exp(i * -log_timescale_increment);
      size_t offset = p * (size_t)D + (size_t)i;
      out[offset] = std::sin(v);
      out[offset + static_cast<int>(num_timescales)] = std::cos(v);
    }
  }
}

void ctx_free(slimt_hip_ctx *c) {
  DevBuf *bufs[] = {&c->pos, &c->ids, &c->lengths, &c->x0, &c->x1, &c->q, &c->k, &c->v, &c->att,
                    &c->h8, &c->a8, &c->ticket, &c->kv, &c->kv_fmt, &c->cl_act, &c->cl_part, &c->cl_sync, &c->dx, &c->dx_pre, &c->dh, &c->datt8, &c->dout, &c->df8,
                    &c->state, &c->part_val, &c->part_idx, &c->prev, &c->out_ids, &c->out_len,
                    &c->finished, &c->n_finished, &c->align, &c->shortlist, &c->logits,
                    &c->attn_dbg, &c->stamps, &c->dbg_embed, &c->dbg_layers, &c->sl_scratch, &c->n_sl_dev, &c->gen_flag};
  for (auto *b : bufs) b->release();
  free_affine(c->out_sl);
  if (c->n_finished_host) (void)hipHostFree(c->n_finished_host);
  for (auto &e : c->prof_events) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (c->sync_event) (void)hipEventDestroy(c->sync_event);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
}

int ctx_alloc(slimt_hip_ctx *c) {
  const slimt_hip_model *m = c->model;
  const size_t B = c->max_B, S = c->max_S, M = c->max_M;
  const size_t D = (size_t)m->D, F = (size_t)m->F, V = (size_t)m->V;
  std::vector<float> pos;
  sinusoid_table((int)S, (int)D, pos);
  HIPCHK(c->pos.reserve(pos.size() * 4));
  HIPCHK(hipMemcpy(c->pos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c->ids.reserve(M * 4));
  HIPCHK(c->lengths.reserve(B * 4));
  HIPCHK(c->x0.reserve(M * D * 4));
  HIPCHK(c->x1.reserve(M * D * 4));
  HIPCHK(c->q.reserve(M * D * 4));
  HIPCHK(c->k.reserve(M * D * 4));
  HIPCHK(c->v.reserve(M * D * 4));
  HIPCHK(c->att.reserve(M * D * 4));
  HIPCHK(c->h8.reserve(M * F));
  HIPCHK(c->a8.reserve(M * D));
  HIPCHK(c->ticket.reserve(16));  // [0] decoder, [1] fused encoder, [2..3] XCD-affine claim state (64 bits)
  HIPCHK(hipMemset(c->ticket.p, 0, 16));
  HIPCHK(c->kv.reserve((size_t)m->Ld * 2 * M * D * 4));
  HIPCHK(c->dx.reserve(B * D * 4));
  HIPCHK(c->dx_pre.reserve(B * D * 4));
  HIPCHK(c->dh.reserve(B * D * 4));
  HIPCHK(c->datt8.reserve(B * D));
  HIPCHK(c->dout.reserve(B * D * 4));
  HIPCHK(c->df8.reserve(B * F));
  HIPCHK(c->state.reserve((size_t)m->Ld * B * D * 4));
  const size_t max_parts = (V + 63) / 64;
  HIPCHK(c->part_val.reserve(B * max_parts * 4));
  HIPCHK(c->part_idx.reserve(B * max_parts * 4));
  HIPCHK(c->prev.reserve(B * 4));
  HIPCHK(c->out_len.reserve(B * 4));
  HIPCHK(c->finished.reserve(B));
  HIPCHK(c->n_finished.reserve(16));
  HIPCHK(c->shortlist.reserve(V * 4));
  HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->n_finished_host), 16, hipHostMallocDefault));
  std::memset(c->n_finished_host, 0, 16);  // [0] early-exit read-back, [1] size of the last generated shortlist
  return 0;
}

struct ProfScope {  // HIP events around one launch of the selected kernel family
  slimt_hip_ctx *c;
  bool on;
  ProfScope(slimt_hip_ctx *ctx, int family, double macs, double bytes)
      : c(ctx), on(ctx->prof_kernel == family) {
    if (!on) return;
    if (c->prof_used == c->prof_events.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        on = false;
        return;
      }
      c->prof_events.emplace_back(a, b);
    }
    c->prof_macs += macs;
    c->prof_bytes += bytes;
    (void)hipEventRecord(c->prof_events[c->prof_used].first, c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(c->prof_events[c->prof_used].second, c->stream);
    c->prof_used++;
  }
};

double gemm_macs(int M, const PreparedWeight &w) { return (double)M * w.K * w.N; }
double gemm_bytes(const PreparedWeight &w) { return (double)w.K * w.n_tiles * 16; }

}  // namespace

// Contexts alive per device in this process. Every context is a stream, and past ~22 of them the device's
// hardware queues are time-sliced whatever GPU_MAX_HW_QUEUES says: 21 / 22 workers 32.5 / 32.1 M tok/s, 24 workers
// 23.8 M, 32 workers 20.4 M (profiles/archive/r04_v1_budget_workers_sweep.txt). Said once, on stderr, when it happens
// (SLIMT_HIP_QUIET=1: not said); slimt_hip_contexts_on_device reports the count.
static constexpr int kMaxDevices = 64;
static constexpr int kContextCliff = 22;
static std::atomic<int> g_live_ctx[kMaxDevices];

static void count_context(int device, int delta) {
  if (device < 0 || device >= kMaxDevices) return;
  const int now = g_live_ctx[device].fetch_add(delta, std::memory_order_relaxed) + delta;
  if (delta > 0 && now > kContextCliff) {
    static std::atomic<bool> warned{false};
    const char *quiet = std::getenv("SLIMT_HIP_QUIET");
    if (!(quiet && quiet[0] == '1') && !warned.exchange(true))
      std::fprintf(stderr,
                   "slimt_hip: %d contexts on device %d in one process: past %d the device's hardware queues are "
                   "time-sliced (24 contexts measured 30 %% below 20); use fewer workers with larger batches, or one "
                   "process per device\n",
                   now, device, kContextCliff);
  }
}

extern "C" int slimt_hip_contexts_on_device(int device, int *count) {
  if (!count) return fail(-1, "null argument");
  if (device < 0 || device >= kMaxDevices) return fail(-1, "device %d out of range", device);
  *count = g_live_ctx[device].load(std::memory_order_relaxed);
  return 0;
}

extern "C" int slimt_hip_ctx_create(slimt_hip_model *model, size_t max_batch,
                                    size_t max_source_length, void *stream, slimt_hip_ctx **out) {
  return slimt_hip_ctx_create_budget(model, max_batch, max_source_length,
                                     max_batch * max_source_length, stream, out);
}

extern "C" int slimt_hip_ctx_create_budget(slimt_hip_model *model, size_t max_batch,
                                           size_t max_source_length, size_t max_tokens,
                                           void *stream, slimt_hip_ctx **out) {
  if (!model || !out) return fail(-1, "null argument");
  *out = nullptr;
  if (max_batch == 0 || max_source_length == 0 || max_tokens == 0) return fail(-1, "empty workspace");
  if (max_source_length > 128)
    return fail(-1, "max_source_length %zu > 128 (slimt wraps at 128, Frontend.hh:27)", max_source_length);
  HIPCHK(hipSetDevice(model->device));
  auto *c = new slimt_hip_ctx();
  c->model = model;
  c->max_B = max_batch;
  c->max_S = max_source_length;
  c->max_M = std::min(max_batch * max_source_length, max_tokens);
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return fail((int)e, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
  }
  int rc = ctx_alloc(c);
  if (rc) {
    ctx_free(c);
    delete c;
    return rc;
  }
  count_context(model->device, +1);
  *out = c;
  return 0;
}

extern "C" int slimt_hip_ctx_destroy(slimt_hip_ctx *ctx) {
  if (!ctx) return 0;
  count_context(ctx->model->device, -1);
  (void)hipSetDevice(ctx->model->device);
  (void)hipStreamSynchronize(ctx->stream);
  {
    std::lock_guard<std::mutex> lock(ctx->model->gate_mu);
    auto &v = ctx->model->gate_ctx;
    for (size_t i = 0; i < v.size();)
      if (v[i].ctx == ctx) v.erase(v.begin() + (long)i); else ++i;
  }
  ctx_free(ctx);
  delete ctx;
  return 0;
}

extern "C" int slimt_hip_ctx_stream(slimt_hip_ctx *ctx, void **stream) {
  if (!ctx || !stream) return fail(-1, "null argument");
  *stream = reinterpret_cast<void *>(ctx->stream);
  return 0;
}

extern "C" int slimt_hip_ctx_set_decode_mode(slimt_hip_ctx *ctx, int mode) {
  if (!ctx) return fail(-1, "ctx is NULL");
  if (mode < 0 || mode > 6) return fail(-1, "bad decode mode %d", mode);
  ctx->decode_mode = mode;
  return 0;
}

extern "C" int slimt_hip_ctx_set_encode_rows(slimt_hip_ctx *ctx, int rows) {
  if (!ctx) return fail(-1, "ctx is NULL");
  if (rows != 0 && rows != 32 && rows != 64) return fail(-1, "encoder rows per workgroup %d not in {0, 32, 64}", rows);
  ctx->encode_rows = rows;
  return 0;
}

extern "C" int slimt_hip_ctx_plan(const slimt_hip_ctx *ctx, size_t S, int *encoder_fused,
                                  int *decoder_fused) {
  if (!ctx) return fail(-1, "ctx is NULL");
  const slimt_hip_model *m = ctx->model;
  const bool fused = fused_decoder_allowed(ctx);
  if (encoder_fused)
    *encoder_fused = fused && (fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S) ||
                               long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S));
  if (decoder_fused) *decoder_fused = fused && fused_decode_supported(m->D, m->F, m->H, m->Ld);
  return 0;
}

// A word of pinned host memory per context that kernels set when one of their bounded waits ran out (a member of a
// logits cluster that never arrived: 2; the in-launch shortlist's publisher never published: 1). Read where the host
// waits for the context's stream; the word is cleared and the call fails -- the batch's results are not to be used.
static unsigned *dev_error_word(slimt_hip_ctx *ctx) { return reinterpret_cast<unsigned *>(ctx->n_finished_host) + 2; }
static unsigned *dev_error_device_view(slimt_hip_ctx *ctx) {
  void *p = nullptr;
  if (!ctx->n_finished_host || hipHostGetDevicePointer(&p, dev_error_word(ctx), 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return static_cast<unsigned *>(p);
}
static int check_dev_error(slimt_hip_ctx *ctx) {
  if (!ctx->n_finished_host) return 0;
  volatile unsigned *w = dev_error_word(ctx);
  const unsigned e = *w;
  if (!e) return 0;
  *w = 0;
  return fail(-1, e == 1 ? "the batch's shortlist was never published inside the encoder launch (bounded wait ran out): results discarded"
                         : "a workgroup of a logits cluster never arrived (bounded wait ran out): results discarded");
}

extern "C" int slimt_hip_ctx_synchronize(slimt_hip_ctx *ctx) {
  if (!ctx) return fail(-1, "ctx is NULL");
  // A blocking-sync event, not hipStreamSynchronize: that one spins, and a host pipeline has one
  // thread per context waiting here for milliseconds -- a dozen spinning threads take the cores
  // the runtime's own threads and the batch builders need (the Service with 8 workers: 11.7 ->
  // 13.3 M tok/s once the waiters sleep).
  HIPCHK(hipSetDevice(ctx->model->device));  // the event is created and recorded on the context's device, whatever the caller's is
  if (!ctx->sync_event) HIPCHK(hipEventCreateWithFlags(&ctx->sync_event, hipEventBlockingSync | hipEventDisableTiming));
  HIPCHK(hipEventRecord(ctx->sync_event, ctx->stream));
  HIPCHK(hipEventSynchronize(ctx->sync_event));
  return check_dev_error(ctx);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
namespace {

EmbedArgs embed_args(const slimt_hip_ctx *c) {
  const slimt_hip_model *m = c->model;
  EmbedArgs e;
  e.wemb = m->wemb.as<int8_t>();
  e.inv_mult = 1 / m->wemb_mult;             // Io.cc:280-281
  e.sqrt_d = std::sqrt(static_cast<float>(m->D));  // Transformer.cc:34
  e.pos = c->pos.as<float>();
  e.D = m->D;
  e.V = m->V;
  return e;
}

// affine on M rows, f32 in -> f32 out
int run_affine_f32(slimt_hip_ctx *c, int family, const AffineW &w, const float *x, int M, float *y,
                   int rows_per_block, int kc_S = 0, int kc_dh = 0, bool raw_acc = false) {
  GemmArgs g;
  g.kc_S = kc_S;
  g.kc_dh = kc_dh;
  g.raw_acc = raw_acc;
  g.x_f32 = x;
  g.lda = w.w.K;
  g.M = M;
  g.w = w.w;
  g.y = y;
  g.ldy = w.w.N;
  ProfScope p(c, family, gemm_macs(M, w.w), gemm_bytes(w.w));
  HIPCHK(launch_gemm(g, EPI_PLAIN, rows_per_block, c->stream));
  return 0;
}

int run_affine_res_ln(slimt_hip_ctx *c, int family, const AffineW &w, const float *x_f32,
                      const int8_t *x_i8, int M, const float *res, const LnW &ln, float *y,
                      int rows_per_block) {
  GemmArgs g;
  g.x_f32 = x_f32;
  g.x_i8 = x_i8;
  g.lda = w.w.K;
  g.M = M;
  g.w = w.w;
  g.y = y;
  g.ldy = w.w.N;
  g.res = res;
  g.ldres = w.w.N;
  g.ln_scale = ln.scale.as<float>();
  g.ln_bias = ln.bias.as<float>();
  g.eps = 1e-6f;  // TensorOps.hh:67-68
  ProfScope p(c, family, gemm_macs(M, w.w), gemm_bytes(w.w));
  HIPCHK(launch_gemm(g, EPI_RES_LN, rows_per_block, c->stream));
  return 0;
}

// affine + residual on M rows (f32 or int8 in) -> f32 pre-LayerNorm sum
int run_affine_res(slimt_hip_ctx *c, int family, const AffineW &w, const float *x_f32,
                   const int8_t *x_i8, int M, const float *res, float *y, int rows_per_block) {
  GemmArgs g;
  g.x_f32 = x_f32;
  g.x_i8 = x_i8;
  g.lda = w.w.K;
  g.M = M;
  g.w = w.w;
  g.y = y;
  g.ldy = w.w.N;
  g.res = res;
  g.ldres = w.w.N;
  ProfScope p(c, family, gemm_macs(M, w.w), gemm_bytes(w.w));
  HIPCHK(launch_gemm(g, EPI_PLAIN, rows_per_block, c->stream));
  return 0;
}

int run_affine_relu_q(slimt_hip_ctx *c, int family, const AffineW &w, const float *x,
                      const int8_t *x_i8, int M, float a_quant_next, int8_t *y8,
                      int rows_per_block) {
  GemmArgs g;
  g.x_f32 = x;
  g.x_i8 = x_i8;
  g.lda = w.w.K;
  g.M = M;
  g.w = w.w;
  g.y_i8 = y8;
  g.ldy8 = w.w.N;
  g.a_quant_out = a_quant_next;
  ProfScope p(c, family, gemm_macs(M, w.w), gemm_bytes(w.w));
  HIPCHK(launch_gemm(g, EPI_RELU_Q, rows_per_block, c->stream));
  return 0;
}

int check_batch(const slimt_hip_ctx *c, size_t B, size_t S) {
  if (B == 0 || S == 0) return fail(-1, "empty batch");
  if (B > c->max_B || S > c->max_S || B * S > c->max_M)
    return fail(-1, "batch %zux%zu exceeds the context workspace %zux%zu (%zu padded tokens)", B, S,
                c->max_B, c->max_S, c->max_M);
  return 0;
}

// Model.cc:195-201: embedding + Encoder::forward. d_ids/d_len already in
// ctx->ids / ctx->lengths. Result in ctx->x0.
// embedded: ctx->x0 already holds the transformed embedding [B,S,D] (Encoder::forward's
// argument, Transformer.cc:57): the stage kernels run from there.
// D = 256: 64-row tiles (encode_tall.hip: the O projection's and the FFN's weights cross the
// CU's L2 path once per 64 rows) from 32 of them on: with other batches in flight they win
// already there (B = 128: 23.3 -> 26.2 M tok/s, B = 64: level), a lone batch of that size pays
// ~0.1 ms of latency; below that twice as many 32-row workgroups (and less padding) are better
bool tall_encoder_chosen(const slimt_hip_ctx *c, int B, int S) {
  const slimt_hip_model *m = c->model;
  const bool tall_mid = S > 32 && c->encode_rows != 32 && tall_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S);
  if (c->decode_mode == 1 || !(tall_mid || fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S))) return false;
  return tall_mid || (tall_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S) &&
                      (c->encode_rows == 64 || (c->encode_rows == 0 && tall_encode_grid(B, S, false) >= 32)));
}

// the encoders with a writer for the tight K/V form: every fused one for S <= 32 (64- and 32-row tiles at D = 256, the D = 512
// one), the 64-row one with a sentence of 33..64 tokens per workgroup, the per-sentence one for 65..128 tokens
static bool kv_tight_writer(const slimt_hip_ctx *c, int B, int S) {
  const slimt_hip_model *m = c->model;
  if (S > 64) return c->decode_mode != 1 && long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S);
  if (S > 32) return tall_encoder_chosen(c, B, S);
  return c->decode_mode != 1 && fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S);
}

// embedding + every encoder layer + the decoder's K/V cache in one launch of encode_tall / encode_fused / encode_wide
bool fused_encoder_chosen(const slimt_hip_ctx *c, int B, int S) {
  (void)B;
  const slimt_hip_model *m = c->model;
  const bool tall_mid = S > 32 && c->encode_rows != 32 && tall_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S);
  return c->decode_mode != 1 && (tall_mid || fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S));
}

// One merged launch pair (kernels.h, MergeIn / MergeOut / MergePack): the sub-batches' tables as the kernels take them, and
// the packing jobs of the launch -- one per distinct shortlist, their buffers job strides apart in ctx->out_sl.
struct MergePlan {
  int n = 0;
  MergeIn in[kMaxMerge];
  MergeOut out[kMaxMerge];
  int n_jobs = 0;
  MergePack jobs[kMaxMerge];
  int max_N = 0;  // columns of the widest job
  size_t stride_wp = 0, stride_cs = 0, stride_pb = 0;
  bool dense = false;  // one output layer for all sub-batches: no holes between them (kernels.h, FusedDecodeArgs::sub_dense)
};

// gen (nullable; only where fused_encoder_chosen): the batch's shortlist is generated inside the encoder launch
// (kernels.h, FusedEncodeArgs::gen) -- its ids / count are pack->idx / pack->n_dev.
int encode_device(slimt_hip_ctx *c, int B, int S, float *h_embed, float *h_layers,
                  const uint32_t *d_ids = nullptr, const uint32_t *d_lengths = nullptr,
                  const PackArgs *pack = nullptr, bool embedded = false, bool keep_out = true,
                  bool kv24 = false, const ShortlistArgs *gen = nullptr, bool kv_store_nt = false,
                  const MergePlan *mp = nullptr) {
  // kv24 (translate_device only: the caller decodes with the persistent kernel's packed-cache
  // variant right behind this launch): the fused encoder leaves the 24-bit K/V cache
  // keep_out = false (the translate path): the persistent encoder leaves the decoder its K/V
  // cache only -- the encoder output itself (B S D f32, 8 MB at the headline size) is not written
  const slimt_hip_model *m = c->model;
  hipStream_t st = c->stream;
  const int M = B * S, D = m->D;
  const size_t nbytes = (size_t)M * D * 4;
  c->kv_fmt_valid = false;  // set below by the one path that records the forms of a packed cache
  c->kv_tight = false;
  c->B = B;
  c->S = S;
  c->have_encoder_out = false;
  c->decode_ready = false;
  c->kv_ready = false;
  // sentences of 33..64 tokens: the 64-row D = 256 encoder takes one per workgroup (else the
  // per-sentence kernel below)
  const bool tall_mid = S > 32 && c->encode_rows != 32 && tall_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S);
  if (!embedded && c->decode_mode != 1 && (tall_mid || fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S))) {
    // embedding + every encoder layer + the decoder's K/V cache in one launch
    FusedEncodeArgs f;
    f.B = B; f.S = S; f.Le = m->Le; f.Ld = m->Ld;
    for (int l = 0; l < m->Le; ++l) {
      const EncLayerW &L = m->enc[(size_t)l];
      FusedEncLayerW &fl = f.L[l];
      fl.q = L.attn.q.w; fl.k = L.attn.k.w; fl.v = L.attn.v.w; fl.o = L.attn.o.w;
      fl.ffn1 = L.ffn1.w; fl.ffn2 = L.ffn2.w;
      fl.attn_ln_s = L.attn.ln.scale.as<float>(); fl.attn_ln_b = L.attn.ln.bias.as<float>();
      fl.ffn_ln_s = L.ffn_ln.scale.as<float>(); fl.ffn_ln_b = L.ffn_ln.bias.as<float>();
    }
    for (int l = 0; l < m->Ld; ++l) {
      f.dec_k[l] = m->dec[(size_t)l].attn.k.w;
      f.dec_v[l] = m->dec[(size_t)l].attn.v.w;
    }
    f.emb = embed_args(c);
    f.ids = d_ids ? d_ids : c->ids.as<uint32_t>();
    f.lengths = d_lengths ? d_lengths : c->lengths.as<uint32_t>();
    f.alpha = 1.0f / std::sqrt(static_cast<float>(m->D / m->H));
    f.kv = c->kv.as<float>();
    f.kv24 = kv24;
    f.kv_store_nt = kv_store_nt;
    if (kv24 && (size_t)M * D * 3 >= (1u << 31)) return fail(-1, "packed K/V cache: %d rows exceed a 2 GB plane", M);
    // the narrow (20-bit) form where the writer and the reader have it -- D = 256 or 512, S <= 32 -- and where a
    // sentence's V block (groups of eight keys) fits the slot of its 24-bit form (groups of four): not S = 1..4, 9..12
    // (... and S = 33..64 at D = 256: the 64-row encoder with one sentence per workgroup, attention_packed64 with Form20)
    c->kv_fmt_valid = kv24 && ((D == 256 && S <= 64) || (D == 512 && S <= 32)) &&
                      ((S + 7) / 8) * 5120 <= ((S + 3) / 4) * 3072 && kv_narrow_wanted(c->model, &f.kv_wide_count);
#ifdef SLIMT_EXP_NO_KV20  // A/B builds: the 24-bit form only
    c->kv_fmt_valid = false;
#endif
    if (c->kv_fmt_valid) {
      HIPCHK(c->kv_fmt.reserve((size_t)m->Ld * c->max_B));
      f.kv_fmt = c->kv_fmt.as<unsigned char>();
      f.kv_narrow_limit = std::min(m->kv_narrow_limit, 1 << 19);
      c->kv_fmt_B = B;
      c->model->kv_layers_submitted.fetch_add((unsigned long long)B * m->Ld, std::memory_order_relaxed);
      f.kv_tight_layers = kv_tight_wanted(c, S, kv_tight_writer(c, B, S), &f.kv_not16_count);
      if (f.kv_tight_layers) {
        f.kv_tight_limit = std::min(m->kv_tight_limit, 1 << 15);
        for (int l = 0; l < m->Ld; ++l)
          for (int p = 0; p < 2; ++p) f.kv_centre[l][p] = m->kv_centre_of(c->kv_gen, l, p);
        c->kv_tight = true;
        for (int l = 0; l < m->Ld; ++l)
          if ((f.kv_tight_layers >> l) & 1u) c->model->kv_tight_submitted[l].fetch_add((unsigned long long)B, std::memory_order_relaxed);
      }
    }
    f.enc_out = keep_out ? c->x0.as<float>() : nullptr;
    if (pack) {
      f.pack = *pack;
      f.pack_tiles = (pack->N + 15) / 16;
    }
    if (mp) {  // a merged launch: the sub-batches' inputs, and one packing job per distinct shortlist
      f.n_sub = mp->n;
      for (int j = 0; j < mp->n; ++j) f.sub[j] = mp->in[j];
      if (pack) {
        f.n_pack = mp->n_jobs;
        for (int j = 0; j < mp->n_jobs; ++j) f.pjob[j] = mp->jobs[j];
        f.pack_tiles = (mp->max_N + 15) / 16;
        f.pack_stride_wp = mp->stride_wp;
        f.pack_stride_cs = mp->stride_cs;
        f.pack_stride_pb = mp->stride_pb;
      }
    }
    if (h_embed) {
      HIPCHK(c->dbg_embed.reserve(nbytes));
      f.embed_out = c->dbg_embed.as<float>();
    }
    if (h_layers) {
      HIPCHK(c->dbg_layers.reserve(nbytes * (size_t)m->Le));
      f.layer_out = c->dbg_layers.as<float>();
    }
    f.trace = g_occ_trace;
    if (c->stamp_step >= 0 && c->stamps.p) {  // encoder stamps live in slots 48..55
      f.stamps = c->stamps.as<unsigned long long>() + 48;
      f.stamp_layer = c->stamp_step < m->Le ? c->stamp_step : m->Le - 1;
    }
    f.ticket = c->ticket.as<unsigned>() + 1;  // over-subscribed launch, tiles claimed by ticket
    f.ticket_base = c->enc_ticket_base;
    const bool tall = tall_encoder_chosen(c, B, S);
    if (gen) {
      if (!pack) return fail(-1, "in-launch shortlist generation needs a packing job");
      HIPCHK(c->gen_flag.reserve(64));
      if (c->gen_epoch == 0) HIPCHK(hipMemsetAsync(c->gen_flag.p, 0, 64, st));
      f.gen = *gen;
      f.gen_flag = c->gen_flag.as<unsigned>();
      f.gen_epoch = ++c->gen_epoch;
      f.dev_error = dev_error_device_view(c);
      f.gen_spin_limit = c->gen_spin_limit;
      if (c->gen_break) f.gen_wait_xor = 0x40000000u;  // (debug: the waiters look for an epoch nobody publishes)
      if (c->gen_epoch == 0xffffffffu) c->gen_epoch = 0;  // (the flag is cleared again before epoch 1 is reused)
    }
    {
      const double macs = (double)M * (m->Le * (4.0 * D * D + 2.0 * D * m->F) + m->Ld * 2.0 * D * D);
      ProfScope p(c, SLIMT_HIP_K_ENCODE_FUSED, macs, 0);
      if (tall)
        HIPCHK(launch_encode_tall(f, m->F, st));
      else
        HIPCHK(launch_encode_fused(f, m->D, m->F, m->H, st));
    }
    c->enc_ticket_base += (unsigned)(tall ? tall_encode_grid(B, S, true) : fused_encode_grid(B, S, true));
    if (h_embed) HIPCHK(hipMemcpyAsync(h_embed, c->dbg_embed.p, nbytes, hipMemcpyDeviceToHost, st));
    if (h_layers)
      HIPCHK(hipMemcpyAsync(h_layers, c->dbg_layers.p, nbytes * (size_t)m->Le, hipMemcpyDeviceToHost, st));
    c->have_encoder_out = keep_out;
    c->kv_ready = !kv24;  // the step-wise decoder reads the f32 form only
    return 0;
  }
  if (mp) return fail(-1, "merged launch: no persistent encoder for this shape");
  if (!embedded && c->decode_mode != 1 && long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, S)) {
    // 32 < S <= 128: one persistent workgroup per sentence (kernels.hip, encode_long_kernel)
    LongEncodeArgs a;
    FusedEncodeArgs &f = a.f;
    f.B = B; f.S = S; f.Le = m->Le; f.Ld = m->Ld;
    for (int l = 0; l < m->Le; ++l) {
      const EncLayerW &L = m->enc[(size_t)l];
      FusedEncLayerW &fl = f.L[l];
      fl.q = L.attn.q.w; fl.k = L.attn.k.w; fl.v = L.attn.v.w; fl.o = L.attn.o.w;
      fl.ffn1 = L.ffn1.w; fl.ffn2 = L.ffn2.w;
      fl.attn_ln_s = L.attn.ln.scale.as<float>(); fl.attn_ln_b = L.attn.ln.bias.as<float>();
      fl.ffn_ln_s = L.ffn_ln.scale.as<float>(); fl.ffn_ln_b = L.ffn_ln.bias.as<float>();
    }
    for (int l = 0; l < m->Ld; ++l) {
      f.dec_k[l] = m->dec[(size_t)l].attn.k.w;
      f.dec_v[l] = m->dec[(size_t)l].attn.v.w;
    }
    f.emb = embed_args(c);
    f.ids = d_ids ? d_ids : c->ids.as<uint32_t>();
    f.lengths = d_lengths ? d_lengths : c->lengths.as<uint32_t>();
    f.alpha = 1.0f / std::sqrt(static_cast<float>(m->D / m->H));
    f.kv = c->kv.as<float>();
    f.kv24 = kv24;
    // (the narrow form of the packed cache: the first path above has the note; here every S up to 128)
    c->kv_fmt_valid = kv24 && D == 256 && ((S + 7) / 8) * 5120 <= ((S + 3) / 4) * 3072 &&
                      kv_narrow_wanted(c->model, &f.kv_wide_count);
#ifdef SLIMT_EXP_NO_KV20
    c->kv_fmt_valid = false;
#endif
    if (c->kv_fmt_valid) {
      HIPCHK(c->kv_fmt.reserve((size_t)m->Ld * c->max_B));
      f.kv_fmt = c->kv_fmt.as<unsigned char>();
      f.kv_narrow_limit = std::min(m->kv_narrow_limit, 1 << 19);
      c->kv_fmt_B = B;
      c->model->kv_layers_submitted.fetch_add((unsigned long long)B * m->Ld, std::memory_order_relaxed);
      f.kv_tight_layers = kv_tight_wanted(c, S, kv_tight_writer(c, B, S), &f.kv_not16_count);
      if (f.kv_tight_layers) {
        f.kv_tight_limit = std::min(m->kv_tight_limit, 1 << 15);
        for (int l = 0; l < m->Ld; ++l)
          for (int p = 0; p < 2; ++p) f.kv_centre[l][p] = m->kv_centre_of(c->kv_gen, l, p);
        c->kv_tight = true;
        for (int l = 0; l < m->Ld; ++l)
          if ((f.kv_tight_layers >> l) & 1u) c->model->kv_tight_submitted[l].fetch_add((unsigned long long)B, std::memory_order_relaxed);
      }
    }
    f.enc_out = c->x0.as<float>();
    if (pack) {
      f.pack = *pack;
      f.pack_tiles = (pack->N + 15) / 16;
    }
    if (h_embed) {
      HIPCHK(c->dbg_embed.reserve(nbytes));
      f.embed_out = c->dbg_embed.as<float>();
    }
    if (h_layers) {
      HIPCHK(c->dbg_layers.reserve(nbytes * (size_t)m->Le));
      f.layer_out = c->dbg_layers.as<float>();
    }
    a.H = m->H; a.D = m->D; a.F = m->F;
    a.x = c->x0.as<float>(); a.y = c->x1.as<float>();
    a.q = c->q.as<float>(); a.k = c->k.as<float>(); a.v = c->v.as<float>();
    a.att = c->att.as<float>();
    a.h8 = c->h8.as<int8_t>();
    {
      const double macs = (double)M * (m->Le * (4.0 * D * D + 2.0 * D * m->F) + m->Ld * 2.0 * D * D);
      ProfScope p(c, SLIMT_HIP_K_ENCODE_FUSED, macs, 0);
      HIPCHK(launch_encode_long(a, st));
    }
    if (h_embed) HIPCHK(hipMemcpyAsync(h_embed, c->dbg_embed.p, nbytes, hipMemcpyDeviceToHost, st));
    if (h_layers)
      HIPCHK(hipMemcpyAsync(h_layers, c->dbg_layers.p, nbytes * (size_t)m->Le, hipMemcpyDeviceToHost, st));
    c->have_encoder_out = true;
    c->kv_ready = !kv24;  // the step-wise decoder reads the f32 form only
    return 0;
  }
  float *x = c->x0.as<float>(), *y = c->x1.as<float>();
  if (!embedded) HIPCHK(launch_embed_encoder(embed_args(c), c->ids.as<uint32_t>(), B, S, x, st));
  if (h_embed) HIPCHK(hipMemcpyAsync(h_embed, x, nbytes, hipMemcpyDeviceToHost, st));
  const int rpb = M >= 2048 ? 64 : (M >= 512 ? 32 : 16);
  const int rpb_ln = M >= 4096 ? 32 : 16;
  // measured at M = 8192, K = N = 512 (base): 64-row blocks 20.9 us, 32-row blocks
  // 14.5 us (twice the blocks in flight); FFN1 (N = 2048) prefers 64 rows
  const int rpb_n = M >= 2048 ? 32 : rpb;
  const bool big = M >= 2048 && (D == 64 || D == 128 || D == 256 || D == 512);
  for (int l = 0; l < m->Le; ++l) {
    const EncLayerW &L = m->enc[(size_t)l];
    // Attention::forward (Modules.cc:287-319)
    RCCHK(run_affine_f32(c, SLIMT_HIP_K_GEMM_ENC, L.attn.q, x, M, c->q.as<float>(), rpb_n));
    RCCHK(run_affine_f32(c, SLIMT_HIP_K_GEMM_ENC, L.attn.k, x, M, c->k.as<float>(), rpb_n));
    RCCHK(run_affine_f32(c, SLIMT_HIP_K_GEMM_ENC, L.attn.v, x, M, c->v.as<float>(), rpb_n));
    AttnArgs a;
    a.q = c->q.as<float>(); a.k = c->k.as<float>(); a.v = c->v.as<float>();
    a.ldq = a.ldk = a.ldv = a.ldo = D;
    a.lengths = c->lengths.as<uint32_t>();
    a.B = B; a.H = m->H; a.Tq = S; a.S = S; a.dh = D / m->H;
    a.alpha = 1.0f / std::sqrt(static_cast<float>(a.dh));
    a.out = c->att.as<float>();
    {
      ProfScope p(c, SLIMT_HIP_K_ATTN_ENC, 0, 0);
      HIPCHK(launch_attention(a, st));
    }
    if (big) {
      // many rows: residual sums from narrow GEMM blocks (no block has to own whole
      // rows), LayerNorm as a pass of its own that also writes FFN1's int8 operand
      RCCHK(run_affine_res(c, SLIMT_HIP_K_GEMM_ENC, L.attn.o, c->att.as<float>(), nullptr, M, x, y, rpb_n));
      HIPCHK(launch_layer_norm_q(y, L.attn.ln.scale.as<float>(), L.attn.ln.bias.as<float>(), 1e-6f, M, D,
                                 y, c->a8.as<int8_t>(), L.ffn1.w.a_quant, st));
      RCCHK(run_affine_relu_q(c, SLIMT_HIP_K_GEMM_ENC, L.ffn1, nullptr, c->a8.as<int8_t>(), M,
                              L.ffn2.w.a_quant, c->h8.as<int8_t>(), rpb));
      RCCHK(run_affine_res(c, SLIMT_HIP_K_GEMM_ENC, L.ffn2, nullptr, c->h8.as<int8_t>(), M, y, x, rpb_n));
      HIPCHK(launch_layer_norm_q(x, L.ffn_ln.scale.as<float>(), L.ffn_ln.bias.as<float>(), 1e-6f, M, D, x,
                                 nullptr, 0.0f, st));
    } else {
      RCCHK(run_affine_res_ln(c, SLIMT_HIP_K_GEMM_ENC, L.attn.o, c->att.as<float>(), nullptr, M, x,
                              L.attn.ln, y, rpb_ln));
      // FFN (Modules.cc:326-331): y -> x
      RCCHK(run_affine_relu_q(c, SLIMT_HIP_K_GEMM_ENC, L.ffn1, y, nullptr, M, L.ffn2.w.a_quant,
                              c->h8.as<int8_t>(), rpb));
      RCCHK(run_affine_res_ln(c, SLIMT_HIP_K_GEMM_ENC, L.ffn2, nullptr, c->h8.as<int8_t>(), M, y,
                              L.ffn_ln, x, rpb_ln));
    }
    if (h_layers)
      HIPCHK(hipMemcpyAsync(h_layers + (size_t)l * M * D, x, nbytes, hipMemcpyDeviceToHost, st));
  }
  c->have_encoder_out = true;
  return 0;
}

// per-batch decoder setup: cross-attention K/V of the encoder output (computed
// ONCE instead of every step, Modules.cc:248), shortlist gather, start states
// (Transformer.cc:78-85).
int decode_setup(slimt_hip_ctx *c, size_t n_sl) {
  const slimt_hip_model *m = c->model;
  hipStream_t st = c->stream;
  const int B = c->B, S = c->S, M = B * S, D = m->D;
  if (!c->have_encoder_out) return fail(-1, "decode before encode");
  const int rpb = M >= 512 ? 32 : 16;
  float *kv = c->kv.as<float>();
  for (int l = 0; l < m->Ld && !c->kv_ready; ++l) {
    const DecLayerW &L = m->dec[(size_t)l];
    // K in the coalescing-friendly cache layout [sentence][head][d/4][key][4]; both as float(accS): the
    // attention applies the projections' u and pb after its sums (kernels.h, kv24)
    RCCHK(run_affine_f32(c, SLIMT_HIP_K_GEMM_ENC, L.attn.k, c->x0.as<float>(), M,
                         kv + (size_t)(2 * l) * M * D, rpb, S, D / m->H, true));
    RCCHK(run_affine_f32(c, SLIMT_HIP_K_GEMM_ENC, L.attn.v, c->x0.as<float>(), M,
                         kv + (size_t)(2 * l + 1) * M * D, rpb, 0, 0, true));
  }
  c->n_sl = (int)n_sl;
  if (n_sl) {
    // affine_with_select's SelectColumnsB + bias gather (Intgemm.inl.cc:48-69),
    // hoisted out of the step loop: the shortlist is fixed per batch (Model.cc:117-120)
    RCCHK(prepare_affine(c->out_sl, m->out_raw.as<int8_t>(), D, (int)n_sl,
                         c->shortlist.as<uint32_t>(), m->out_bias.as<float>(), m->out_a_quant,
                         m->wemb_mult, st));
  }
  HIPCHK(hipMemsetAsync(c->state.p, 0, (size_t)m->Ld * B * D * 4, st));
  c->decode_ready = true;
  return 0;
}

const AffineW &output_layer(const slimt_hip_ctx *c) {
  return c->n_sl ? c->out_sl : c->model->out_full;
}

// The last decoder LayerNorm is applied by the consumer of ctx->dx_pre (the
// logits GEMM, or the next layer's SSRU): rows of the final decoder state.
RowSrc decoder_out_rows(const slimt_hip_ctx *c) {
  const DecLayerW &L = c->model->dec.back();
  RowSrc r;
  r.x = c->dx_pre.as<float>();
  r.ln_scale = L.ffn_ln.scale.as<float>();
  r.ln_bias = L.ffn_ln.bias.as<float>();
  return r;
}

// DecoderLayer::forward x Ld (Modules.cc:237-259). Input: the step's target
// embedding in ctx->dx; output: pre-LayerNorm rows in ctx->dx_pre.
int decoder_layers(slimt_hip_ctx *c, float *d_align, int Tmax, const uint32_t *d_out_len,
                   float *d_attn_dbg) {
  const slimt_hip_model *m = c->model;
  hipStream_t st = c->stream;
  const int B = c->B, S = c->S, M = B * S, D = m->D;
  float *kv = c->kv.as<float>();
  for (int l = 0; l < m->Ld; ++l) {
    const DecLayerW &L = m->dec[(size_t)l];
    // layer input rows: embedding (layer 0) or LN(previous layer's pre-LN rows)
    RowSrc xin;
    if (l == 0) {
      xin.x = c->dx.as<float>();
    } else {
      const DecLayerW &P = m->dec[(size_t)l - 1];
      xin.x = c->dx_pre.as<float>();
      xin.ln_scale = P.ffn_ln.scale.as<float>();
      xin.ln_bias = P.ffn_ln.bias.as<float>();
    }
    DSsruArgs s;
    s.B = B; s.D = D;
    s.x = xin;
    s.wf = L.rnn_f.w; s.w = L.rnn_w.w;
    s.state = c->state.as<float>() + (size_t)l * B * D;
    s.h_pre = c->dh.as<float>();
    {
      ProfScope p(c, SLIMT_HIP_K_SSRU, 2.0 * B * D * D, 2.0 * D * D);
      HIPCHK(launch_dssru(s, st));
    }
    RowSrc h;  // LN(x + relu(c')), Modules.cc:230
    h.x = c->dh.as<float>();
    h.ln_scale = L.rnn_ln.scale.as<float>();
    h.ln_bias = L.rnn_ln.bias.as<float>();
    // Attention::forward (Modules.cc:287-319): Q projection + SDPA over the cached K/V
    DQAttnArgs a;
    a.B = B; a.D = D; a.H = m->H; a.S = S;
    a.x = h;
    a.wq = L.attn.q.w;
    a.k = kv + (size_t)(2 * l) * M * D;
    a.v = kv + (size_t)(2 * l + 1) * M * D;
    a.ldv = D;
    a.uk = L.attn.k.w.u;
    a.uv = L.attn.v.w.u;
    a.pbk = L.attn.k.w.pb;
    a.pbv = L.attn.v.w.pb;
    a.lengths = c->lengths.as<uint32_t>();
    a.alpha = 1.0f / std::sqrt(static_cast<float>(D / m->H));
    a.literal = m->kv_format == 3;
    a.out_i8 = c->datt8.as<int8_t>();
    a.a_quant_out = L.attn.o.w.a_quant;
    if (l + 1 == m->Ld) {  // alignment = last layer (Transformer.cc:165-174)
      a.attn = d_attn_dbg;
      if (d_align) {
        a.align = d_align;
        a.out_len = d_out_len;
        a.finished = c->finished.as<uint8_t>();
        a.Tmax = Tmax;
      }
    }
    {
      ProfScope p(c, SLIMT_HIP_K_ATTN_DEC, (double)B * D * D, (double)D * D);
      HIPCHK(launch_dqattn(a, st));
    }
    // O projection + residual h (Modules.cc:308-314) -> pre-LN rows
    DGemmArgs o;
    o.B = B; o.D = D;
    o.a_i8 = c->datt8.as<int8_t>();
    o.w = L.attn.o.w;
    o.res = h;
    o.y = c->dout.as<float>();
    o.ldy = D;
    {
      ProfScope p(c, SLIMT_HIP_K_GEMM_DEC, gemm_macs(B, o.w), gemm_bytes(o.w));
      HIPCHK(launch_dgemm(o, EPI_PLAIN, st));
    }
    RowSrc ao;  // LN(h + O(...)), Modules.cc:316
    ao.x = c->dout.as<float>();
    ao.ln_scale = L.attn.ln.scale.as<float>();
    ao.ln_bias = L.attn.ln.bias.as<float>();
    // FFN (Modules.cc:251-257)
    DGemmArgs f1;
    f1.B = B; f1.D = D;
    f1.a = ao;
    f1.w = L.ffn1.w;
    f1.y_i8 = c->df8.as<int8_t>();
    f1.ldy8 = m->F;
    f1.a_quant_out = L.ffn2.w.a_quant;
    {
      ProfScope p(c, SLIMT_HIP_K_GEMM_DEC, gemm_macs(B, f1.w), gemm_bytes(f1.w));
      HIPCHK(launch_dgemm(f1, EPI_RELU_Q, st));
    }
    DGemmArgs f2;
    f2.B = B; f2.D = D;
    f2.a_i8 = c->df8.as<int8_t>();
    f2.w = L.ffn2.w;
    f2.res = ao;
    f2.y = c->dx_pre.as<float>();
    f2.ldy = D;
    {
      ProfScope p(c, SLIMT_HIP_K_GEMM_DEC, gemm_macs(B, f2.w), gemm_bytes(f2.w));
      HIPCHK(launch_dgemm(f2, EPI_PLAIN, st));
    }
  }
  return 0;
}

// d_ids / d_lengths / d_shortlist: device pointers (the caller's, or the
// context's staging buffers). When both persistent kernels apply, a translate
// call is exactly two launches: [encoder + K/V cache + shortlist packing] and
// [decode loop]; nothing is copied or memset in between (small helper kernels
// queue behind the long-running decoders of the other workers).
int translate_device(slimt_hip_ctx *c, const uint32_t *d_ids, const uint32_t *d_lengths,
                     const uint32_t *d_shortlist, size_t B, size_t S, size_t n_sl,
                     float limit_factor, uint32_t eos_id, uint32_t *d_out_ids, uint32_t *d_out_len,
                     float *d_align, int steps_hint, const uint32_t *d_n_sl = nullptr,
                     float *align_out = nullptr, size_t n_sl_hint = 0, const ShortlistArgs *gen = nullptr,
                     MergePlan *mp = nullptr) {
  // mp (merged launch; lean path only, checked by the caller): B = the launch's global sentences, n_sl = the widest job's
  // columns (0 = full vocabulary for every sub-batch); d_ids / d_lengths / d_shortlist / d_out_* / d_align are not used
  // gen (with d_n_sl, lean path, 64-row encoder): the shortlist is generated inside the encoder launch
  // n_sl_hint (with d_n_sl): what the host expects the device-side shortlist size to be (the size of
  // this context's previous generated shortlist): tuning decisions only
  // align_out != nullptr (persistent decoder only): d_align is a staging buffer in device memory and
  // the decoder copies each sentence's rows from there to align_out when its loop ends (see
  // FusedDecodeArgs::align_out)
  // d_n_sl != nullptr: the shortlist was generated on this stream; its size is on
  // the device and n_sl is only the capacity of d_shortlist (persistent kernels only)
  const slimt_hip_model *m = c->model;
  hipStream_t st = c->stream;
  // Model.cc:144-161: the first step is unconditional (one token is always recorded), the
  // loop then runs while i < (size_t)(limit_factor * S): at least one output column
  const size_t Tmax = std::max<size_t>(1, (size_t)(limit_factor * (float)S));
  const bool fused_dec = fused_decoder_allowed(c) && fused_decode_supported(m->D, m->F, m->H, m->Ld);
  const bool lean = fused_dec && (fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S) ||
                                  long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S));
  // packed 24-bit K/V cache: fused encoder -> fused decoder, D = 256 / d_head 32 or D = 512 / d_head 64, S <= 32
  // (V is cached in groups of four keys: S = 1, 2, 5 would not fit the f32 form's plane)
  // ... and for 33..64-token sentences of the D = 256 / F = 1536 shape (64-row encoder, one sentence per workgroup)
  const bool kv24_mid = S > 32 && S <= 64 && c->encode_rows != 32 && c->decode_mode != 3 &&
                        fused_decode_mid_supported(m->D, m->F, m->H, m->Ld) &&
                        tall_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S);
  // ... and for 65..128-token sentences of that shape (the per-sentence encoder, attention_packed128 with Form24)
  const bool kv24_long = S > 64 && S <= 128 && c->decode_mode != 3 && fused_decode_long24_supported(m->D, m->F, m->H, m->Ld) &&
                         long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S);
  const bool kv_packed = lean && (m->kv_format == 0 || m->kv_format == 2) &&
                    ((m->D == 256 && m->D / m->H == 32) || (m->D == 512 && m->D / m->H == 64 && m->F == 2048)) &&
                    ((S <= 32 && ((S + 3) & ~(size_t)3) * 3 <= S * 4 &&
                      fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S)) || kv24_mid || kv24_long);
  // The tight form's centres (engine.h, kv_centre): the first batch of enough rows that could take the form is cached as
  // f32 instead, and its column means become the centres (behind the encoder, on this stream).
  bool calibrate = false;
  if (kv_packed && c->model->kv_centre_state.load(std::memory_order_acquire) == 0 && B * S >= 1024 &&
      kv_tight_shape(c, (int)S, kv_tight_writer(c, (int)B, (int)S)) &&
      !c->model->kv_centre_claimed.exchange(true, std::memory_order_acq_rel))
    calibrate = true;
  // (a call that claimed the calibration and leaves before its reduction is queued -- an error on the way -- gives the claim back)
  struct ClaimGuard {
    slimt_hip_model *m;
    bool armed;
    ~ClaimGuard() {
      if (armed && m->kv_centre_state.load(std::memory_order_acquire) == 0) m->kv_centre_claimed.store(false, std::memory_order_release);
    }
  } claim_guard{c->model, calibrate};
  const bool kv24 = kv_packed && !calibrate;
  // One thread at a time queues a persistent translate (a few runtime calls, ~50 us): the runtime
  // serialises launches internally anyway, and a dozen worker threads contending inside it take
  // far longer per call than the same calls made one after the other (Service, 10 workers: 2.1 ms
  // per launch call against 0.05 ms with two).
  static const bool submit_lock = !(std::getenv("SLIMT_SUBMIT_LOCK") && std::getenv("SLIMT_SUBMIT_LOCK")[0] == '0');
  std::unique_lock<std::mutex> submit(c->model->submit_mu, std::defer_lock);
  StageClock clk;
  if (lean && submit_lock) submit.lock();
  clk.lap(2);
  if (g_timing.on) g_timing.calls += 1;
  unsigned long long call_slot = 0;
  int call_k = 8;
  DecodeState ds;
  ds.prev = c->prev.as<uint32_t>();
  ds.out_ids = d_out_ids;
  ds.out_len = d_out_len;
  ds.finished = c->finished.as<uint8_t>();
  ds.n_finished = c->n_finished.as<int>();
  ds.Tmax = (int)Tmax;
  ds.eos = eos_id;
  if (lean) {
    PackArgs job;
    if (n_sl && mp) {
      // one allocation for the launch's jobs, job j at j strides (a stride holds the widest job)
      mp->stride_wp = (packed_weight_bytes(m->D, (int)n_sl) + 255) / 256 * 256;
      mp->stride_cs = (colsum_alloc_bytes((int)n_sl) + 255) / 256 * 256;
      mp->stride_pb = (((size_t)n_sl + 15) / 16 * 16 * sizeof(float) + 255) / 256 * 256;
      HIPCHK(c->out_sl.Wp.reserve(mp->stride_wp * mp->n_jobs));
      HIPCHK(c->out_sl.colsum.reserve(mp->stride_cs * mp->n_jobs));
      HIPCHK(c->out_sl.pb.reserve(mp->stride_pb * mp->n_jobs));
      RCCHK(prepare_affine_meta(c->out_sl, m->out_raw.as<int8_t>(), m->D, mp->jobs[0].N, mp->jobs[0].idx,
                                m->out_bias.as<float>(), m->out_a_quant, m->wemb_mult, job, m->V));
    } else if (n_sl) {
      RCCHK(prepare_affine_meta(c->out_sl, m->out_raw.as<int8_t>(), m->D, (int)n_sl, d_shortlist,
                                m->out_bias.as<float>(), m->out_a_quant, m->wemb_mult, job, m->V));
    }
    job.n_dev = d_n_sl;
    // the decoder's cache policy of THIS call (decided further down, under the admission lock, from the same two
    // numbers): will its caches be kept in the Infinity Cache, or streamed? Streamed ones are also WRITTEN past it
    call_slot = c->model->kv_call_seq.fetch_add(1, std::memory_order_relaxed);
    call_k = c->model->kv_k_last.load(std::memory_order_relaxed);
    static const bool store_nt = std::getenv("SLIMT_KV_STORE_NT") && std::getenv("SLIMT_KV_STORE_NT")[0] == '1';
    RCCHK(encode_device(c, (int)B, (int)S, nullptr, nullptr, d_ids, d_lengths, n_sl ? &job : nullptr, false, false,
                        kv24, gen, store_nt && kv24 && call_k < 8 && (int)(call_slot % 8) >= call_k, mp));
    c->n_sl = (int)n_sl;
    if (calibrate) {
      slimt_hip_model *gm = c->model;
      int rc = 0;
      DevBuf &centres = gm->kv_centre[gm->kv_gen.load(std::memory_order_relaxed)];  // this generation's own buffer
      if (centres.reserve((size_t)m->Ld * 2 * m->D * 4) != hipSuccess || gm->kv_centre_sums.reserve((size_t)m->Ld * 2 * m->D * 8) != hipSuccess ||
          (!gm->kv_centre_ev && hipEventCreateWithFlags(&gm->kv_centre_ev, hipEventDisableTiming) != hipSuccess) ||
          launch_kv_centres(c->kv.as<float>(), m->Ld, (int)B, (int)S, m->D, gm->kv_centre_sums.as<unsigned long long>(), centres.as<int>(), st) != hipSuccess ||
          hipEventRecord(gm->kv_centre_ev, st) != hipSuccess)
        rc = 1;
      if (rc)  // (no centres: the next suitable batch tries again -- claim_guard --; this one goes on with its f32 cache)
        (void)hipGetLastError();
      else
        gm->kv_centre_state.store(1, std::memory_order_release);
    }
    clk.lap(1);
  } else {
    if (d_ids != c->ids.as<uint32_t>())
      HIPCHK(hipMemcpyAsync(c->ids.p, d_ids, B * S * 4, hipMemcpyDeviceToDevice, st));
    if (d_lengths != c->lengths.as<uint32_t>())
      HIPCHK(hipMemcpyAsync(c->lengths.p, d_lengths, B * 4, hipMemcpyDeviceToDevice, st));
    if (n_sl && d_shortlist != c->shortlist.as<uint32_t>()) {
      c->sl_host.clear();  // ctx->shortlist no longer holds what translate_host uploaded last
      HIPCHK(hipMemcpyAsync(c->shortlist.p, d_shortlist, n_sl * 4, hipMemcpyDeviceToDevice, st));
    }
    d_lengths = c->lengths.as<uint32_t>();
    d_shortlist = c->shortlist.as<uint32_t>();
    RCCHK(encode_device(c, (int)B, (int)S, nullptr, nullptr));
    RCCHK(decode_setup(c, n_sl));
    HIPCHK(hipMemsetAsync(d_out_ids, 0, B * (Tmax ? Tmax : 1) * 4, st));
    HIPCHK(hipMemsetAsync(d_out_len, 0, B * 4, st));
    HIPCHK(hipMemsetAsync(c->finished.p, 0, B, st));
    HIPCHK(hipMemsetAsync(c->n_finished.p, 0, 16, st));
    if (d_align) HIPCHK(hipMemsetAsync(d_align, 0, B * Tmax * S * 4, st));
  }
  ds.shortlist = n_sl ? d_shortlist : nullptr;
  const AffineW &out = output_layer(c);
  ds.pb0 = out.w.pb;
  ds.u_out = out.w.u;
  if (fused_decoder_allowed(c) && fused_decode_supported(m->D, m->F, m->H, m->Ld)) {
    // the whole greedy loop in one persistent launch (decode_fused.hip)
    FusedDecodeArgs f;
    f.B = (int)B; f.S = (int)S; f.Ld = m->Ld;
    f.max_steps = steps_hint > 0 ? steps_hint : (int)(Tmax > 1 ? Tmax : 1);
    f.Tmax = (int)Tmax;
    // 32 sentences per workgroup (where the kernel has it) once the output layer dominates the
    // weights a step streams: with the full 32k vocabulary it is 8 of 10 MB per workgroup and step, and
    // halving it per sentence beats the longer attention chain (B = 512, full vocabulary: 20.4 -> 23-24 M
    // tok/s; at 16k columns the two are level, below that 16 rows win)
    const size_t n_expected = d_n_sl && n_sl_hint ? n_sl_hint : mp ? (size_t)(n_sl ? mp->max_N : m->V) : (size_t)out.w.N;
    // ... and 8 or 4 (decode_fused.hip, SPW) when the decoders in flight would leave most of the chip idle:
    // decided below, under the admission lock, from the contexts that have a decoder pending
    // Round 5: where the kernel has it, output layers that wide are SHARED by clusters of four 16-sentence workgroups
    // instead (decode_fused.hip, CL: each member streams a quarter of the columns for all 64 sentences) -- under the
    // decoder admission only: the members wait for each other, and the admission is what guarantees every admitted
    // workgroup a CU without waiting for another decoder (decode mode 6 forces clusters, 3 the 32-sentence tiling)
    const bool cluster_ok = kv24 && m->D == 256 && m->F == 1536 && S <= 32 && c->model->decoder_budget > 0;
    // Measured on config 4 (B = 512, 32,000 columns, 20 workers; profiles/r05_cluster_logits.txt): clusters 23.4 M tok/s,
    // the 32-sentence tiling 24.9 M, the 16-sentence one 22.0 M -- a member's quarter of the columns takes 27 us + 14 us of
    // skew between its waves + 6 us of hand-overs where the whole layer took 70, but the 32-sentence tiling's per-sentence
    // cost is lower still: the phase is bound by the arg-max epilogue's issue slots and the address path's load
    // instructions (six per column tile here, four and a half there), not by the bytes the split saves. So: on request only.
    // a batch whose encoder was allowed the tight cache form: the tilings with its reader, whatever is asked for now (the
    // encoder is only allowed it when this context's decoders take those: kv_tight_wanted; a mode changed between the two
    // calls, or a first large output layer, ends up here)
    const bool tight = kv24 && c->kv_tight && c->kv_fmt_valid && c->kv_fmt_B == (int)B;
    const bool clusters = cluster_ok && c->decode_mode == 6 && !tight && !mp;
    c->expect_large_output = n_expected > 16384;
    f.kv_tight = tight;
    f.rows_per_wg = clusters ? 16 : c->decode_mode == 2 ? 16 : c->decode_mode == 3 ? 32 : c->decode_mode == 4 ? 8 : c->decode_mode == 5 ? 4
                    : (n_expected > 16384 ? 32 : 0);
    if (tight && f.rows_per_wg == 32 && !(S <= 32 && fused_decode_tight_rows32_supported(m->D, m->F, m->H, m->Ld))) f.rows_per_wg = 16;
    if (mp && f.rows_per_wg == 32) f.rows_per_wg = 16;  // (merged launches: the 16-row tilings only, decode_fused.hip)
    int rows = fused_decode_rows(m->D, m->F, m->H, m->Ld, (int)S, (int)B, f.rows_per_wg, kv24);
    if (clusters) {
      const size_t tiles = (B + 15) / 16, n_clusters = (tiles + 3) / 4;
      HIPCHK(c->cl_act.reserve(tiles * 16 * (size_t)m->D));
      HIPCHK(c->cl_part.reserve(tiles * (16 * 4 + 1) * 8));
      HIPCHK(c->cl_sync.reserve(n_clusters * 4));
      HIPCHK(hipMemsetAsync(c->cl_sync.p, 0, n_clusters * 4, st));
      f.cluster = 4;
      f.cl_act = c->cl_act.as<unsigned char>();
      f.cl_part = c->cl_part.as<int>();
      f.cl_sync = c->cl_sync.as<unsigned>();
      f.dev_error = dev_error_device_view(c);
    }
    for (int l = 0; l < m->Ld; ++l) {
      const DecLayerW &L = m->dec[(size_t)l];
      FusedLayerW &fl = f.L[l];
      fl.rnn_f = L.rnn_f.w; fl.rnn_w = L.rnn_w.w; fl.q = L.attn.q.w; fl.o = L.attn.o.w;
      fl.ffn1 = L.ffn1.w; fl.ffn2 = L.ffn2.w;
      fl.rnn_ln_s = L.rnn_ln.scale.as<float>(); fl.rnn_ln_b = L.rnn_ln.bias.as<float>();
      fl.attn_ln_s = L.attn.ln.scale.as<float>(); fl.attn_ln_b = L.attn.ln.bias.as<float>();
      fl.ffn_ln_s = L.ffn_ln.scale.as<float>(); fl.ffn_ln_b = L.ffn_ln.bias.as<float>();
    }
    f.out = out.w;
    f.out_n_dev = d_n_sl;
    f.shortlist = ds.shortlist;
    f.emb = embed_args(c);
    f.kv = c->kv.as<float>();
    f.kv24 = kv24;
    if (kv24 && c->kv_fmt_valid && c->kv_fmt_B == (int)B) f.kv_fmt = c->kv_fmt.as<unsigned char>();
    for (int l = 0; l < m->Ld; ++l) {  // the projections' constants, applied after the attention's sums (kernels.h, kv24)
      const AffineW &wk = m->dec[(size_t)l].attn.k, &wv = m->dec[(size_t)l].attn.v;
      f.kv_pb[l][0] = wk.w.pb;
      f.kv_pb[l][1] = wv.w.pb;
      f.kv_cs[l][0] = wk.w.colsum;
      f.kv_cs[l][1] = wv.w.colsum;
      f.kv_u[l][0] = wk.w.u;
      f.kv_u[l][1] = wv.w.u;
      f.kv_u256[l][0] = wk.w.u * (1.0f / 256.0f);  // exact scalings: the packed integers come back
      f.kv_u256[l][1] = wv.w.u * (1.0f / 256.0f);  // as accS * 256 (decode_fused.hip, unpack24f)
      f.kv_centre[l][0] = m->kv_centre_of(c->kv_gen, l, 0);  // (the tight form, the generation this batch's encoder wrote
      f.kv_centre[l][1] = m->kv_centre_of(c->kv_gen, l, 1);  //  it against; null until calibrated, not read then)
      f.kv_u4096[l][0] = wk.w.u * (1.0f / 4096.0f);  // ... as accS * 4096 from the narrow form (unpack20)
      f.kv_u4096[l][1] = wv.w.u * (1.0f / 4096.0f);
    }
    if (mp) {
      f.n_sub = mp->n;
      f.sub_dense = mp->dense ? 1 : 0;
      for (int j = 0; j < mp->n; ++j) f.sub[j] = mp->out[j];
      f.out_stride_wp = mp->stride_wp;
      f.out_stride_cs = mp->stride_cs;
      f.out_stride_pb = mp->stride_pb;
      f.shortlist = nullptr;
    }
    f.cells = c->state.as<float>();
    f.lengths = d_lengths;
    f.alpha = 1.0f / std::sqrt(static_cast<float>(m->D / m->H));
    f.eos = eos_id;
    f.out_ids = d_out_ids;
    f.out_len = d_out_len;
    f.align = d_align;
    f.align_out = d_align ? align_out : nullptr;
    f.trace = g_occ_trace;
    f.ticket = c->ticket.as<unsigned>();
    f.ticket_base = c->ticket_base;  // advanced below, once the launch is in the stream
    unsigned tickets = (unsigned)fused_decode_grid((int)B, true, rows);
    if (c->stamp_step >= 0 && c->stamps.p) {
      f.stamps = c->stamps.as<unsigned long long>();
      f.stamp_step = c->stamp_step;
    }
    const double macs = (double)B * f.max_steps *
                        (m->Ld * (4.0 * m->D * m->D + 2.0 * m->D * m->F) + (double)m->D * out.w.N);
    double wbytes = (double)f.max_steps * ((B + rows - 1) / rows) *
                    (m->Ld * (4.0 * m->D * m->D + 2.0 * m->D * m->F) + (double)m->D * out.w.n_tiles * 16);
    slimt_hip_model *gm = c->model;
    int wgs = ((int)B + rows - 1) / rows;
    if (gm->decoder_budget > 0) {
      // decoder admission (engine.h): launch k waits for launch k - n on its own stream
      constexpr size_t kRing = 64;
      clk.lap(7);
      std::lock_guard<std::mutex> lock(gm->gate_mu);
      clk.lap(2);
      while (gm->gate_ev.size() < kRing) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        gm->gate_ev.push_back(ev);
      }
      // K/V cache policy (decode_fused.hip, KV_AUX). The caches that are being read at any
      // moment are those of the decoders that run: at most one per context (a context is a
      // stream) and at most n by admission. While they fit the 256 MB Infinity Cache every
      // step re-reads them from there and non-temporal loads only lose that; beyond it the
      // cache streams from HBM anyway and non-temporal loads keep it from displacing the
      // weights. In between, the first layers' caches stay temporal (they then fit) and the
      // rest stream: t_layers = as many of the Ld per-layer caches as 300 MB cover.
      // Measured (tok/s, t_layers 0 / 1 / 2, Ld = 2): B=256, S=32 at 8 workers 17.1 / - /
      // 18.8 M, 12: 21.9 / 22.5 / 22.1, 16: 23.6 / 24.2 / 22.6, 20: 23.9 / 24.5 / 23.5;
      // B=512 full vocabulary 16.7 / 17.1 / 15.8; B=128, S=64 9.0 / 8.5 / 7.9; B=64, S=32
      // 11.9 / - / 13.6; base 6.4 / 5.9 / 5.7.
      // Which contexts have a decoder pending is judged from when they launched last (a translate
      // call takes a few milliseconds), not by asking the runtime: one hipEventQuery per context
      // under this lock made every launch O(contexts) runtime calls, and with a dozen worker threads
      // the launches queued behind each other (Service, 16 workers: 9.9 ms per launch call).
      // (the narrow form is what the model's sentences are expected to take where the kernels have it: 2.5 bytes per value)
      const double kv_bytes = (double)m->Ld * 2.0 * (double)B * (double)S * m->D * (f.kv_tight ? 2.0 : f.kv_fmt ? 2.5 : kv24 ? 3.0 : 4.0);
      const auto now = std::chrono::steady_clock::now();
      double pending = kv_bytes;
      size_t contexts = 1;
      bool known = false;
      for (auto &g : gm->gate_ctx) {
        if (g.ctx == c) {
          g.seq = gm->gate_seq;
          g.kv_bytes = kv_bytes;
          g.when = now;
          known = true;
        } else if (gm->gate_seq - g.seq < kRing && now - g.when < std::chrono::milliseconds(25)) {
          pending += g.kv_bytes;
          contexts += 1;
        }
      }
      if (!known) gm->gate_ctx.push_back({c, gm->gate_seq, kv_bytes, now});
      // Sentences per workgroup (decode mode 0): the fewest of 16 / 8 / 4 with which the decoders of the
      // contexts that have one pending (this launch's shape taken for all of them) still fit the budget --
      // one batch of 256 alone runs on 64 CUs instead of 16, twenty batches of 64 on 160 instead of 80, and
      // the headline's twenty batches of 256 stay at 16 (a workgroup of fewer sentences streams the same
      // weights for them: worth it only for CUs that would idle). Results do not depend on it.
      if (c->decode_mode == 0 && f.rows_per_wg == 0 && gm->adaptive_rows) {
        for (int spw : {4, 8}) {
          if (fused_decode_rows(m->D, m->F, m->H, m->Ld, (int)S, (int)B, spw, kv24) != spw) break;
          static const double oversub = std::getenv("SLIMT_ROWS_OVERSUB") ? std::atof(std::getenv("SLIMT_ROWS_OVERSUB")) : 1.0;
          if ((double)(contexts * (size_t)(((int)B + spw - 1) / spw)) <= oversub * (double)gm->decoder_budget) {
            f.rows_per_wg = spw;
            rows = spw;
            tickets = (unsigned)fused_decode_grid((int)B, true, rows);
            wgs = ((int)B + rows - 1) / rows;
            wbytes = (double)f.max_steps * wgs *
                     (m->Ld * (4.0 * m->D * m->D + 2.0 * m->D * m->F) + (double)m->D * out.w.n_tiles * 16);
            break;
          }
        }
      }
      size_t n = (size_t)std::max(1, gm->decoder_budget / wgs);
      if (n > kRing) n = kRing;
      const double active = pending / (double)contexts * (double)std::min(contexts, n);
      // ... in eighths of a layer's caches (kernels.h, kv_temporal_eighths): SLIMT_KV_BUDGET_MB /
      // SLIMT_KV_GRAIN tune the rule (defaults: 300 MB in whole layers, the measured optimum above)
      static const double budget = (std::getenv("SLIMT_KV_BUDGET_MB") ? std::atof(std::getenv("SLIMT_KV_BUDGET_MB")) : 300.0) * 1e6;
      static const int grain = std::getenv("SLIMT_KV_GRAIN") ? std::max(1, std::atoi(std::getenv("SLIMT_KV_GRAIN"))) : 8;
      const int all = 8 * m->Ld;
      int eighths = gm->kv_policy == 1 ? all : gm->kv_policy == 2 ? 0
                    : (int)std::min((double)all, std::floor((double)all * budget / active));
      if (gm->kv_policy == 0) eighths = eighths / grain * grain;
      // Round 4: WHICH caches stay temporal. "Layer 0 of every decoder" (the rule above) makes every workgroup pay one
      // cached and one streamed attention per step; keeping BOTH layers of k of every 8 launches (by admission order)
      // and streaming both layers of the others holds the same bytes but lets the kept decoders run their whole step
      // at the cached speed (and finish together, so no CU idles inside a launch): 32.35 -> 33.05-33.26 M tok/s at
      // k = 6, 32.5-32.9 at k = 5, 32.3-32.6 at k = 7, 31.7 all temporal (profiles/archive/r04_v3_kv_by_launch.txt).
      // k = as many eighths of the pending decoders as SLIMT_KV_LAUNCH_BUDGET_MB covers (round 4: 270, the 24-bit form;
      // round 5: 300 -- with the narrow form the headline's decoders hold 232 MB, k = 6 / 7 / 8 measure the same
      // (35.8 / 35.7 / 35.6 M tok/s, profiles/r05_kv_keep_sweep.txt), and every launch then runs the kept instantiation);
      // SLIMT_KV_BY_LAUNCH: 0 = the per-layer rule, 1..8 = that k.
      static const bool store_nt_rule = std::getenv("SLIMT_KV_STORE_NT") && std::getenv("SLIMT_KV_STORE_NT")[0] == '1';
      static const int by_launch = std::getenv("SLIMT_KV_BY_LAUNCH") ? std::atoi(std::getenv("SLIMT_KV_BY_LAUNCH")) : -1;
      static const double launch_budget =
          (std::getenv("SLIMT_KV_LAUNCH_BUDGET_MB") ? std::atof(std::getenv("SLIMT_KV_LAUNCH_BUDGET_MB")) : 300.0) * 1e6;
      // (sentences of up to 32 tokens; longer ones measured 2-3 % slower this way and keep the per-layer rule:
      // S = 64 18.5 -> 17.9 M, S = 128 8.2 -> 8.0 M)
      // ... and launches of which at least four fit the budget: ONE batch of 4096 holds 400 MB by itself, and keeping
      // "all of it" thrashes where "its layer 0" fits (29.8 -> 27.2 M tok/s)
      if (gm->kv_policy == 0 && by_launch != 0 && eighths < all && S <= 32 && 4.0 * kv_bytes <= launch_budget) {
        const int k = by_launch > 0 ? std::min(by_launch, 8) : (int)std::min(8.0, std::floor(8.0 * launch_budget / active));
        // (the slot was drawn when the call started, with the k of the admission before it: the encoder of a call
        // whose cache will be streamed can then write it non-temporally, SLIMT_KV_STORE_NT=1)
        eighths = (int)(call_slot % 8) < (store_nt_rule ? call_k : k) ? all : 0;
        gm->kv_k_last.store(k, std::memory_order_relaxed);
      } else {
        gm->kv_k_last.store(8, std::memory_order_relaxed);
      }
      f.kv_nt = eighths < all;
      f.kv_temporal_eighths = eighths;
      if (gm->gate_seq >= n) HIPCHK(hipStreamWaitEvent(st, gm->gate_ev[(gm->gate_seq - n) % kRing], 0));
      clk.lap(3);
      const int n_home = gm->xcd_affinity;
      const bool affine = n_home > 0 && rows == 16 && wgs <= 16 * n_home;
      if (affine) {
        if (gm->gate_home.size() < kRing) gm->gate_home.assign(kRing, 0u);
        unsigned mask = gm->gate_seq >= n ? gm->gate_home[(gm->gate_seq - n) % kRing] : 0u;
        if (!mask || __builtin_popcount(mask) != n_home) {
          const unsigned slot = (unsigned)(gm->gate_seq % (size_t)(8 / n_home));
          mask = n_home == 1 ? 1u << slot : n_home == 2 ? 3u << (2 * slot) : 0xfu << (4 * (slot & 1));
        }
        gm->gate_home[gm->gate_seq % kRing] = mask;
        f.home_mask = mask;
        f.xstate = reinterpret_cast<unsigned long long *>(c->ticket.as<unsigned>() + 2);
        f.xarr_base = c->xarr_base;
        f.xclaim_base = c->xclaim_base;
        f.xgrid = 8u * (unsigned)((wgs + n_home - 1) / n_home);
      } else if (!gm->gate_home.empty()) {
        gm->gate_home[gm->gate_seq % kRing] = 0u;
      }
      {
        ProfScope p(c, SLIMT_HIP_K_DECODE_FUSED, macs, wbytes);
        HIPCHK(launch_decode_fused(f, m->D, m->F, m->H, st));
      }
      clk.lap(4);
      if (affine) {
        c->xarr_base += f.xgrid;
        c->xclaim_base += (unsigned)wgs;
      } else {
        c->ticket_base += tickets;
      }
      HIPCHK(hipEventRecord(gm->gate_ev[gm->gate_seq % kRing], st));
      clk.lap(5);
      gm->gate_seq += 1;
      return 0;
    }
    ProfScope p(c, SLIMT_HIP_K_DECODE_FUSED, macs, wbytes);
    HIPCHK(launch_decode_fused(f, m->D, m->F, m->H, st));
    c->ticket_base += tickets;
    return 0;
  }
  const int n_parts = dgemm_col_blocks(out.w.K, out.w.N, (int)B);
  const EmbedArgs e = embed_args(c);
  const size_t max_steps = steps_hint > 0 ? (size_t)steps_hint : (Tmax > 1 ? Tmax : 1);
  int rc = 0;
  size_t t = 0;
  bool all_done = false;
  for (; t < max_steps && !rc; ++t) {
    hipError_t he = launch_decode_begin_step(e, ds, (int)B, t == 0, 1, c->part_val.as<float>(),
                                             c->part_idx.as<int>(), n_parts, c->dx.as<float>(), st);
    if (he != hipSuccess) { rc = fail((int)he, "decode_begin_step: %s", hipGetErrorString(he)); break; }
    if (steps_hint <= 0 && t > 0 && (t % 8) == 0) {
      // stop as soon as every sentence has emitted EOS (Model.cc:161)
      he = hipMemcpyAsync(c->n_finished_host, c->n_finished.p, sizeof(int), hipMemcpyDeviceToHost, st);
      if (he == hipSuccess) he = hipStreamSynchronize(st);
      if (he != hipSuccess) { rc = fail((int)he, "early-exit readback: %s", hipGetErrorString(he)); break; }
      if (*c->n_finished_host >= (int)B) { all_done = true; break; }
    }
    rc = decoder_layers(c, d_align, (int)Tmax, d_out_len, nullptr);
    if (rc) break;
    DGemmArgs g;
    g.B = (int)B;
    g.D = m->D;
    g.a = decoder_out_rows(c);
    g.w = out.w;
    g.part_val = c->part_val.as<float>();
    g.part_idx = c->part_idx.as<int>();
    g.n_parts = n_parts;
    {
      ProfScope p(c, SLIMT_HIP_K_LOGITS, gemm_macs((int)B, out.w), gemm_bytes(out.w));
      he = launch_dgemm(g, EPI_ARGMAX, st);
    }
    if (he != hipSuccess) { rc = fail((int)he, "logits gemm: %s", hipGetErrorString(he)); break; }
  }
  if (!rc && !all_done) {
    hipError_t he = launch_decode_begin_step(e, ds, (int)B, 0, 0, c->part_val.as<float>(),
                                             c->part_idx.as<int>(), n_parts, c->dx.as<float>(), st);
    if (he != hipSuccess) rc = fail((int)he, "final record: %s", hipGetErrorString(he));
  }
  return rc;
}

}  // namespace

namespace {
// Staging for the alignment rows of a pinned asynchronous translate: sized ONCE for the largest batch
// the context takes at this limit factor (a buffer that grows batch by batch frees and allocates device
// memory under the other contexts' streams, and hipFree waits for the device).
size_t align_staging_bytes(const slimt_hip_ctx *ctx, size_t B, size_t S, size_t Tmax, float limit_factor) {
  const size_t t_max = (size_t)(limit_factor * (float)ctx->max_S) + 1;
  const size_t worst = ctx->max_M * t_max * 4;
  return std::max(worst <= ((size_t)256 << 20) ? worst : 0, B * Tmax * S * 4);
}

// the device view of a pinned host allocation (hipHostMalloc / slimt_hip_host_alloc), else nullptr
void *host_device_view(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return a.type == hipMemoryTypeHost ? a.devicePointer : nullptr;
}

void shortlist_args(const slimt_hip_shortlist *sl, const uint32_t *d_ids, const uint32_t *d_len,
                    size_t B, size_t S, uint32_t *d_out, uint32_t *d_n, ShortlistArgs &a);
int shortlist_scratch(DevBuf &buf, const slimt_hip_shortlist *sl, hipStream_t st);

// Model::forward's order (Model.cc:117-120): the batch's shortlist first -- generated on ctx's
// stream into ctx->shortlist -- then translate_device with it. Arrays the device can read.
int translate_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const uint32_t *d_src_ids,
                        const uint32_t *d_lengths, size_t B, size_t S, float limit_factor, uint32_t eos_id,
                        uint32_t *d_out_ids, uint32_t *d_out_len, float *d_align, int steps_hint,
                        float *align_out) {
  const slimt_hip_model *m = ctx->model;
  hipStream_t st = ctx->stream;
  ctx->sl_host.clear();  // ctx->shortlist no longer holds what translate_host uploaded last
  HIPCHK(ctx->n_sl_dev.reserve(4));
  RCCHK(shortlist_scratch(ctx->sl_scratch, sl, st));
  ShortlistArgs a;
  shortlist_args(sl, d_src_ids, d_lengths, B, S, ctx->shortlist.as<uint32_t>(),
                 ctx->n_sl_dev.as<uint32_t>(), a);
  a.scratch = ctx->sl_scratch.as<uint32_t>();
  // the count also goes to a pinned word: what this context's PREVIOUS shortlist held is the host's
  // estimate of this one's size (a batch's shortlist is sized on the device; the host only picks
  // the decoder variant by it)
  uint32_t *hint = reinterpret_cast<uint32_t *>(ctx->n_finished_host) + 1;
  const size_t n_hint = *hint;
  void *hint_dev = nullptr;
  if (hipHostGetDevicePointer(&hint_dev, hint, 0) == hipSuccess) a.n_out_host = static_cast<uint32_t *>(hint_dev);
  const bool lean = fused_decoder_allowed(ctx) && fused_decode_supported(m->D, m->F, m->H, m->Ld) &&
                    (fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S) ||
                     long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S));
  // With the 64-row encoder the generator runs INSIDE the encoder launch, by the workgroup that starts first
  // (encode_tall.hip): as a launch of its own -- one workgroup -- it waited ~0.5 ms for a CU behind the
  // other batches' persistent kernels (47 us alone). Same ids, same count, same consumers.
  static const bool fold = !(std::getenv("SLIMT_SHORTLIST_FOLD") && std::getenv("SLIMT_SHORTLIST_FOLD")[0] == '0');
  // (the persistent encoders of sentences up to 64 tokens: 64-row / 32-row tiles at D = 256, D = 512; the
  // per-sentence kernel for 65..128 tokens keeps the two-kernel generator)
  const bool in_launch = fold && lean && fused_encoder_chosen(ctx, (int)B, (int)S) &&
                         shortlist_in_launch_lds_bytes(a.src_vocab, a.tgt_vocab) <= 64 * 1024;
  if (!in_launch) HIPCHK(launch_shortlist_generate(a, st));
  if (lean)  // the size stays on the device: capacity V, actual count read by the kernels
    return translate_device(ctx, d_src_ids, d_lengths, ctx->shortlist.as<uint32_t>(), B, S,
                            (size_t)m->V, limit_factor, eos_id, d_out_ids, d_out_len, d_align,
                            steps_hint, ctx->n_sl_dev.as<uint32_t>(), align_out, n_hint, in_launch ? &a : nullptr);
  uint32_t n = 0;  // stage kernels are sized on the host: one 4-byte read-back
  HIPCHK(hipMemcpyAsync(&n, ctx->n_sl_dev.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return translate_device(ctx, d_src_ids, d_lengths, ctx->shortlist.as<uint32_t>(), B, S, n,
                          limit_factor, eos_id, d_out_ids, d_out_len, d_align, steps_hint);
}
}  // namespace

extern "C" int slimt_hip_translate_device(slimt_hip_ctx *ctx, const uint32_t *d_src_ids,
                                          const uint32_t *d_lengths, size_t B, size_t S,
                                          const uint32_t *d_shortlist, size_t n_shortlist,
                                          float limit_factor, uint32_t eos_id,
                                          uint32_t *d_out_ids, uint32_t *d_out_len,
                                          float *d_align, int steps_hint) {
  if (!ctx || !d_src_ids || !d_lengths || !d_out_ids || !d_out_len) return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  if (n_shortlist > (size_t)ctx->model->V) return fail(-1, "shortlist larger than the vocabulary");
  if (n_shortlist && !d_shortlist) return fail(-1, "shortlist is NULL");
  HIPCHK(hipSetDevice(ctx->model->device));
  return translate_device(ctx, d_src_ids, d_lengths, d_shortlist, B, S, n_shortlist, limit_factor,
                          eos_id, d_out_ids, d_out_len, d_align, steps_hint);
}

namespace {
// Model::forward on host buffers: validate, queue H2D + kernels + D2H on the ctx stream.
int translate_host(slimt_hip_ctx *ctx, const uint32_t *src_ids, const uint32_t *lengths, size_t B,
                   size_t S, const uint32_t *shortlist, size_t n_shortlist, float limit_factor,
                   uint32_t eos_id, uint32_t *out_ids, uint32_t *out_len, float *align, bool wait) {
  if (!ctx || !src_ids || !lengths || !out_ids || !out_len) return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  StageClock hclk;
  const slimt_hip_model *m = ctx->model;
  if (n_shortlist > (size_t)m->V) return fail(-1, "shortlist larger than the vocabulary");
  if (n_shortlist && !shortlist) return fail(-1, "shortlist is NULL");
  for (size_t i = 0; i < B * S; ++i)
    if (src_ids[i] >= (uint32_t)m->V) return fail(-1, "token id %u out of range", src_ids[i]);
  for (size_t i = 0; i < n_shortlist; ++i)
    if (shortlist[i] >= (uint32_t)m->V) return fail(-1, "shortlist id %u out of range", shortlist[i]);
  for (size_t i = 0; i < B; ++i)
    if (lengths[i] > S) return fail(-1, "length %u > S", lengths[i]);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  const size_t Tmax = std::max<size_t>(1, (size_t)(limit_factor * (float)S));
  // the shortlist is read at random every step: it stays on the device, uploaded when it changes
  if (n_shortlist && (ctx->sl_host.size() != n_shortlist ||
                      std::memcmp(ctx->sl_host.data(), shortlist, n_shortlist * 4) != 0)) {
    ctx->sl_host.clear();  // until the upload has succeeded the device copy is nobody's
    HIPCHK(hipMemcpyAsync(ctx->shortlist.p, shortlist, n_shortlist * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // (the caller's array may be pageable: the copy is staged; done once per shortlist)
    ctx->sl_host.assign(shortlist, shortlist + n_shortlist);
  }
  // Asynchronous callers with PINNED buffers (hipHostMalloc / slimt_hip_host_alloc): the persistent
  // kernels read the ids / lengths and write tokens, lengths and alignments in host memory themselves
  // -- no copy on either side. Copies are what a host pipeline stalls on: an asynchronous copy
  // of stream A sits in a DMA queue behind copies that wait for stream B's kernels (measured with the
  // Service: 12 contexts of equal batches ran strictly one after the other, 3.7 M tok/s).
  const bool persistent = fused_decoder_allowed(ctx) && fused_decode_supported(m->D, m->F, m->H, m->Ld) &&
                          (fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S) ||
                           long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S));
  hclk.lap(0);
  if (!wait && persistent) {
    void *v_ids = host_device_view(src_ids), *v_len = host_device_view(lengths), *v_out = host_device_view(out_ids),
         *v_ol = host_device_view(out_len), *v_al = align ? host_device_view(align) : nullptr;
    hclk.lap(6);
    if (v_ids && v_len && v_out && v_ol && (!align || v_al)) {
      // alignment rows (Model.cc:84-108) are staged in device memory and leave for the host once per
      // sentence, as whole 16-byte stores when its loop ends: written row by row across PCIe from
      // inside the step loop they cost the Service 28 % (12.9 against 17.9 M tok/s)
      if (align) HIPCHK(ctx->align.reserve(align_staging_bytes(ctx, B, S, Tmax, limit_factor)));
      RCCHK(translate_device(ctx, static_cast<const uint32_t *>(v_ids), static_cast<const uint32_t *>(v_len),
                             ctx->shortlist.as<uint32_t>(), B, S, n_shortlist, limit_factor, eos_id,
                             static_cast<uint32_t *>(v_out), static_cast<uint32_t *>(v_ol),
                             align ? ctx->align.as<float>() : nullptr, (int)Tmax, nullptr,
                             static_cast<float *>(v_al)));
      return 0;
    }
  }
  HIPCHK(ctx->out_ids.reserve(B * Tmax * 4));
  if (align) HIPCHK(ctx->align.reserve(B * Tmax * S * 4));
  HIPCHK(hipMemcpyAsync(ctx->ids.p, src_ids, B * S * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->lengths.p, lengths, B * 4, hipMemcpyHostToDevice, st));
  // asynchronous callers never read back inside the loop: a fixed step budget (the persistent
  // decoder still leaves its loop as soon as every sentence has emitted EOS)
  RCCHK(translate_device(ctx, ctx->ids.as<uint32_t>(), ctx->lengths.as<uint32_t>(),
                         ctx->shortlist.as<uint32_t>(), B, S, n_shortlist, limit_factor, eos_id,
                         ctx->out_ids.as<uint32_t>(), ctx->out_len.as<uint32_t>(),
                         align ? ctx->align.as<float>() : nullptr, wait ? 0 : (int)Tmax));
  HIPCHK(hipMemcpyAsync(out_ids, ctx->out_ids.p, B * Tmax * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out_len, ctx->out_len.p, B * 4, hipMemcpyDeviceToHost, st));
  if (align) HIPCHK(hipMemcpyAsync(align, ctx->align.p, B * Tmax * S * 4, hipMemcpyDeviceToHost, st));
  if (wait) RCCHK(slimt_hip_ctx_synchronize(ctx));  // sleeps on a blocking-sync event (a worker per context: no spinning)
  return 0;
}
}  // namespace

extern "C" int slimt_hip_translate(slimt_hip_ctx *ctx, const uint32_t *src_ids,
                                   const uint32_t *lengths, size_t B, size_t S,
                                   const uint32_t *shortlist, size_t n_shortlist,
                                   float limit_factor, uint32_t eos_id, uint32_t *out_ids,
                                   uint32_t *out_len, float *align) {
  return translate_host(ctx, src_ids, lengths, B, S, shortlist, n_shortlist, limit_factor, eos_id,
                        out_ids, out_len, align, true);
}

extern "C" int slimt_hip_translate_async(slimt_hip_ctx *ctx, const uint32_t *src_ids,
                                         const uint32_t *lengths, size_t B, size_t S,
                                         const uint32_t *shortlist, size_t n_shortlist,
                                         float limit_factor, uint32_t eos_id, uint32_t *out_ids,
                                         uint32_t *out_len, float *align) {
  return translate_host(ctx, src_ids, lengths, B, S, shortlist, n_shortlist, limit_factor, eos_id,
                        out_ids, out_len, align, false);
}

// ---- several batches in one launch pair (include/slimt_hip.h, slimt_hip_translate_many*) ----------------------------
namespace {
constexpr size_t kMergeAlign = 32;  // a sub-batch starts at a multiple of the widest decoder tile (kernels.h, MergeOut)

bool merge_supported(const slimt_hip_ctx *c, size_t rows, size_t S) {
  const slimt_hip_model *m = c->model;
  return fused_decoder_allowed(c) && c->decode_mode != 6 && fused_decode_supported(m->D, m->F, m->H, m->Ld) &&
         fused_encoder_chosen(c, (int)rows, (int)S);
}

// the sub-batches' tables. align_staging (nullable): the alignment rows are staged there (global sentence order) and
// each batch's `align` (a device view of the caller's pinned array) is where they go when a sentence ends
// align: sub-batches start at multiples of it -- the decoder's tile where every sub-batch has its own output layer; with
// ONE layer for all of them (one shortlist pointer and size, or the full vocabulary) they follow each other densely
// whatever it says
int build_merge_plan(const slimt_hip_ctx *c, const slimt_hip_batch *b, size_t n, size_t S, size_t Tmax, float limit_factor,
                     int steps_hint, float *align_staging, MergePlan &mp, size_t &rows, size_t align = kMergeAlign) {
  mp.n = (int)n;
  rows = 0;
  mp.dense = true;
  for (size_t j = 1; j < n; ++j)
    if (b[j].shortlist != b[0].shortlist || b[j].n_shortlist != b[0].n_shortlist) mp.dense = false;
  static const bool no_dense = std::getenv("SLIMT_MERGE_DENSE") && std::getenv("SLIMT_MERGE_DENSE")[0] == '0';  // A/B
  if (no_dense) mp.dense = false;
  if (mp.dense) align = 1;
  size_t stage_at = 0;  // floats of alignment staging handed out so far (16-byte pieces: the decoder copies whole quads)
  for (size_t j = 0; j < n; ++j) {
    const size_t Sj = b[j].S ? b[j].S : S;
    if (Sj > S) return fail(-1, "batch %zu is padded to %zu tokens, the launch to %zu", j, Sj, S);
    const size_t Tj = std::max<size_t>(1, (size_t)(limit_factor * (float)Sj));  // Model.cc:159-161 on this batch's length
    if (!b[j].src_ids || !b[j].lengths || !b[j].out_ids || !b[j].out_len) return fail(-1, "batch %zu: null array", j);
    if (b[j].B == 0) return fail(-1, "batch %zu is empty", j);
    if (b[j].n_shortlist > (size_t)c->model->V) return fail(-1, "batch %zu: shortlist larger than the vocabulary", j);
    if (b[j].n_shortlist && !b[j].shortlist) return fail(-1, "batch %zu: shortlist is NULL", j);
    if ((b[j].n_shortlist != 0) != (b[0].n_shortlist != 0))
      return fail(-1, "merged batches: all with a shortlist or all with the full vocabulary");
    MergeIn &in = mp.in[j];
    MergeOut &o = mp.out[j];
    in.ids = b[j].src_ids;
    in.lengths = b[j].lengths;
    in.first = (int)rows;
    in.n = (int)b[j].B;
    in.S = (int)Sj;
    o.lengths = b[j].lengths;
    o.out_ids = b[j].out_ids;
    o.out_len = b[j].out_len;
    if (b[j].align && align_staging) {
      o.align = align_staging + stage_at;
      o.align_out = b[j].align;
      stage_at += (b[j].B * Tj * Sj + 3) / 4 * 4;
    } else {
      o.align = b[j].align;
      o.align_out = nullptr;
    }
    o.shortlist = b[j].n_shortlist ? b[j].shortlist : nullptr;
    o.first = (int)rows;
    o.n = (int)b[j].B;
    o.N = b[j].n_shortlist ? (int)b[j].n_shortlist : c->model->V;
    o.job = 0;
    o.S = (int)Sj;
    o.Tmax = (int)Tj;
    o.max_steps = steps_hint > 0 ? std::min(steps_hint, (int)Tj) : (int)Tj;
    if (b[j].n_shortlist) {  // one packing job per distinct (pointer, size)
      int job = -1;
      for (int q = 0; q < mp.n_jobs; ++q)
        if (mp.jobs[q].idx == b[j].shortlist && mp.jobs[q].N == (int)b[j].n_shortlist) job = q;
      if (job < 0) {
        job = mp.n_jobs++;
        mp.jobs[job].idx = b[j].shortlist;
        mp.jobs[job].N = (int)b[j].n_shortlist;
        mp.max_N = std::max(mp.max_N, (int)b[j].n_shortlist);
      }
      o.job = job;
    }
    rows += (b[j].B + align - 1) / align * align;
  }
  return 0;
}

// the decoder tile of a merged launch: always 16 sentences -- where an unmerged call would take the 32-sentence tiling (decode
// mode 3, or mode 0 with an output layer of more than 16k columns) a merged one runs the 16-sentence tiling instead
// (translate_device; decode_fused.hip compiles the merged paths into the 16-row tilings only)
size_t merge_tile(const slimt_hip_ctx *c, size_t n_columns) {
  (void)c;
  (void)n_columns;
  return 16;  // (merged launches never take the 32-sentence tiling: translate_device)
}
}  // namespace

extern "C" size_t slimt_hip_translate_many_rows(const size_t *B, size_t n_batches) {
  size_t rows = 0;
  for (size_t j = 0; B && j < n_batches; ++j) rows += (B[j] + kMergeAlign - 1) / kMergeAlign * kMergeAlign;
  return rows;
}

extern "C" int slimt_hip_translate_many_device(slimt_hip_ctx *ctx, const slimt_hip_batch *batches, size_t n_batches, size_t S,
                                               float limit_factor, uint32_t eos_id, int steps_hint) {
  if (!ctx || !batches || n_batches == 0) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(ctx->model->device));
  const size_t Tmax = std::max<size_t>(1, (size_t)(limit_factor * (float)S));
  MergePlan mp;
  size_t rows = 0;
  const bool mergeable = n_batches > 1 && n_batches <= (size_t)kMaxMerge;
  size_t n_cols = 0;  // the widest output layer of the launch
  for (size_t j = 0; j < n_batches; ++j) n_cols = std::max(n_cols, batches[j].n_shortlist ? batches[j].n_shortlist : (size_t)ctx->model->V);
  if (mergeable)
    RCCHK(build_merge_plan(ctx, batches, n_batches, S, Tmax, limit_factor, steps_hint, nullptr, mp, rows, merge_tile(ctx, n_cols)));
  if (mergeable && rows <= ctx->max_B && rows * S <= ctx->max_M && S <= ctx->max_S && merge_supported(ctx, rows, S))
    return translate_device(ctx, batches[0].src_ids, batches[0].lengths, batches[0].shortlist, rows, S, (size_t)mp.max_N,
                            limit_factor, eos_id, batches[0].out_ids, batches[0].out_len, batches[0].align, steps_hint, nullptr,
                            nullptr, 0, nullptr, &mp);
  for (size_t j = 0; j < n_batches; ++j) {  // batch by batch, in order, on the same stream
    const slimt_hip_batch &b = batches[j];
    RCCHK(slimt_hip_translate_device(ctx, b.src_ids, b.lengths, b.B, b.S ? b.S : S, b.shortlist, b.n_shortlist, limit_factor,
                                     eos_id, b.out_ids, b.out_len, b.align, steps_hint));
  }
  return 0;
}

extern "C" int slimt_hip_translate_many_async(slimt_hip_ctx *ctx, const slimt_hip_batch *batches, size_t n_batches, size_t S,
                                              float limit_factor, uint32_t eos_id) {
  if (!ctx || !batches || n_batches == 0) return fail(-1, "null argument");
  const slimt_hip_model *m = ctx->model;
  HIPCHK(hipSetDevice(m->device));
  const size_t Tmax = std::max<size_t>(1, (size_t)(limit_factor * (float)S));
  bool merged = n_batches > 1 && n_batches <= (size_t)kMaxMerge && S <= ctx->max_S;
  slimt_hip_batch dev[kMaxMerge];
  bool any_align = false;
  size_t rows = 0;
  for (size_t j = 0; merged && j < n_batches; ++j) {
    const slimt_hip_batch &b = batches[j];
    if (!b.src_ids || !b.lengths || !b.out_ids || !b.out_len || b.B == 0) return fail(-1, "batch %zu: null array or empty", j);
    if (b.shortlist != batches[0].shortlist || b.n_shortlist != batches[0].n_shortlist) merged = false;  // one host shortlist
    const size_t Sj = b.S ? b.S : S;
    if (Sj > S) return fail(-1, "batch %zu is padded to %zu tokens, the launch to %zu", j, Sj, S);
    for (size_t i = 0; merged && i < b.B * Sj; ++i)
      if (b.src_ids[i] >= (uint32_t)m->V) return fail(-1, "batch %zu: token id %u out of range", j, b.src_ids[i]);
    for (size_t i = 0; merged && i < b.B; ++i)
      if (b.lengths[i] > Sj) return fail(-1, "batch %zu: length %u > S", j, b.lengths[i]);
    dev[j] = b;
    dev[j].src_ids = static_cast<const uint32_t *>(host_device_view(b.src_ids));
    dev[j].lengths = static_cast<const uint32_t *>(host_device_view(b.lengths));
    dev[j].out_ids = static_cast<uint32_t *>(host_device_view(b.out_ids));
    dev[j].out_len = static_cast<uint32_t *>(host_device_view(b.out_len));
    dev[j].align = b.align ? static_cast<float *>(host_device_view(b.align)) : nullptr;
    if (!dev[j].src_ids || !dev[j].lengths || !dev[j].out_ids || !dev[j].out_len || (b.align && !dev[j].align)) merged = false;
    any_align = any_align || b.align != nullptr;
    rows += b.B;  // (one shortlist for all: the sub-batches follow each other densely)
  }
  merged = merged && rows <= ctx->max_B && rows * S <= ctx->max_M && merge_supported(ctx, rows, S);
  if (!merged) {
    for (size_t j = 0; j < n_batches; ++j) {
      const slimt_hip_batch &b = batches[j];
      RCCHK(translate_host(ctx, b.src_ids, b.lengths, b.B, b.S ? b.S : S, b.shortlist, b.n_shortlist, limit_factor, eos_id,
                           b.out_ids, b.out_len, b.align, false));
    }
    return 0;
  }
  const size_t n_sl = batches[0].n_shortlist;
  const uint32_t *shortlist = batches[0].shortlist;
  if (n_sl > (size_t)m->V) return fail(-1, "shortlist larger than the vocabulary");
  if (n_sl && !shortlist) return fail(-1, "shortlist is NULL");
  for (size_t i = 0; i < n_sl; ++i)
    if (shortlist[i] >= (uint32_t)m->V) return fail(-1, "shortlist id %u out of range", shortlist[i]);
  hipStream_t st = ctx->stream;
  if (n_sl && (ctx->sl_host.size() != n_sl || std::memcmp(ctx->sl_host.data(), shortlist, n_sl * 4) != 0)) {
    ctx->sl_host.clear();  // (translate_host: uploaded when it changes)
    HIPCHK(hipMemcpyAsync(ctx->shortlist.p, shortlist, n_sl * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->sl_host.assign(shortlist, shortlist + n_sl);
  }
  if (any_align) HIPCHK(ctx->align.reserve(align_staging_bytes(ctx, rows, S, Tmax, limit_factor)));
  for (size_t j = 0; j < n_batches; ++j) {
    dev[j].shortlist = n_sl ? ctx->shortlist.as<uint32_t>() : nullptr;
    dev[j].n_shortlist = n_sl;
  }
  MergePlan mp;
  RCCHK(build_merge_plan(ctx, dev, n_batches, S, Tmax, limit_factor, 0, any_align ? ctx->align.as<float>() : nullptr, mp, rows));
  return translate_device(ctx, dev[0].src_ids, dev[0].lengths, dev[0].shortlist, rows, S, (size_t)mp.max_N, limit_factor, eos_id,
                          dev[0].out_ids, dev[0].out_len, any_align ? ctx->align.as<float>() : nullptr, (int)Tmax, nullptr,
                          any_align ? dev[0].align : nullptr, 0, nullptr, &mp);
}

extern "C" int slimt_hip_host_alloc(size_t bytes, void **out) {
  if (!out) return fail(-1, "null argument");
  *out = nullptr;
  g_hip_called.store(true, std::memory_order_relaxed);
  HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return 0;
}

extern "C" int slimt_hip_host_free(void *p) {
  if (p) HIPCHK(hipHostFree(p));
  return 0;
}

extern "C" int slimt_hip_encode(slimt_hip_ctx *ctx, const uint32_t *src_ids,
                                const uint32_t *lengths, size_t B, size_t S, float *embed_out,
                                float *layer_out, float *enc_out) {
  if (!ctx || !src_ids || !lengths) return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  const slimt_hip_model *m = ctx->model;
  for (size_t i = 0; i < B * S; ++i)
    if (src_ids[i] >= (uint32_t)m->V) return fail(-1, "token id %u out of range", src_ids[i]);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(ctx->ids.p, src_ids, B * S * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->lengths.p, lengths, B * 4, hipMemcpyHostToDevice, st));
  RCCHK(encode_device(ctx, (int)B, (int)S, embed_out, layer_out));
  if (enc_out)
    HIPCHK(hipMemcpyAsync(enc_out, ctx->x0.p, B * S * (size_t)m->D * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int slimt_hip_encode_embedded(slimt_hip_ctx *ctx, const float *embedding,
                                         const uint32_t *lengths, size_t B, size_t S, float *enc_out) {
  if (!ctx || !embedding || !lengths || !enc_out) return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  const slimt_hip_model *m = ctx->model;
  for (size_t i = 0; i < B; ++i)
    if (lengths[i] > S) return fail(-1, "length %u > S", lengths[i]);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  const size_t nbytes = B * S * (size_t)m->D * 4;
  HIPCHK(hipMemcpyAsync(ctx->x0.p, embedding, nbytes, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->lengths.p, lengths, B * 4, hipMemcpyHostToDevice, st));
  RCCHK(encode_device(ctx, (int)B, (int)S, nullptr, nullptr, nullptr, nullptr, nullptr, true));
  HIPCHK(hipMemcpyAsync(enc_out, ctx->x0.p, nbytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int slimt_hip_decode_begin_from(slimt_hip_ctx *ctx, const float *encoder_out,
                                           const uint32_t *lengths, size_t B, size_t S,
                                           const uint32_t *shortlist, size_t n_shortlist) {
  if (!ctx || !encoder_out || !lengths) return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  const slimt_hip_model *m = ctx->model;
  for (size_t i = 0; i < B; ++i)
    if (lengths[i] > S) return fail(-1, "length %u > S", lengths[i]);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(ctx->x0.p, encoder_out, B * S * (size_t)m->D * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->lengths.p, lengths, B * 4, hipMemcpyHostToDevice, st));
  ctx->B = (int)B;
  ctx->S = (int)S;
  ctx->have_encoder_out = true;
  ctx->kv_ready = false;  // the cross-attention K/V of THIS encoder output are computed below
  ctx->decode_ready = false;
  return slimt_hip_decode_begin(ctx, shortlist, n_shortlist);
}

extern "C" int slimt_hip_decode_begin(slimt_hip_ctx *ctx, const uint32_t *shortlist,
                                      size_t n_shortlist) {
  if (!ctx) return fail(-1, "ctx is NULL");
  const slimt_hip_model *m = ctx->model;
  if (n_shortlist > (size_t)m->V) return fail(-1, "shortlist larger than the vocabulary");
  if (n_shortlist && !shortlist) return fail(-1, "shortlist is NULL");
  for (size_t i = 0; i < n_shortlist; ++i)
    if (shortlist[i] >= (uint32_t)m->V) return fail(-1, "shortlist id %u out of range", shortlist[i]);
  HIPCHK(hipSetDevice(m->device));
  if (n_shortlist) {
    ctx->sl_host.clear();  // ctx->shortlist no longer holds what translate_host uploaded last
    HIPCHK(hipMemcpyAsync(ctx->shortlist.p, shortlist, n_shortlist * 4, hipMemcpyHostToDevice,
                          ctx->stream));
  }
  RCCHK(decode_setup(ctx, n_shortlist));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

static int decode_step_impl(slimt_hip_ctx *ctx, const uint32_t *prev, const float *states_in,
                            float *logits, float *attn, float *states);

extern "C" int slimt_hip_decode_step(slimt_hip_ctx *ctx, const uint32_t *prev, float *logits,
                                     float *attn, float *states) {
  return decode_step_impl(ctx, prev, nullptr, logits, attn, states);
}

extern "C" int slimt_hip_decode_step_states(slimt_hip_ctx *ctx, const uint32_t *prev,
                                            const float *states_in, float *logits, float *attn,
                                            float *states_out) {
  if (!states_in) return fail(-1, "null argument");
  return decode_step_impl(ctx, prev, states_in, logits, attn, states_out);
}

static int decode_step_impl(slimt_hip_ctx *ctx, const uint32_t *prev, const float *states_in,
                            float *logits, float *attn, float *states) {
  if (!ctx || !logits) return fail(-1, "null argument");
  if (!ctx->decode_ready) return fail(-1, "decode_step before decode_begin");
  const slimt_hip_model *m = ctx->model;
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  const size_t B = (size_t)ctx->B, S = (size_t)ctx->S, D = (size_t)m->D;
  if (prev) {
    for (size_t i = 0; i < B; ++i)
      if (prev[i] >= (uint32_t)m->V) return fail(-1, "token id %u out of range", prev[i]);
    HIPCHK(hipMemcpyAsync(ctx->prev.p, prev, B * 4, hipMemcpyHostToDevice, st));
  }
  if (states_in)  // the caller's SSRU cells (Decoder::step's `states`, Transformer.cc:120-128)
    HIPCHK(hipMemcpyAsync(ctx->state.p, states_in, (size_t)m->Ld * B * D * 4, hipMemcpyHostToDevice, st));
  HIPCHK(launch_embed_decoder(embed_args(ctx), ctx->prev.as<uint32_t>(), (int)B, prev == nullptr,
                              ctx->dx.as<float>(), st));
  if (attn) HIPCHK(ctx->attn_dbg.reserve(B * (size_t)m->H * S * 4));
  RCCHK(decoder_layers(ctx, nullptr, 0, nullptr, attn ? ctx->attn_dbg.as<float>() : nullptr));
  const AffineW &out = output_layer(ctx);
  const size_t N = (size_t)out.w.N;
  HIPCHK(ctx->logits.reserve(B * N * 4));
  DGemmArgs g;
  g.B = (int)B;
  g.D = m->D;
  g.a = decoder_out_rows(ctx);
  g.w = out.w;
  g.y = ctx->logits.as<float>();
  g.ldy = (int)N;
  HIPCHK(launch_dgemm(g, EPI_PLAIN, st));
  HIPCHK(hipMemcpyAsync(logits, ctx->logits.p, B * N * 4, hipMemcpyDeviceToHost, st));
  if (attn)
    HIPCHK(hipMemcpyAsync(attn, ctx->attn_dbg.p, B * (size_t)m->H * S * 4, hipMemcpyDeviceToHost, st));
  if (states)
    HIPCHK(hipMemcpyAsync(states, ctx->state.p, (size_t)m->Ld * B * D * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// ---------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------
extern "C" int slimt_hip_profile_enable(slimt_hip_ctx *ctx, int kernel_id) {
  if (!ctx) return fail(-1, "ctx is NULL");
  if (kernel_id < 0 || kernel_id >= SLIMT_HIP_K_COUNT) return fail(-1, "bad kernel id %d", kernel_id);
  ctx->prof_kernel = kernel_id;
  return slimt_hip_profile_reset(ctx);
}

extern "C" int slimt_hip_profile_reset(slimt_hip_ctx *ctx) {
  if (!ctx) return fail(-1, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->prof_used = 0;
  ctx->prof_macs = 0;
  ctx->prof_bytes = 0;
  return 0;
}

extern "C" int slimt_hip_profile_read(slimt_hip_ctx *ctx, uint64_t *launches, double *total_ms,
                                      double *int8_macs, double *weight_bytes) {
  if (!ctx) return fail(-1, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double ms = 0;
  for (size_t i = 0; i < ctx->prof_used; ++i) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, ctx->prof_events[i].first, ctx->prof_events[i].second));
    ms += t;
  }
  if (launches) *launches = ctx->prof_used;
  if (total_ms) *total_ms = ms;
  if (int8_macs) *int8_macs = ctx->prof_macs;
  if (weight_bytes) *weight_bytes = ctx->prof_bytes;
  return 0;
}

// Diagnostic occupancy trace of the persistent kernels (kernels.h, OccTrace)
extern "C" int slimt_hip_debug_occupancy_trace(void *device_buf, size_t capacity) {
  g_occ_trace.buf = static_cast<unsigned long long *>(device_buf);
  g_occ_trace.capacity = device_buf ? (unsigned)capacity : 0u;
  return 0;
}

// Diagnostic: wall-clock stamps (100 MHz ticks) at the phase boundaries of the
// persistent decoder, workgroup 0, decode step `step`. step < 0 disables.
extern "C" int slimt_hip_debug_decode_stamps(slimt_hip_ctx *ctx, int step, uint64_t *out,
                                             size_t n) {
  if (!ctx) return fail(-1, "ctx is NULL");
  HIPCHK(hipSetDevice(ctx->model->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (out && n && ctx->stamps.p) {
    if (n > 64) n = 64;
    HIPCHK(hipMemcpy(out, ctx->stamps.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  ctx->stamp_step = step;
  if (step >= 0) {
    HIPCHK(ctx->stamps.reserve(64 * sizeof(uint64_t)));
    HIPCHK(hipMemset(ctx->stamps.p, 0, 64 * sizeof(uint64_t)));
  }
  return 0;
}

// ---------------------------------------------------------------------------
// lexical shortlist (slimt/Shortlist.{hh,cc})
// ---------------------------------------------------------------------------
namespace {

constexpr uint64_t kShortlistMagic = 0xF11A48D5013417F5ull;  // Shortlist.hh:40

// hash_combine over uint64 words from header.frequent on (Shortlist.cc:66-73,
// Utils.hh:47-67 with libstdc++'s identity std::hash<uint64_t>)
uint64_t shortlist_checksum(const unsigned char *blob, size_t size) {
  uint64_t seed = 0;
  for (size_t i = 16; i + 8 <= size; i += 8) {
    uint64_t v;
    std::memcpy(&v, blob + i, 8);
    seed ^= v + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
  }
  return seed;
}

// bitmaps start zeroed and every generate call leaves them zeroed
int shortlist_scratch(DevBuf &buf, const slimt_hip_shortlist *sl, hipStream_t st) {
  const size_t need = shortlist_scratch_bytes((int)sl->source_vocab, (int)sl->target_vocab);
  if (buf.bytes < need) {
    HIPCHK(buf.reserve(need));
    HIPCHK(hipMemsetAsync(buf.p, 0, buf.bytes, st));
  }
  return 0;
}

void shortlist_args(const slimt_hip_shortlist *sl, const uint32_t *d_ids, const uint32_t *d_len,
                    size_t B, size_t S, uint32_t *d_out, uint32_t *d_n, ShortlistArgs &a) {
  a.w2o = sl->w2o.as<unsigned long long>();
  a.lists = sl->lists.as<uint32_t>();
  a.frequent = sl->frequent;
  a.shared = sl->shared ? 1 : 0;
  a.src_vocab = (int)sl->source_vocab;
  a.tgt_vocab = (int)sl->target_vocab;
  a.ids = d_ids;
  a.lengths = d_len;
  a.B = (int)B;
  a.S = (int)S;
  a.out = d_out;
  a.n_out = d_n;
}

}  // namespace

extern "C" int slimt_hip_shortlist_create(const void *blob, size_t blob_size, size_t source_vocab,
                                          size_t target_vocab, int shared, int check, int device,
                                          slimt_hip_shortlist **out) {
  if (!blob || !out) return fail(-1, "null argument");
  if (source_vocab == 0 || target_vocab == 0 || source_vocab > (1u << 22) || target_vocab > (1u << 22))
    return fail(-1, "bad vocabulary sizes %zu / %zu", source_vocab, target_vocab);
  if (shortlist_lds_bytes((int)source_vocab, (int)target_vocab) > 160 * 1024)
    return fail(-1, "vocabularies too large for the on-chip truth tables");
  uint64_t h[6];
  if (blob_size < sizeof(h))  // Shortlist.cc:49-51
    return fail(-1, "shortlist length too short to have a header: %zu", blob_size);
  const unsigned char *p = static_cast<const unsigned char *>(blob);
  std::memcpy(h, p, sizeof(h));
  if (h[0] != kShortlistMagic) return fail(-1, "incorrect magic in binary shortlist");  // :56
  const uint64_t n_off = h[4], n_ids = h[5];
  if (n_off > (1ull << 32) || n_ids > (1ull << 40)) return fail(-1, "implausible shortlist header");
  const uint64_t expected = sizeof(h) + n_off * 8 + n_ids * 4;  // :58-64
  if (expected != blob_size)
    return fail(-1, "shortlist header claims file size should be %llu but file is %zu",
                (unsigned long long)expected, blob_size);
  if (check && shortlist_checksum(p, blob_size) != h[1])  // :66-76
    return fail(-1, "checksum check failed: this binary shortlist is corrupted");
  if (n_off != source_vocab + 1)
    return fail(-1, "shortlist has %llu offsets, expected source vocabulary + 1 = %zu",
                (unsigned long long)n_off, source_vocab + 1);
  // content_check (Shortlist.cc:16-38), always: the kernel trusts these ranges
  const unsigned char *po = p + sizeof(h), *pl = po + n_off * 8;
  uint64_t prev = 0;
  for (uint64_t i = 0; i < n_off; ++i) {
    uint64_t v;
    std::memcpy(&v, po + 8 * i, 8);
    // the reference's check wants every offset but the last strictly inside (Shortlist.cc:18-21)
    if (v > n_ids || v < prev || (check && i + 1 < n_off && v >= n_ids))
      return fail(-1, "offset table not within shortlist size");
    prev = v;
  }
  if (prev != n_ids) return fail(-1, "word_to_offset != shortlist_size");
  for (uint64_t j = 0; j < n_ids; ++j) {
    uint32_t v;
    std::memcpy(&v, pl + 4 * j, 4);
    if (v >= target_vocab) return fail(-1, "shortlist indices are out of bounds");
  }
  HIPCHK(hipSetDevice(device));
  auto *sl = new slimt_hip_shortlist();
  sl->device = device;
  sl->frequent = h[2];
  sl->best = h[3];
  sl->source_vocab = source_vocab;
  sl->target_vocab = target_vocab;
  sl->shared = shared != 0;
  hipError_t e = sl->w2o.reserve(n_off * 8);
  if (e == hipSuccess) e = sl->lists.reserve(n_ids ? n_ids * 4 : 4);
  if (e == hipSuccess) e = sl->out.reserve(target_vocab * 4);
  if (e == hipSuccess) e = sl->n_out.reserve(4);
  if (e == hipSuccess) e = hipMemcpy(sl->w2o.p, po, n_off * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && n_ids) e = hipMemcpy(sl->lists.p, pl, n_ids * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    slimt_hip_shortlist_destroy(sl);
    return fail((int)e, "shortlist upload: %s", hipGetErrorString(e));
  }
  *out = sl;
  return 0;
}

extern "C" int slimt_hip_shortlist_destroy(slimt_hip_shortlist *sl) {
  if (!sl) return 0;
  (void)hipSetDevice(sl->device);
  for (DevBuf *b : {&sl->w2o, &sl->lists, &sl->ids, &sl->lengths, &sl->out, &sl->n_out, &sl->scratch})
    b->release();
  delete sl;
  return 0;
}

extern "C" int slimt_hip_shortlist_info(const slimt_hip_shortlist *sl, uint64_t *frequent,
                                        uint64_t *best) {
  if (!sl) return fail(-1, "shortlist is NULL");
  if (frequent) *frequent = sl->frequent;
  if (best) *best = sl->best;
  return 0;
}

extern "C" int slimt_hip_shortlist_generate(slimt_hip_shortlist *sl, const uint32_t *src_ids,
                                            const uint32_t *lengths, size_t B, size_t S,
                                            uint32_t *out_ids, size_t *n_out) {
  if (!sl || !src_ids || !lengths || !out_ids || !n_out) return fail(-1, "null argument");
  if (B == 0 || S == 0) return fail(-1, "empty batch");
  if (B * S > (1u << 28)) return fail(-1, "batch too large");
  for (size_t b = 0; b < B; ++b) {
    if (lengths[b] > S) return fail(-1, "length %u > S", lengths[b]);
    for (size_t j = 0; j < lengths[b]; ++j)
      if (src_ids[b * S + j] >= sl->source_vocab)
        return fail(-1, "token id %u out of range", src_ids[b * S + j]);
  }
  // ShortlistGenerator::generate is const and is called by every Async worker on ONE shared
  // generator (Model.cc:117-120); this entry point stages through per-handle buffers, so
  // concurrent callers take turns (the _device variants use the caller's context instead)
  std::lock_guard<std::mutex> lock(sl->mu);
  HIPCHK(hipSetDevice(sl->device));
  HIPCHK(sl->ids.reserve(B * S * 4));
  HIPCHK(sl->lengths.reserve(B * 4));
  HIPCHK(hipMemcpy(sl->ids.p, src_ids, B * S * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(sl->lengths.p, lengths, B * 4, hipMemcpyHostToDevice));
  RCCHK(shortlist_scratch(sl->scratch, sl, nullptr));
  ShortlistArgs a;
  shortlist_args(sl, sl->ids.as<uint32_t>(), sl->lengths.as<uint32_t>(), B, S,
                 sl->out.as<uint32_t>(), sl->n_out.as<uint32_t>(), a);
  a.scratch = sl->scratch.as<uint32_t>();
  HIPCHK(launch_shortlist_generate(a, nullptr));
  uint32_t n = 0;
  HIPCHK(hipMemcpy(&n, sl->n_out.p, 4, hipMemcpyDeviceToHost));
  if (n > sl->target_vocab) return fail(-1, "shortlist kernel returned %u ids", n);
  if (n) HIPCHK(hipMemcpy(out_ids, sl->out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  *n_out = n;
  return 0;
}

extern "C" int slimt_hip_shortlist_generate_device(slimt_hip_shortlist *sl, slimt_hip_ctx *ctx,
                                                   const uint32_t *d_src_ids,
                                                   const uint32_t *d_lengths, size_t B, size_t S,
                                                   uint32_t *d_out_ids, uint32_t *d_n_out) {
  if (!sl || !ctx || !d_src_ids || !d_lengths || !d_out_ids || !d_n_out)
    return fail(-1, "null argument");
  if (B == 0 || S == 0) return fail(-1, "empty batch");
  if (sl->device != ctx->model->device) return fail(-1, "shortlist and context are on different devices");
  HIPCHK(hipSetDevice(sl->device));
  RCCHK(shortlist_scratch(ctx->sl_scratch, sl, ctx->stream));
  ShortlistArgs a;
  shortlist_args(sl, d_src_ids, d_lengths, B, S, d_out_ids, d_n_out, a);
  a.scratch = ctx->sl_scratch.as<uint32_t>();
  HIPCHK(launch_shortlist_generate(a, ctx->stream));
  return 0;
}

extern "C" int slimt_hip_translate_device_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl,
                                                    const uint32_t *d_src_ids,
                                                    const uint32_t *d_lengths, size_t B, size_t S,
                                                    float limit_factor, uint32_t eos_id,
                                                    uint32_t *d_out_ids, uint32_t *d_out_len,
                                                    float *d_align, int steps_hint) {
  if (!ctx || !sl || !d_src_ids || !d_lengths || !d_out_ids || !d_out_len)
    return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  const slimt_hip_model *m = ctx->model;
  if (sl->device != m->device) return fail(-1, "shortlist and context are on different devices");
  if (sl->target_vocab != (size_t)m->V)
    return fail(-1, "shortlist target vocabulary %zu != model vocabulary %d", sl->target_vocab, m->V);
  HIPCHK(hipSetDevice(m->device));
  return translate_generated(ctx, sl, d_src_ids, d_lengths, B, S, limit_factor, eos_id, d_out_ids, d_out_len,
                             d_align, steps_hint, nullptr);
}

namespace {
// Model::forward (Model.cc:111-204) on HOST buffers with the batch's lexical shortlist generated on
// the device (Model.cc:117-120): no host shortlist, no upload, no synchronisation in the
// asynchronous form.
int translate_host_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const uint32_t *src_ids,
                             const uint32_t *lengths, size_t B, size_t S, float limit_factor,
                             uint32_t eos_id, uint32_t *out_ids, uint32_t *out_len, float *align, bool wait) {
  if (!ctx || !sl || !src_ids || !lengths || !out_ids || !out_len) return fail(-1, "null argument");
  RCCHK(check_batch(ctx, B, S));
  const slimt_hip_model *m = ctx->model;
  if (sl->device != m->device) return fail(-1, "shortlist and context are on different devices");
  if (sl->target_vocab != (size_t)m->V)
    return fail(-1, "shortlist target vocabulary %zu != model vocabulary %d", sl->target_vocab, m->V);
  const uint32_t vmax = (uint32_t)std::min((size_t)m->V, sl->source_vocab);
  for (size_t i = 0; i < B * S; ++i)
    if (src_ids[i] >= vmax) return fail(-1, "token id %u out of range", src_ids[i]);
  for (size_t i = 0; i < B; ++i)
    if (lengths[i] > S) return fail(-1, "length %u > S", lengths[i]);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = ctx->stream;
  const size_t Tmax = std::max<size_t>(1, (size_t)(limit_factor * (float)S));
  const bool persistent = fused_decoder_allowed(ctx) && fused_decode_supported(m->D, m->F, m->H, m->Ld) &&
                          (fused_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S) ||
                           long_encode_supported(m->D, m->F, m->H, m->Le, m->Ld, (int)S));
  if (!wait && persistent) {  // pinned buffers: the kernels read and write host memory themselves (translate_host)
    void *v_ids = host_device_view(src_ids), *v_len = host_device_view(lengths), *v_out = host_device_view(out_ids),
         *v_ol = host_device_view(out_len), *v_al = align ? host_device_view(align) : nullptr;
    if (v_ids && v_len && v_out && v_ol && (!align || v_al)) {
      if (align) HIPCHK(ctx->align.reserve(align_staging_bytes(ctx, B, S, Tmax, limit_factor)));
      return translate_generated(ctx, sl, static_cast<const uint32_t *>(v_ids), static_cast<const uint32_t *>(v_len),
                                 B, S, limit_factor, eos_id, static_cast<uint32_t *>(v_out),
                                 static_cast<uint32_t *>(v_ol), align ? ctx->align.as<float>() : nullptr,
                                 (int)Tmax, static_cast<float *>(v_al));
    }
  }
  HIPCHK(ctx->out_ids.reserve(B * Tmax * 4));
  if (align) HIPCHK(ctx->align.reserve(B * Tmax * S * 4));
  HIPCHK(hipMemcpyAsync(ctx->ids.p, src_ids, B * S * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->lengths.p, lengths, B * 4, hipMemcpyHostToDevice, st));
  RCCHK(translate_generated(ctx, sl, ctx->ids.as<uint32_t>(), ctx->lengths.as<uint32_t>(), B, S, limit_factor,
                            eos_id, ctx->out_ids.as<uint32_t>(), ctx->out_len.as<uint32_t>(),
                            align ? ctx->align.as<float>() : nullptr, wait ? 0 : (int)Tmax, nullptr));
  HIPCHK(hipMemcpyAsync(out_ids, ctx->out_ids.p, B * Tmax * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out_len, ctx->out_len.p, B * 4, hipMemcpyDeviceToHost, st));
  if (align) HIPCHK(hipMemcpyAsync(align, ctx->align.p, B * Tmax * S * 4, hipMemcpyDeviceToHost, st));
  if (wait) RCCHK(slimt_hip_ctx_synchronize(ctx));  // sleeps on a blocking-sync event (a worker per context: no spinning)
  return 0;
}
}  // namespace

// ---- merged launches whose batches each get THEIR lexical shortlist, generated inside the encoder launch ---------------
namespace {
// dev: the batches with arrays the device can read (device memory, or the device views of pinned arrays); the merged launch
// when it is possible, else batch by batch through translate_generated
int translate_many_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const slimt_hip_batch *dev, size_t n, size_t S,
                             float limit_factor, uint32_t eos_id, int steps_hint, bool stage_align,
                             const slimt_hip_batch *host = nullptr) {
  // host (the asynchronous entry point): the same batches with the caller's HOST pointers -- what the batch-by-batch
  // fallback hands to translate_host_generated, which knows when the kernels can use pinned arrays in place
  const slimt_hip_model *m = ctx->model;
  const size_t Tmax = std::max<size_t>(1, (size_t)(limit_factor * (float)S));
  const size_t V = (size_t)m->V;
  size_t rows = 0;
  bool any_align = false;
  uint32_t *hint = reinterpret_cast<uint32_t *>(ctx->n_finished_host) + 1;  // (translate_generated: the previous shortlist's size)
  const size_t n_hint = *hint;
  const size_t tile = merge_tile(ctx, n_hint ? n_hint : V);  // (translate_device picks the decoder tiling from the same number)
  for (size_t j = 0; j < n; ++j) {
    rows += (dev[j].B + tile - 1) / tile * tile;
    any_align = any_align || dev[j].align != nullptr;
  }
  static const bool fold = !(std::getenv("SLIMT_SHORTLIST_FOLD") && std::getenv("SLIMT_SHORTLIST_FOLD")[0] == '0');
  const bool merged = n > 1 && n <= (size_t)kMaxMerge && S <= ctx->max_S && rows <= ctx->max_B && rows * S <= ctx->max_M &&
                      merge_supported(ctx, rows, S) && fold &&
                      shortlist_in_launch_lds_bytes((int)sl->source_vocab, (int)sl->target_vocab) <= 64 * 1024;
  if (!merged && host) {
    for (size_t j = 0; j < n; ++j) {
      const slimt_hip_batch &b = host[j];
      RCCHK(translate_host_generated(ctx, sl, b.src_ids, b.lengths, b.B, b.S ? b.S : S, limit_factor, eos_id, b.out_ids, b.out_len,
                                     b.align, false));
    }
    return 0;
  }
  if (!merged) {
    for (size_t j = 0; j < n; ++j) {
      const slimt_hip_batch &b = dev[j];
      const size_t Sj = b.S ? b.S : S, Tj = std::max<size_t>(1, (size_t)(limit_factor * (float)Sj));
      float *staging = nullptr;
      if (b.align && stage_align) {
        HIPCHK(ctx->align.reserve(align_staging_bytes(ctx, b.B, Sj, Tj, limit_factor)));
        staging = ctx->align.as<float>();
      }
      RCCHK(check_batch(ctx, b.B, Sj));
      RCCHK(translate_generated(ctx, sl, b.src_ids, b.lengths, b.B, Sj, limit_factor, eos_id, b.out_ids, b.out_len,
                                staging ? staging : b.align, steps_hint > 0 ? std::min(steps_hint, (int)Tj) : (stage_align ? (int)Tj : 0),
                                staging ? b.align : nullptr));
    }
    return 0;
  }
  hipStream_t st = ctx->stream;
  ctx->sl_host.clear();  // ctx->shortlist no longer holds what translate_host uploaded last
  HIPCHK(ctx->shortlist.reserve(n * V * 4));  // job j's ids at j * V, its count at n_sl_dev[j]
  HIPCHK(ctx->n_sl_dev.reserve(4 * (size_t)kMaxMerge));
  slimt_hip_batch plan[kMaxMerge];
  for (size_t j = 0; j < n; ++j) {
    plan[j] = dev[j];
    plan[j].shortlist = ctx->shortlist.as<uint32_t>() + j * V;
    plan[j].n_shortlist = V;  // the capacity; the kernels read the count
  }
  if (any_align && stage_align) HIPCHK(ctx->align.reserve(align_staging_bytes(ctx, rows, S, Tmax, limit_factor)));
  MergePlan mp;
  RCCHK(build_merge_plan(ctx, plan, n, S, Tmax, limit_factor, steps_hint, any_align && stage_align ? ctx->align.as<float>() : nullptr, mp, rows, tile));
  ShortlistArgs a;
  shortlist_args(sl, plan[0].src_ids, plan[0].lengths, plan[0].B, S, ctx->shortlist.as<uint32_t>(), ctx->n_sl_dev.as<uint32_t>(), a);
  void *hint_dev = nullptr;
  if (hipHostGetDevicePointer(&hint_dev, hint, 0) == hipSuccess) a.n_out_host = static_cast<uint32_t *>(hint_dev);
  (void)st;
  return translate_device(ctx, plan[0].src_ids, plan[0].lengths, ctx->shortlist.as<uint32_t>(), rows, S, V, limit_factor, eos_id,
                          plan[0].out_ids, plan[0].out_len, any_align ? (stage_align ? ctx->align.as<float>() : plan[0].align) : nullptr,
                          steps_hint > 0 ? steps_hint : (int)Tmax, ctx->n_sl_dev.as<uint32_t>(),
                          any_align && stage_align ? plan[0].align : nullptr, n_hint, &a, &mp);
}

int check_generator(const slimt_hip_ctx *ctx, const slimt_hip_shortlist *sl) {
  const slimt_hip_model *m = ctx->model;
  if (sl->device != m->device) return fail(-1, "shortlist and context are on different devices");
  if (sl->target_vocab != (size_t)m->V)
    return fail(-1, "shortlist target vocabulary %zu != model vocabulary %d", sl->target_vocab, m->V);
  return 0;
}
}  // namespace

extern "C" int slimt_hip_translate_many_device_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const slimt_hip_batch *batches,
                                                         size_t n_batches, size_t S, float limit_factor, uint32_t eos_id,
                                                         int steps_hint) {
  if (!ctx || !sl || !batches || n_batches == 0) return fail(-1, "null argument");
  RCCHK(check_generator(ctx, sl));
  for (size_t j = 0; j < n_batches; ++j) {
    const slimt_hip_batch &b = batches[j];
    if (!b.src_ids || !b.lengths || !b.out_ids || !b.out_len || b.B == 0) return fail(-1, "batch %zu: null array or empty", j);
    if ((b.S ? b.S : S) > S) return fail(-1, "batch %zu is padded to %zu tokens, the launch to %zu", j, b.S, S);
  }
  HIPCHK(hipSetDevice(ctx->model->device));
  return translate_many_generated(ctx, sl, batches, n_batches, S, limit_factor, eos_id, steps_hint, false);
}

extern "C" int slimt_hip_translate_many_async_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const slimt_hip_batch *batches,
                                                        size_t n_batches, size_t S, float limit_factor, uint32_t eos_id) {
  if (!ctx || !sl || !batches || n_batches == 0) return fail(-1, "null argument");
  RCCHK(check_generator(ctx, sl));
  const slimt_hip_model *m = ctx->model;
  HIPCHK(hipSetDevice(m->device));
  const uint32_t vmax = (uint32_t)std::min((size_t)m->V, sl->source_vocab);
  slimt_hip_batch dev[kMaxMerge];
  bool pinned = n_batches <= (size_t)kMaxMerge;
  for (size_t j = 0; j < n_batches; ++j) {
    const slimt_hip_batch &b = batches[j];
    if (!b.src_ids || !b.lengths || !b.out_ids || !b.out_len || b.B == 0) return fail(-1, "batch %zu: null array or empty", j);
    const size_t Sj = b.S ? b.S : S;
    if (Sj > S) return fail(-1, "batch %zu is padded to %zu tokens, the launch to %zu", j, Sj, S);
    for (size_t i = 0; i < b.B * Sj; ++i)
      if (b.src_ids[i] >= vmax) return fail(-1, "batch %zu: token id %u out of range", j, b.src_ids[i]);
    for (size_t i = 0; i < b.B; ++i)
      if (b.lengths[i] > Sj) return fail(-1, "batch %zu: length %u > S", j, b.lengths[i]);
    if (!pinned) continue;
    dev[j] = b;
    dev[j].src_ids = static_cast<const uint32_t *>(host_device_view(b.src_ids));
    dev[j].lengths = static_cast<const uint32_t *>(host_device_view(b.lengths));
    dev[j].out_ids = static_cast<uint32_t *>(host_device_view(b.out_ids));
    dev[j].out_len = static_cast<uint32_t *>(host_device_view(b.out_len));
    dev[j].align = b.align ? static_cast<float *>(host_device_view(b.align)) : nullptr;
    if (!dev[j].src_ids || !dev[j].lengths || !dev[j].out_ids || !dev[j].out_len || (b.align && !dev[j].align)) pinned = false;
  }
  if (pinned) return translate_many_generated(ctx, sl, dev, n_batches, S, limit_factor, eos_id, 0, true, batches);
  for (size_t j = 0; j < n_batches; ++j) {  // pageable arrays, or too many batches: one by one through the copying path
    const slimt_hip_batch &b = batches[j];
    RCCHK(translate_host_generated(ctx, sl, b.src_ids, b.lengths, b.B, b.S ? b.S : S, limit_factor, eos_id, b.out_ids, b.out_len,
                                   b.align, false));
  }
  return 0;
}

extern "C" int slimt_hip_translate_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl, const uint32_t *src_ids,
                                             const uint32_t *lengths, size_t B, size_t S, float limit_factor,
                                             uint32_t eos_id, uint32_t *out_ids, uint32_t *out_len, float *align) {
  return translate_host_generated(ctx, sl, src_ids, lengths, B, S, limit_factor, eos_id, out_ids, out_len, align, true);
}

extern "C" int slimt_hip_translate_async_generated(slimt_hip_ctx *ctx, slimt_hip_shortlist *sl,
                                                   const uint32_t *src_ids, const uint32_t *lengths, size_t B,
                                                   size_t S, float limit_factor, uint32_t eos_id,
                                                   uint32_t *out_ids, uint32_t *out_len, float *align) {
  return translate_host_generated(ctx, sl, src_ids, lengths, B, S, limit_factor, eos_id, out_ids, out_len, align, false);
}
