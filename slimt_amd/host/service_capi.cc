// extern "C" face of host/Service (include/slimt_hip_service.h).
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <vector>

#include "Service.hh"
#include "slimt_hip_service.h"

namespace {
thread_local char g_err[512] = "";
int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
}  // namespace

struct slimt_hip_service {
  std::vector<std::unique_ptr<slimt::Model>> models;  // non-owning views of the caller's replicas
  std::unique_ptr<slimt::Service> service;
};

struct slimt_hip_result {
  std::vector<uint32_t> targets, padded;
  std::vector<uint64_t> target_offsets, batch, align_offsets;
  std::vector<float> alignments;
};

extern "C" const char *slimt_hip_service_last_error(void) { return g_err; }

extern "C" int slimt_hip_service_create(const slimt_hip_service_config *config, slimt_hip_model *const *replicas,
                                        size_t n_replicas, slimt_hip_service **out) {
  if (!config || !replicas || !out || n_replicas == 0) return fail("null argument");
  *out = nullptr;
  try {
    auto s = std::make_unique<slimt_hip_service>();
    std::vector<const slimt::Model *> views;
    for (size_t i = 0; i < n_replicas; ++i) {
      if (!replicas[i]) return fail("replica %zu is NULL", i);
      slimt::Model::Config mc;
      mc.eos_id = config->eos_id;
      s->models.push_back(std::make_unique<slimt::Model>(mc, replicas[i]));
      views.push_back(s->models.back().get());
    }
    slimt::ServiceConfig sc;
    sc.max_words = config->max_words;
    sc.wrap_length = config->wrap_length;
    sc.tgt_length_limit_factor = config->limit_factor;
    sc.workers_per_device = config->workers_per_device;
    sc.pad_id = config->pad_id;
    if (config->merge_batches) sc.merge_batches = config->merge_batches;
    if (config->merge_words) sc.merge_words = config->merge_words;
    sc.alignments = config->alignments != 0;
    sc.flat_alignments = true;  // arrays out: one block per sentence
    if (config->lexical_shortlist && config->lexical_shortlist_bytes) {
      sc.lexical_shortlist = slimt::View{config->lexical_shortlist, static_cast<size_t>(config->lexical_shortlist_bytes)};
      sc.source_vocab = config->source_vocab;
      sc.target_vocab = config->target_vocab;
      sc.shortlist_shared_vocab = config->shortlist_shared_vocab != 0;
      sc.shortlist_check = config->shortlist_check != 0;
    } else if (config->shortlist && config->n_shortlist) {
      sc.shortlist = slimt::Words(config->shortlist, config->shortlist + config->n_shortlist);
    }
    s->service = std::make_unique<slimt::Service>(sc, views);
    *out = s.release();
    return 0;
  } catch (const std::exception &e) {
    return fail("%s", e.what());
  }
}

extern "C" int slimt_hip_service_destroy(slimt_hip_service *service) {
  delete service;
  return 0;
}

extern "C" int slimt_hip_service_translate(slimt_hip_service *service, const uint32_t *tokens, const uint64_t *offsets,
                                           size_t n, slimt_hip_result **out) {
  if (!service || !out || (n && (!tokens || !offsets))) return fail("null argument");
  *out = nullptr;
  try {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<slimt::Words> sentences(n);
    for (size_t i = 0; i < n; ++i) {
      if (offsets[i + 1] < offsets[i]) return fail("offsets decrease at sentence %zu", i);
      sentences[i].assign(tokens + offsets[i], tokens + offsets[i + 1]);
    }
    const auto t1 = std::chrono::steady_clock::now();
    slimt::Histories hs = service->service->translate(std::move(sentences)).get();
    const auto t2 = std::chrono::steady_clock::now();
    auto r = std::make_unique<slimt_hip_result>();
    r->target_offsets.assign(n + 1, 0);
    r->align_offsets.assign(n + 1, 0);
    r->padded.resize(n);
    r->batch.resize(n);
    size_t n_tok = 0, n_al = 0;
    for (size_t i = 0; i < n; ++i) {
      n_tok += hs[i]->target.size();
      n_al += hs[i]->alignment_flat.size();
    }
    r->targets.reserve(n_tok);
    r->alignments.reserve(n_al);
    for (size_t i = 0; i < n; ++i) {
      const slimt::Hypothesis &h = *hs[i];
      r->targets.insert(r->targets.end(), h.target.begin(), h.target.end());
      r->alignments.insert(r->alignments.end(), h.alignment_flat.begin(), h.alignment_flat.end());
      r->target_offsets[i + 1] = r->targets.size();
      r->align_offsets[i + 1] = r->alignments.size();
      r->padded[i] = static_cast<uint32_t>(h.padded_length);
      r->batch[i] = h.batch;
    }
    *out = r.release();
    if (std::getenv("SLIMT_SERVICE_STATS")) {
      const auto t3 = std::chrono::steady_clock::now();
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::fprintf(stderr, "service-call: %zu sentences: copy in %.1f ms, translate %.1f ms, flatten %.1f ms\n", n,
                   ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
    return 0;
  } catch (const std::exception &e) {
    return fail("%s", e.what());
  }
}

extern "C" int slimt_hip_result_view(const slimt_hip_result *r, size_t *n, const uint32_t **targets,
                                     const uint64_t **target_offsets, const uint32_t **padded_length,
                                     const uint64_t **batch, const float **alignments, const uint64_t **align_offsets) {
  if (!r) return fail("result is NULL");
  if (n) *n = r->padded.size();
  if (targets) *targets = r->targets.data();
  if (target_offsets) *target_offsets = r->target_offsets.data();
  if (padded_length) *padded_length = r->padded.data();
  if (batch) *batch = r->batch.data();
  if (alignments) *alignments = r->alignments.data();
  if (align_offsets) *align_offsets = r->align_offsets.data();
  return 0;
}

extern "C" int slimt_hip_result_destroy(slimt_hip_result *result) {
  delete result;
  return 0;
}
