/*
 * slimt_hip_service.h -- C ABI of the batching service of the MI355X backend (libslimt_hip_host.so,
 * over libslimt_hip.so): what the reference's bindings reach through slimt::Async
 * (slimt/Frontend.hh:20-78, Frontend.cc:207-227: requests -> token-budget batches -> workers ->
 * Model::forward -> per-sentence Histories), for callers that have tokenised text already (text
 * processing stays with the caller: Vocabulary / TextProcessor, slimt/TextProcessor.cc). It wraps
 * host/Service.{hh,cc}; the reference-side binding would be the same three calls from
 * bindings/python/slimt.cpp:44-135. Plain pointers and sizes; 0 = success, else
 * slimt_hip_service_last_error().
 */
#ifndef SLIMT_HIP_SERVICE_H
#define SLIMT_HIP_SERVICE_H

#include <stddef.h>
#include <stdint.h>

#include "slimt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slimt_hip_service slimt_hip_service;
typedef struct slimt_hip_result slimt_hip_result;

typedef struct slimt_hip_service_config {
  uint64_t max_words;          /* word budget of a batch: (B + 1) * S <= max_words (Batcher.cc:95-120) */
  uint64_t wrap_length;        /* longest sentence accepted (<= 128) */
  float limit_factor;          /* tgt_length_limit_factor (Model.cc:159-161) */
  uint32_t workers_per_device; /* double-buffered worker threads per model replica */
  uint32_t pad_id, eos_id;
  int32_t alignments;          /* != 0: return every sentence's alignment rows (Model.cc:84-108) */
  /* output vocabulary of a batch: the lexical shortlist file (generated per batch on the device,
   * Model.cc:117-120), else a fixed sorted id list, else (both empty) the full vocabulary */
  const void *lexical_shortlist;
  uint64_t lexical_shortlist_bytes;
  uint64_t source_vocab, target_vocab;
  int32_t shortlist_shared_vocab, shortlist_check;
  const uint32_t *shortlist;
  uint64_t n_shortlist;
  /* merged launches (slimt_hip_translate_many_async; host/Service.hh, ServiceConfig::merge_batches): a worker takes up
   * to merge_batches consecutive batches of one padded length -- each formed under max_words, each with its own results --
   * into one launch pair while their rows x length stay within merge_words. 0 = the defaults (8 batches, 8192 words);
   * merge_batches = 1: never. With a lexical shortlist every merged batch still gets its own list. */
  uint64_t merge_batches, merge_words;
} slimt_hip_service_config;

const char *slimt_hip_service_last_error(void); /* thread-local, never NULL */

/* replicas[i]: a model created by slimt_hip_model_create on some device; they stay the caller's
 * and must outlive the service. encoder / decoder layer counts and heads are the models' own. */
int slimt_hip_service_create(const slimt_hip_service_config *config, slimt_hip_model *const *replicas,
                             size_t n_replicas, slimt_hip_service **out);
int slimt_hip_service_destroy(slimt_hip_service *service); /* drains, joins the workers */

/* One request: n sentences, sentence i = tokens[offsets[i] .. offsets[i + 1]) (EOS included). Blocks
 * until every sentence is translated; thread-safe (any number of callers). */
int slimt_hip_service_translate(slimt_hip_service *service, const uint32_t *tokens, const uint64_t *offsets,
                                size_t n, slimt_hip_result **out);

/* The result of one request, owned by the handle:
 *  target_offsets [n + 1] into targets (target ids, EOS included),
 *  padded_length [n]: the S of the batch sentence i travelled in, batch [n]: that batch's serial number,
 *  align_offsets [n + 1] into alignments (floats): sentence i holds target_len(i) rows of
 *  source_len(i) probabilities, row-major (NULL / all zero offsets when the service returns none). */
int slimt_hip_result_view(const slimt_hip_result *result, size_t *n, const uint32_t **targets,
                          const uint64_t **target_offsets, const uint32_t **padded_length,
                          const uint64_t **batch, const float **alignments, const uint64_t **align_offsets);
int slimt_hip_result_destroy(slimt_hip_result *result);

#ifdef __cplusplus
}
#endif
#endif /* SLIMT_HIP_SERVICE_H */
