// slimt::qmm with a HIP provider: the five free functions of the reference's
// compile-time provider facade (slimt/QMM.hh:48-63), forwarding to the C ABI
// of libslimt_hip.so exactly as a `Provider::Hip` specialisation in
// slimt/QMM.cc:3-34 would (INTEGRATION.md shows that .inl.cc). Call-compatible
// with the reference: same names, argument order and meaning.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "Tensor.hh"

namespace slimt::qmm {

using Label = std::string;
using ColumnIndices = std::vector<uint32_t>;

constexpr float kInt8Maxf = 127.0F;

// ---- load time (slimt/Io.cc:215,234) --------------------------------------
// Re-layout of a file weight (int8 [cols = N][rows = K], K contiguous) into the
// provider layout: a copy here, the prepared layout IS the file layout.
void prepare_weight_quantized_transposed(const int8_t *file_weight, int8_t *provider_weight,
                                         size_t rows, size_t cols);
// Quantise the f32 embedding matrix [cols = V][rows = D] with its own multiplier
// into the output layer's B matrix.
void prepare_weight_transposed(const float *f32_weight, int8_t *provider_weight, float multiplier,
                               size_t cols, size_t rows);

// ---- run time (slimt/Modules.cc:145-180) -----------------------------------
// activations f32 [..., K]; weight int8 in the prepared layout [N][K] with its
// f32 multiplier stored right after the payload; bias f32 [1, N]. Result f32
// [..., N] (leading dimensions flattened). Like the reference, shape errors are
// programming errors: abort.
Tensor dot(const Tensor &activations, const Tensor &weight, float activation_multiplier,
           float weight_multiplier, const Label &label = "");
Tensor affine(const Tensor &activations, const Tensor &weight, const Tensor &bias,
              float activation_multiplier, float weight_multiplier, const Label &label = "");
// logits over a sorted column subset (the batch's shortlist)
Tensor affine_with_select(const Tensor &activations, const Tensor &weight, const Tensor &bias,
                          float activation_multiplier, float weight_multiplier,
                          const ColumnIndices &columns, const Label &label = "");

}  // namespace slimt::qmm
