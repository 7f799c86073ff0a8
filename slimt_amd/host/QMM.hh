// slimt::qmm with a HIP provider: the five free functions of the reference's
// compile-time provider facade (slimt/QMM.hh:48-63), forwarding to the C ABI
// of libslimt_hip.so exactly as a `Provider::Hip` specialisation in
// slimt/QMM.cc:3-34 would (INTEGRATION.md shows that .inl.cc).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "Tensor.hh"

namespace slimt::qmm {

constexpr float kInt8Maxf = 127.0F;

// x f32 [..., K]; W int8 in the prepared layout [N][K] with its f32 b_quant
// stored right after the payload; b f32 [1, N]. Result f32 [..., N].
// Like the reference, shape errors are programming errors: abort.
Tensor affine(const Tensor &x, const Tensor &W, const Tensor &b, float a_quant, float b_quant,
              const std::string &name = "");
Tensor affine_with_select(const Tensor &x, const Tensor &W, const Tensor &b, float a_quant,
                          float b_quant, const std::vector<uint32_t> &indices,
                          const std::string &name = "");
Tensor dot(const Tensor &x, const Tensor &W, float a_quant, float b_quant,
           const std::string &name = "");
void prepare_weight_transposed(const float *weights, int8_t *prepared,
                               float quantization_multiplier, size_t cols, size_t rows);
void prepare_weight_quantized_transposed(const int8_t *input, int8_t *output, size_t rows,
                                         size_t cols);

}  // namespace slimt::qmm
