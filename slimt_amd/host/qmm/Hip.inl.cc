// HIP provider of slimt::qmm (the analogue of slimt/qmm/Intgemm.inl.cc for
// this backend). The tensor W is logically [K, N] (W.dim(-2) = K,
// W.dim(-1) = N) and physically the prepared [N][K] layout.
#include <cstdio>
#include <cstdlib>

#include "slimt_hip.h"

namespace slimt::qmm {

namespace {
[[noreturn]] void die(const char *what) {
  std::fprintf(stderr, "slimt::qmm (hip): %s: %s\n", what, slimt_hip_last_error());
  std::abort();  // the reference's providers assert/abort as well (Intgemm.inl.cc:111)
}
}  // namespace

Tensor affine(const Tensor &x, const Tensor &W, const Tensor &b, float a_quant, float b_quant,
              const std::string &name) {
  const size_t K = x.dim(-1), N = W.dim(-1), M = x.size() / K;
  if (W.size() / N != K) die("affine: inner dimensions differ");
  Shape out_shape = x.shape();
  out_shape.set_dim(-1, N);
  Tensor y(Type::f32, out_shape, name.empty() ? x.name() : name);
  if (slimt_hip_affine(x.data<float>(), M, K, W.data<int8_t>(), N, b.data<float>(), a_quant, b_quant,
                       y.data<float>()))
    die("affine");
  return y;
}

Tensor affine_with_select(const Tensor &x, const Tensor &W, const Tensor &b, float a_quant,
                          float b_quant, const std::vector<uint32_t> &indices,
                          const std::string &name) {
  const size_t K = x.dim(-1), N = W.dim(-1), M = x.size() / K;
  if (W.size() / N != K) die("affine_with_select: inner dimensions differ");
  Shape out_shape = x.shape();
  out_shape.set_dim(-1, indices.size());
  Tensor y(Type::f32, out_shape, name.empty() ? x.name() : name);
  if (slimt_hip_affine_select(x.data<float>(), M, K, W.data<int8_t>(), N, b.data<float>(), a_quant,
                              b_quant, indices.data(), indices.size(), y.data<float>()))
    die("affine_with_select");
  return y;
}

Tensor dot(const Tensor &x, const Tensor &W, float a_quant, float b_quant, const std::string &name) {
  const size_t K = x.dim(-1), N = W.dim(-1), M = x.size() / K;
  if (W.size() / N != K) die("dot: inner dimensions differ");
  Shape out_shape = x.shape();
  out_shape.set_dim(-1, N);
  Tensor y(Type::f32, out_shape, name.empty() ? x.name() : name);
  if (slimt_hip_affine(x.data<float>(), M, K, W.data<int8_t>(), N, nullptr, a_quant, b_quant,
                       y.data<float>()))
    die("dot");
  return y;
}

void prepare_weight_transposed(const float *weights, int8_t *prepared,
                               float quantization_multiplier, size_t cols, size_t rows) {
  if (slimt_hip_prepare_weight_transposed(weights, prepared, quantization_multiplier, cols, rows))
    die("prepare_weight_transposed");
}

void prepare_weight_quantized_transposed(const int8_t *input, int8_t *output, size_t rows,
                                         size_t cols) {
  if (slimt_hip_prepare_weight_quantized_transposed(input, output, rows, cols))
    die("prepare_weight_quantized_transposed");
}

}  // namespace slimt::qmm
