// slimt/qmm/Hip.inl.cc -- the MI355X (gfx950) provider of slimt::qmm, in the form of the
// reference's other providers (slimt/qmm/Intgemm.inl.cc, Ruy.inl.cc, Gemmology.inl.cc):
// explicit specialisations of the five templates slimt/QMM.hh:24-44 declares, for
// Provider::Hip, textually included by slimt/QMM.cc under SLIMT_HAS_HIP (see
// integration/patches/0001-qmm-provider-hip.patch). Every function forwards to one entry
// point of libslimt_hip.so (include/slimt_hip.h): host tensors in, host tensors out.
//
// W is logically [K, N] (W.dim(-2) = K, W.dim(-1) = N) and physically the prepared layout,
// which for this provider IS the Marian intgemm8 file layout, int8 [N][K] (slimt/Io.cc:225-239);
// the trailing b_quant float stays where retrieve_quantization_multiplier (slimt/Modules.cc:18-22)
// reads it. A failing call aborts, as the reference's providers assert (Intgemm.inl.cc:111).
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "slimt/Tensor.hh"
#include "slimt_hip.h"

namespace slimt::qmm::detail {

namespace hip_provider {
[[noreturn]] inline void die(const char* what) {
  std::fprintf(stderr, "slimt::qmm (hip): %s: %s\n", what, slimt_hip_last_error());
  std::abort();
}

// x [..., K] against W [K, N]: the output keeps x's leading dimensions (Intgemm.inl.cc:101-104)
inline Tensor like_rows(const Tensor& x, size_t columns, const std::string& name) {
  Shape out = x.shape();
  out.set_dim(-1, static_cast<int>(columns));
  return Tensor(Type::f32, out, name.empty() ? x.name() : name);
}
}  // namespace hip_provider

template <>
Tensor affine<Provider::Hip>(const Tensor& x, const Tensor& W, const Tensor& b,
                             float a_quant, float b_quant,
                             const std::string& name) {
  const size_t K = x.dim(-1), N = W.dim(-1), M = x.size() / K;
  if (W.size() / N != K) hip_provider::die("affine: inner dimensions differ");
  Tensor y = hip_provider::like_rows(x, N, name);
  if (slimt_hip_affine(x.data<float>(), M, K, W.data<int8_t>(), N,
                       b.data<float>(), a_quant, b_quant, y.data<float>()) != 0)
    hip_provider::die("affine");
  return y;
}

template <>
Tensor affine_with_select<Provider::Hip>(const Tensor& x, const Tensor& W,
                                         const Tensor& b, float a_quant,
                                         float b_quant,
                                         const std::vector<uint32_t>& indices,
                                         const std::string& name) {
  const size_t K = x.dim(-1), N = W.dim(-1), M = x.size() / K;
  if (W.size() / N != K) hip_provider::die("affine_with_select: inner dimensions differ");
  Tensor y = hip_provider::like_rows(x, indices.size(), name);
  if (slimt_hip_affine_select(x.data<float>(), M, K, W.data<int8_t>(), N,
                              b.data<float>(), a_quant, b_quant, indices.data(),
                              indices.size(), y.data<float>()) != 0)
    hip_provider::die("affine_with_select");
  return y;
}

template <>
Tensor dot<Provider::Hip>(const Tensor& x, const Tensor& W, float a_quant,
                          float b_quant, const std::string& name) {
  const size_t K = x.dim(-1), N = W.dim(-1), M = x.size() / K;
  if (W.size() / N != K) hip_provider::die("dot: inner dimensions differ");
  Tensor y = hip_provider::like_rows(x, N, name);
  // bias == NULL: no bias term at all (the reference builds a zero bias, Intgemm.inl.cc:188-189)
  if (slimt_hip_affine(x.data<float>(), M, K, W.data<int8_t>(), N, nullptr,
                       a_quant, b_quant, y.data<float>()) != 0)
    hip_provider::die("dot");
  return y;
}

template <>
void prepare_weight_transposed<Provider::Hip>(const float* weights,
                                              int8_t* prepared,
                                              float quantization_multiplier,
                                              size_t cols, size_t rows) {
  if (slimt_hip_prepare_weight_transposed(weights, prepared,
                                          quantization_multiplier, cols, rows) != 0)
    hip_provider::die("prepare_weight_transposed");  // slimt/Io.cc:215
}

template <>
void prepare_weight_quantized_transposed<Provider::Hip>(const int8_t* input,
                                                        int8_t* output,
                                                        size_t rows,
                                                        size_t cols) {
  if (slimt_hip_prepare_weight_quantized_transposed(input, output, rows, cols) != 0)
    hip_provider::die("prepare_weight_quantized_transposed");  // slimt/Io.cc:234 (a copy)
}

}  // namespace slimt::qmm::detail
