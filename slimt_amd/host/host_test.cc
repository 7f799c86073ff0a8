// Test driver for the C++ host mirror (run by tests/test_gpu_host_cpp.py).
//   host_test <model.bin> <case.bin> <out.bin>
// case.bin: u32 {enc_layers, dec_layers, heads, B, S, n_shortlist, M, K, N},
//           f32 limit_factor, f32 a_quant, f32 b_quant, then u32 ids[B*S],
//           u32 lengths[B], u32 shortlist[n_sl], f32 x[M*K], i8 W[N*K],
//           f32 bias[N], u32 n_idx, u32 idx[n_idx]
// out.bin : per sentence u32 n, u32 tokens[n], f32 align[n][len];
//           then f32 affine[M*N], f32 dot[M*N], f32 select[M*n_idx]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <stdexcept>
#include <vector>

#include "Model.hh"
#include "Shortlist.hh"
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

int main(int argc, char **argv) {
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
