// Minimal host tensor with the slice of slimt::Tensor's interface that the
// hot-path boundary uses (slimt/Tensor.hh:46-154): a typed, shaped, named
// buffer that is either owned (64-byte aligned, like slimt/Aligned.hh:7) or a
// borrowed view. Written for this backend; not a copy of the reference class.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace slimt {

enum class Type { i8, ig8, i32, u32, f32 };  // slimt/Tensor.hh:16-22

inline size_t size_in_bytes(Type t) { return (t == Type::i8 || t == Type::ig8) ? 1 : 4; }

class Shape {
 public:
  Shape() = default;
  Shape(std::initializer_list<size_t> dims) : dims_(dims) {}
  explicit Shape(std::vector<size_t> dims) : dims_(std::move(dims)) {}
  size_t dim(int i) const {  // negative = from the back, like slimt::Shape::dim
    const int n = static_cast<int>(dims_.size());
    const int k = i < 0 ? n + i : i;
    assert(k >= 0 && k < n);
    return dims_[static_cast<size_t>(k)];
  }
  void set_dim(int i, size_t v) {
    const int n = static_cast<int>(dims_.size());
    dims_[static_cast<size_t>(i < 0 ? n + i : i)] = v;
  }
  size_t elements() const {
    size_t n = 1;
    for (size_t d : dims_) n *= d;
    return n;
  }
  size_t rank() const { return dims_.size(); }
  const std::vector<size_t> &dims() const { return dims_; }
  bool operator==(const Shape &o) const { return dims_ == o.dims_; }

 private:
  std::vector<size_t> dims_;
};

class Tensor {
 public:
  Tensor() = default;
  // Owning: 64-byte aligned, size rounded up to 64 B (slimt/Aligned.cc:45-52).
  // `extra_bytes` leaves room for the trailing quantisation multiplier that
  // int8 weights carry right after their payload (slimt/Modules.cc:18-22).
  Tensor(Type type, Shape shape, std::string name = "", size_t extra_bytes = 0)
      : type_(type), shape_(std::move(shape)), name_(std::move(name)) {
    const size_t bytes = shape_.elements() * size_in_bytes(type_) + extra_bytes;
    const size_t rounded = ((bytes + 63) / 64) * 64;
    void *p = std::aligned_alloc(64, rounded ? rounded : 64);
    std::memset(p, 0, rounded ? rounded : 64);
    owner_ = std::shared_ptr<void>(p, std::free);
    data_ = p;
  }
  // Borrowed view (weights living in an mmap, slimt/Tensor.hh:104-110).
  static Tensor view(void *data, Type type, Shape shape, std::string name = "") {
    Tensor t;
    t.type_ = type;
    t.shape_ = std::move(shape);
    t.name_ = std::move(name);
    t.data_ = data;
    return t;
  }
  template <class T>
  T *data() { return reinterpret_cast<T *>(data_); }
  template <class T>
  const T *data() const { return reinterpret_cast<const T *>(data_); }
  template <class T>
  const T *end() const { return data<T>() + size(); }
  template <class T>
  T item() const { return *data<T>(); }
  size_t dim(int i) const { return shape_.dim(i); }
  size_t size() const { return shape_.elements(); }
  const Shape &shape() const { return shape_; }
  Type type() const { return type_; }
  const std::string &name() const { return name_; }
  void fill_in_place(float v) {
    float *p = data<float>();
    for (size_t i = 0; i < size(); ++i) p[i] = v;
  }

 private:
  Type type_ = Type::f32;
  Shape shape_;
  std::string name_;
  std::shared_ptr<void> owner_;
  void *data_ = nullptr;
};

}  // namespace slimt
