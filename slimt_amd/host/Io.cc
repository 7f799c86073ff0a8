#include "Io.hh"

#include <cstring>
#include <stdexcept>

namespace slimt::io {

namespace {

struct Reader {
  const char *p;
  const char *end;
  template <class T>
  T get() {
    need(sizeof(T));
    T v;
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  const char *take(size_t n) {
    need(n);
    const char *q = p;
    p += n;
    return q;
  }
  void need(size_t n) const {
    if (static_cast<size_t>(end - p) < n) throw std::runtime_error("truncated Marian .bin");
  }
};

struct Header {  // slimt/Io.hh:24-29
  uint64_t name_length, type, shape_length, data_length;
};

ItemType decode_type(uint64_t t) {  // slimt/Io.cc:37-84
  switch (t) {
    case 0x0404: return ItemType::f32;
    case 0x0101: return ItemType::i8;
    case 0x4101: return ItemType::ig8;
    default: return ItemType::other;
  }
}

}  // namespace

std::vector<Item> load_items(const void *data, size_t size) {
  Reader r{static_cast<const char *>(data), static_cast<const char *>(data) + size};
  const uint64_t version = r.get<uint64_t>();
  if (version != kBinaryFileVersion)
    throw std::runtime_error("Marian .bin version " + std::to_string(version) + " != 1");
  const uint64_t n = r.get<uint64_t>();
  if (n > (1u << 20)) throw std::runtime_error("implausible item count");
  std::vector<Header> headers(n);
  for (auto &h : headers) h = r.get<Header>();
  std::vector<Item> items(n);
  for (uint64_t i = 0; i < n; ++i) {
    const char *name = r.take(headers[i].name_length);
    items[i].name.assign(name, headers[i].name_length ? headers[i].name_length - 1 : 0);
    items[i].type = decode_type(headers[i].type);
  }
  for (uint64_t i = 0; i < n; ++i) {
    items[i].shape.resize(headers[i].shape_length);
    for (auto &d : items[i].shape) d = r.get<int32_t>();
  }
  const uint64_t pad = r.get<uint64_t>();  // to a 256-byte boundary, slimt/Io.cc:151-153
  r.take(pad);
  for (uint64_t i = 0; i < n; ++i) {
    items[i].bytes = headers[i].data_length;
    items[i].data = r.take(headers[i].data_length);
    if (items[i].type == ItemType::other) continue;  // kept as an opaque view (e.g. special:model.yml)
    // the payload must cover the shape: consumers index it by rows * cols
    uint64_t elems = 1;
    for (int d : items[i].shape) {
      if (d <= 0) throw std::runtime_error("item " + items[i].name + " has a non-positive dimension");
      elems *= static_cast<uint64_t>(d);
      if (elems > (1ull << 40)) throw std::runtime_error("item " + items[i].name + " has an implausible shape");
    }
    const uint64_t need = items[i].type == ItemType::f32 ? elems * sizeof(float)
                          : items[i].type == ItemType::ig8 ? elems + sizeof(float) : elems;
    if (items[i].bytes < need)
      throw std::runtime_error((items[i].type == ItemType::ig8 ? "intgemm8 item " + items[i].name + " lacks its multiplier"
                                                               : "item " + items[i].name + " is shorter than its shape") +
                               " (" + std::to_string(items[i].bytes) + " < " + std::to_string(need) + " bytes)");
  }
  return items;
}

}  // namespace slimt::io
