// Marian .bin (v1) container reader for the HIP backend. Same on-disk format
// the reference parses in slimt/Io.cc:114-161 (header, names, shapes, 256-byte
// aligned payloads; type ids slimt/Io.cc:37-84). Unlike the reference loader,
// no host-side re-layout happens here: int8 payloads are handed to
// slimt_hip_model_create as they are (the file layout [N][K] IS the prepared
// layout of this backend, and the Wemb dequantise / re-quantise of
// slimt/Io.cc:182-224 is done inside model_create).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace slimt::io {

constexpr uint64_t kBinaryFileVersion = 1;

enum class ItemType { f32, i8, ig8, other };

struct Item {
  std::string name;
  ItemType type = ItemType::other;
  std::vector<int> shape;
  const void *data = nullptr;  // view into the caller's buffer
  size_t bytes = 0;
};

// Throws std::runtime_error on a malformed container (the reference aborts).
std::vector<Item> load_items(const void *data, size_t size);

}  // namespace slimt::io
