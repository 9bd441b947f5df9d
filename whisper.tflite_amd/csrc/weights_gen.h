// Synthetic-weight generator API (see weights_gen.cpp).
#pragma once
#include <cstdint>
#include <string>

#include "wtw_format.h"

#include <vector>

namespace wtw {
// One tensor of a .wtw file: OpenAI parameter path (wtw_format.h), shape, fp32 row-major data.
struct NamedTensor {
  std::string name;
  std::vector<uint32_t> shape;
  std::vector<float> data;
};
// Names and shapes of every tensor a model of these dims carries, in file order (data left empty).
std::vector<NamedTensor> tensor_specs(const Dims& dims);
// Writes a .wtw file from complete tensors; returns 0 on success (1 bad input, 2 I/O).
int write_tensors(const char* path, const Dims& dims, const std::vector<NamedTensor>& tensors, uint64_t seed,
                  std::string* err);
// Named presets: "tiny", "tiny.en", "base", "micro" (test-sized).
bool dims_by_name(const char* name, Dims* out);
// Writes a complete .wtw file; returns 0 on success (1 bad dims, 2 I/O).
int write_synthetic(const char* path, const Dims& dims, uint64_t seed, std::string* err);
}  // namespace wtw
