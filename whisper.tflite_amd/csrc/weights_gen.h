// Synthetic-weight generator API (see weights_gen.cpp).
#pragma once
#include <cstdint>
#include <string>

#include "wtw_format.h"

namespace wtw {
// Named presets: "tiny", "tiny.en", "base", "micro" (test-sized).
bool dims_by_name(const char* name, Dims* out);
// Writes a complete .wtw file; returns 0 on success (1 bad dims, 2 I/O).
int write_synthetic(const char* path, const Dims& dims, uint64_t seed, std::string* err);
}  // namespace wtw
