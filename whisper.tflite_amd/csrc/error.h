// Failure type shared by the engine and the kernel launchers: carries the wt_status code the C ABI
// reports (include/wt_capi.h).  Nothing below the C ABI calls abort()/exit(): a shape the kernels do
// not support is an Error, which capi.cpp's guarded() turns into a status code + wt_last_error().
#pragma once
#include <stdexcept>
#include <string>

namespace wt {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// status codes used below the ABI (mirror enum wt_status)
constexpr int kErrInvalidArg = 1, kErrIo = 2, kErrFormat = 3, kErrUnsupported = 4, kErrDevice = 5;

}  // namespace wt
