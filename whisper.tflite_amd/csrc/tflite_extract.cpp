#include "tflite_extract.h"

#include <sys/stat.h>

#include "error.h"

namespace wt {

bool file_exists(const std::string& path) {
  struct stat st;
  return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

void convert_tflite(const std::string& model_prefix, const std::string& out_path) {
  (void)out_path;
  throw Error(kErrUnsupported, "tflite extractor: not built yet (" + model_prefix + ")");
}

}  // namespace wt
