// Weight extractor for the reference's model files: reads "<prefix>.encoder.tflite" and
// "<prefix>.decoder.tflite" (whisper.tflite/whisper.cpp:743-744; produced by export/generate_onnx.py:135-163)
// and writes the flat .wtw file the engine loads.  Host code, no TFLite / FlatBuffers library: the
// FlatBuffer is walked by hand (tflite_extract.cpp).
#pragma once
#include <string>

namespace wt {
bool file_exists(const std::string& path);
// throws wt::Error (kErrIo / kErrFormat) with a message naming what could not be identified
void convert_tflite(const std::string& model_prefix, const std::string& out_path);
}  // namespace wt
