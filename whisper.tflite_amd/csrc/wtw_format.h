// .wtw — the flat, aligned, mmap-able weight file that replaces the reference's
// `<prefix>.encoder.tflite` / `<prefix>.decoder.tflite` FlatBuffers
// (reference: whisper.tflite/whisper.cpp:743-744 opens the pair, :261-271 mmaps
// them through tflite::FlatBufferModel::BuildFromFile).
//
// Layout (little endian):
//   WtwHeader                         (fixed 128 bytes)
//   WtwTensor  table[n_tensors]       (128 bytes each)
//   payload, every tensor 256-byte aligned, fp32 row-major
//
// Tensor names follow the OpenAI Whisper module tree the reference's exporter
// traces (export/generate_onnx.py:85-120): "encoder.conv1.weight",
// "decoder.blocks.0.cross_attn.key.weight", ...
#pragma once
#include <cstdint>

namespace wtw {

static constexpr uint32_t kMagic = 0x31575457u;  // "WTW1"
static constexpr uint32_t kVersion = 1;
static constexpr uint32_t kAlign = 256;

// Model dimensions; names follow OpenAI's ModelDimensions.
struct Dims {
  int32_t n_mels;         // 80
  int32_t n_audio_ctx;    // 1500 (mel frames = 2 * n_audio_ctx)
  int32_t n_audio_state;  // 384 tiny, 512 base
  int32_t n_audio_head;   // 6 tiny, 8 base
  int32_t n_audio_layer;  // 4 tiny, 6 base
  int32_t n_vocab;        // 51865 multilingual, 51864 English
  int32_t n_text_ctx;     // 448
  int32_t n_text_state;
  int32_t n_text_head;
  int32_t n_text_layer;
};

struct WtwHeader {
  uint32_t magic;
  uint32_t version;
  uint32_t n_tensors;
  uint32_t table_offset;  // byte offset of the tensor table (== sizeof(WtwHeader))
  Dims dims;              // 40 bytes
  uint64_t payload_offset;
  uint64_t file_bytes;
  uint64_t seed;          // generator seed (0 when converted from real weights)
  uint8_t reserved[128 - 16 - 40 - 24];
};
static_assert(sizeof(WtwHeader) == 128, "WtwHeader must be 128 bytes");

struct WtwTensor {
  char name[80];
  uint32_t dtype;  // 0 = f32
  uint32_t ndim;
  uint32_t shape[4];
  uint64_t offset;  // from file start
  uint64_t nbytes;
  uint8_t reserved[128 - 80 - 8 - 16 - 16];
};
static_assert(sizeof(WtwTensor) == 128, "WtwTensor must be 128 bytes");

}  // namespace wtw
