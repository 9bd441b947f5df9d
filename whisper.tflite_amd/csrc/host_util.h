// Host-side (CPU, load-time / per-call trivial) pieces of the EncDec path that stay on
// the host in the MI355X build, mirroring the reference's own helpers:
//   WAV reader          whisper.tflite/wav_util.cpp:18-87
//   vocab+filter file   whisper.tflite/whisper.cpp:519-611, :218-226, :746-749
//   token -> text       whisper.tflite/whisper.cpp:634-665, :613-631
//   language table      whisper.tflite/whisper.cpp:405-517
// plus writers for the two on-disk assets, used to create synthetic
// fixtures where the upstream assets are absent.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace wt {

struct VocabData {
  std::map<int, std::string> id_to_token;
  int n_vocab = 51864;  // English defaults (reference whisper.h:69-91)
  int token_eot = 50256;
  int token_sot = 50257;
  int token_translate = 50358;
  int token_transcribe = 50359;
  int token_prev = 50360;
  int token_solm = 50361;
  int token_not = 50362;
  int token_beg = 50363;
};

struct FilterBank {
  int n_mel = 0;
  int n_fft = 0;  // number of frequency bins (201)
  std::vector<float> data;  // [n_mel][n_fft]
};

// Parses the vocab/filter file EncDec consumes.  Throws std::runtime_error when the file
// cannot be opened (as the reference's MmapFile does) or is truncated.
void read_vocab_file(const std::string& path, bool multilingual, FilterBank* filters,
                     VocabData* vocab);
// The same parse over memory that starts at the u32 magic (i.e. after the file's leading u64 payload size),
// as the reference's Reader does (whisper.cpp:519-611); never reads at or past `end`.
void parse_vocab(const char* head, const char* end, bool multilingual, FilterBank* filters, VocabData* vocab);
// whisper.cpp:218-226
void transform_vocab_multilingual(VocabData* vocab);

// Inverse of read_vocab_file: [u64 payload][u32 magic "USEN"][n_mel][n_fft][filters]
// [n_vocab]{u32 len, bytes}.
void write_vocab_file(const std::string& path, const FilterBank& filters,
                      const std::vector<std::string>& tokens);

// Slaney-normalised triangular mel filter bank (librosa.filters.mel defaults; what the
// upstream filters_vocab_*.bin assets contain).
FilterBank make_slaney_filterbank(int n_mel, int n_fft_size, int sample_rate);

// Synthetic token table: the 256 single-byte tokens followed by printable "<tN>" words.
std::vector<std::string> make_synthetic_tokens(int n_tokens);

// Returns false where the reference returns an empty vector (open failure / bad magic).
bool wav_read_legacy(const std::string& path, std::vector<float>* samples, bool verbose);
// Writes a canonical 44-byte-header PCM16 mono WAV (test/bench fixture helper).
bool wav_write_pcm16(const std::string& path, const std::vector<float>& samples, int sample_rate);

std::string decode_tokens(const VocabData& vocab, const int64_t* ids, int n,
                          bool omit_special_tokens, bool* missing);
std::string remove_extra_spaces(const std::string& in);

int language_count();
int language_id(const std::string& code);  // == language_count() when absent
const std::string& lang_code(size_t id);
const std::string& lang_name(size_t id);

}  // namespace wt
