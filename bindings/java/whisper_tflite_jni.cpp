// JNI shim for io.github.jerinphilip.whisper.EngineNative over the MI355X engine's C ABI.
//
// Stands where the reference's bindings/java/whisper.tflite.cpp:17-71 stands: the four natives that
// io/github/jerinphilip/whisper/EngineNative.java declares (create, destroy, transcribeBuffer, transcribeFile;
// library names from EngineNative.java:35-38: "whisper-tflite" then "whisper-tflite-jni").  It binds
// include/wt_capi.h directly — the jlong handle is the wt_engine* — so the Java package runs unchanged on the
// HIP engine.  NOT BUILT in this repository's container (no JDK: no jni.h, no javac); build where one exists:
//
//   g++ -std=c++17 -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
//       bindings/java/whisper_tflite_jni.cpp -Lwhisper.tflite_amd/lib -lwhisper-tflite \
//       -o libwhisper-tflite-jni.so
#include <jni.h>

#include <string>
#include <vector>

#include "wt_capi.h"

namespace {

// RAII view of a Java string's UTF-8 bytes
class Utf8 {
 public:
  Utf8(JNIEnv* env, jstring s) : env_(env), s_(s), p_(s ? env->GetStringUTFChars(s, nullptr) : nullptr) {}
  ~Utf8() {
    if (p_) env_->ReleaseStringUTFChars(s_, p_);
  }
  const char* get() const { return p_ ? p_ : ""; }

 private:
  JNIEnv* env_;
  jstring s_;
  const char* p_;
};

wt_engine* engine_of(jlong native_ptr) { return reinterpret_cast<wt_engine*>(static_cast<intptr_t>(native_ptr)); }

// runs one of the two transcribe entry points with a growing text buffer; "" on failure, as
// Engine::transcribe returns (whisper.cpp:760)
template <class Call>
jstring text_of(JNIEnv* env, wt_engine* h, Call&& call) {
  std::string text(8192, '\0');
  size_t len = 0;
  int rc = call(&text[0], text.size(), &len);
  if (rc == WT_ERR_BUFFER) {
    text.assign(len + 1, '\0');
    rc = call(&text[0], text.size(), &len);
  }
  if (rc != WT_OK) {
    std::fprintf(stderr, "transcribe failed: %s\n", wt_last_error(h));
    return env->NewStringUTF("");
  }
  text.resize(len);
  return env->NewStringUTF(text.c_str());
}

}  // namespace

extern "C" {

// EngineNative.create(long engineType, String modelPath, String vocabPath, boolean isMultilingual) -> handle
// (0 when the engine could not be created; the reference returns the raw Engine*, nullptr likewise)
JNIEXPORT jlong JNICALL Java_io_github_jerinphilip_whisper_EngineNative_create(JNIEnv* env, jobject, jlong engine_type,
                                                                               jstring model_path, jstring vocab_path,
                                                                               jboolean is_multilingual) {
  const Utf8 model(env, model_path), vocab(env, vocab_path);
  wt_engine* h = nullptr;
  const int rc = wt_engine_create(static_cast<int>(engine_type), model.get(), vocab.get(), is_multilingual ? 1 : 0,
                                  /*device_id=*/0, &h);
  if (rc != WT_OK) std::fprintf(stderr, "EngineNative.create: %s\n", wt_last_error(nullptr));
  return static_cast<jlong>(reinterpret_cast<intptr_t>(h));
}

JNIEXPORT void JNICALL Java_io_github_jerinphilip_whisper_EngineNative_destroy(JNIEnv*, jobject, jlong native_ptr) {
  wt_engine_destroy(engine_of(native_ptr));
}

JNIEXPORT jstring JNICALL Java_io_github_jerinphilip_whisper_EngineNative_transcribeBuffer(JNIEnv* env, jobject,
                                                                                           jlong native_ptr,
                                                                                           jfloatArray samples) {
  wt_engine* h = engine_of(native_ptr);
  const jsize n = env->GetArrayLength(samples);
  std::vector<float> pcm(static_cast<size_t>(n));
  env->GetFloatArrayRegion(samples, 0, n, pcm.data());
  return text_of(env, h, [&](char* out, size_t cap, size_t* len) {
    return wt_transcribe_pcm(h, pcm.data(), pcm.size(), out, cap, len);
  });
}

JNIEXPORT jstring JNICALL Java_io_github_jerinphilip_whisper_EngineNative_transcribeFile(JNIEnv* env, jobject,
                                                                                         jlong native_ptr,
                                                                                         jstring wave_file) {
  wt_engine* h = engine_of(native_ptr);
  const Utf8 path(env, wave_file);
  return text_of(env, h, [&](char* out, size_t cap, size_t* len) {
    return wt_transcribe_file(h, path.get(), out, cap, len);
  });
}

}  // extern "C"
