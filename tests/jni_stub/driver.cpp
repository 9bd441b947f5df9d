// TEST DRIVER for bindings/java/whisper_tflite_jni.cpp over tests/jni_stub/jni.h: plays the part of
// io/github/jerinphilip/whisper/EngineNative.java (create -> transcribeBuffer / transcribeFile -> destroy, reference
// Driver.java:4-27) without a JVM.  usage: driver <engine_type> <model_prefix> <vocab> <multilingual> [pcm.f32 [wav]]
// Prints "handle <0|1>", then one line per transcript: "buffer: <text>" / "file: <text>".
#include <jni.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>

extern "C" {
jlong Java_io_github_jerinphilip_whisper_EngineNative_create(JNIEnv*, jobject, jlong, jstring, jstring, jboolean);
void Java_io_github_jerinphilip_whisper_EngineNative_destroy(JNIEnv*, jobject, jlong);
jstring Java_io_github_jerinphilip_whisper_EngineNative_transcribeBuffer(JNIEnv*, jobject, jlong, jfloatArray);
jstring Java_io_github_jerinphilip_whisper_EngineNative_transcribeFile(JNIEnv*, jobject, jlong, jstring);
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  JNIEnv env;
  _jstring model, vocab;
  model.utf8 = argv[2];
  vocab.utf8 = argv[3];
  const jlong h = Java_io_github_jerinphilip_whisper_EngineNative_create(&env, nullptr, std::atol(argv[1]), &model, &vocab,
                                                                         std::atoi(argv[4]) ? 1 : 0);
  std::printf("handle %d\n", h != 0 ? 1 : 0);
  if (h != 0 && argc > 5) {
    std::ifstream f(argv[5], std::ios::binary);
    _jfloatArray pcm;
    f.seekg(0, std::ios::end);
    pcm.data.resize(static_cast<size_t>(f.tellg()) / sizeof(float));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(pcm.data.data()), static_cast<std::streamsize>(pcm.data.size() * sizeof(float)));
    jstring t = Java_io_github_jerinphilip_whisper_EngineNative_transcribeBuffer(&env, nullptr, h, &pcm);
    std::printf("buffer: %s\n", t->utf8.c_str());
    delete t;
    if (argc > 6) {
      _jstring wav;
      wav.utf8 = argv[6];
      jstring u = Java_io_github_jerinphilip_whisper_EngineNative_transcribeFile(&env, nullptr, h, &wav);
      std::printf("file: %s\n", u->utf8.c_str());
      delete u;
    }
  }
  Java_io_github_jerinphilip_whisper_EngineNative_destroy(&env, nullptr, h);  // 0 is a no-op, like `delete nullptr`
  std::printf("leaked_utf_chars %d\n", env.live_utf_chars);
  return env.live_utf_chars == 0 ? 0 : 1;
}
