// TEST FIXTURE, not the JDK's header: the image has no JDK (no jni.h, no javac), so the JNI bridge
// (bindings/java/whisper_tflite_jni.cpp) is compiled and RUN against this stand-in — the handful of JNI types and
// JNIEnv members the bridge uses, backed by plain C++ objects instead of a JVM.  It proves that the bridge compiles,
// links against libwhisper-tflite.so and drives the C ABI correctly; it says nothing about a real JVM.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef float jfloat;
typedef jint jsize;

struct _jobject {
  virtual ~_jobject() {}
};
struct _jstring : _jobject {
  std::string utf8;
};
struct _jfloatArray : _jobject {
  std::vector<float> data;
};
typedef _jobject* jobject;
typedef _jstring* jstring;
typedef _jfloatArray* jfloatArray;

struct JNIEnv {
  int live_utf_chars = 0;  // GetStringUTFChars without Release (the driver checks it is 0 at the end)
  const char* GetStringUTFChars(jstring s, jboolean* is_copy) {
    if (is_copy) *is_copy = 0;
    ++live_utf_chars;
    return s->utf8.c_str();
  }
  void ReleaseStringUTFChars(jstring, const char*) { --live_utf_chars; }
  jstring NewStringUTF(const char* bytes) {
    jstring s = new _jstring;
    s->utf8 = bytes ? bytes : "";
    return s;
  }
  jsize GetArrayLength(jfloatArray a) { return static_cast<jsize>(a->data.size()); }
  void GetFloatArrayRegion(jfloatArray a, jsize start, jsize len, jfloat* buf) {
    std::memcpy(buf, a->data.data() + start, static_cast<size_t>(len) * sizeof(float));
  }
};
