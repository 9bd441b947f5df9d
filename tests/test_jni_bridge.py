"""The JNI bridge (bindings/java/whisper_tflite_jni.cpp; replaces the reference's bindings/java/whisper.tflite.cpp:17-71)
compiled and RUN against tests/jni_stub/jni.h — a stand-in for the JDK header that backs the few JNIEnv members the
bridge uses with plain C++ objects (the image has no JDK).  tests/jni_stub/driver.cpp plays EngineNative.java's part:
create -> transcribeBuffer / transcribeFile -> destroy (reference Driver.java:4-27).  What this pins: the bridge
compiles, links against libwhisper-tflite.so, maps the four natives onto the C ABI with the reference's conventions
(0 handle when creation fails, "" on a failed transcribe, destroy(0) harmless).  A real JVM is NOT exercised."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, synth_pcm

LIBDIR = os.path.join(ROOT, "whisper.tflite_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "jni_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings", "java", "whisper_tflite_jni.cpp"),
           os.path.join(ROOT, "tests", "jni_stub", "driver.cpp"), "-o", exe, "-L" + LIBDIR, "-lwhisper-tflite",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def test_jni_bridge_compiles_links_and_fails_softly(pkg, tmp_path):
    """CPU: a create that cannot succeed (missing files; and no GPU here) returns handle 0 with the reason on stderr,
    destroy(0) is a no-op, no UTF string is left unreleased; an unknown engine type is refused the same way."""
    exe = _build(tmp_path)
    r = subprocess.run([exe, "1", str(tmp_path / "nope"), str(tmp_path / "nope.bin"), "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["handle 0", "leaked_utf_chars 0"]
    assert "EngineNative.create:" in r.stderr
    r = subprocess.run([exe, "7", str(tmp_path / "nope"), str(tmp_path / "nope.bin"), "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "handle 0" in r.stdout and "Unknown engine-type" in r.stderr


@pytest.mark.gpu
def test_jni_bridge_transcribes_like_the_engine(pkg, assets, tmp_path):
    """GPU: the four natives end to end for both engine types — transcribeBuffer(float[]) and transcribeFile(String)
    return exactly what the engine's own entry points return for the same input."""
    exe = _build(tmp_path)
    prefix, vocab = assets("tiny")
    pcm = synth_pcm("speechlike", 160000, 77)
    (tmp_path / "clip.f32").write_bytes(np.ascontiguousarray(pcm, np.float32).tobytes())
    pcm16 = np.clip(np.round(pcm * 32767), -32768, 32767).astype("<i2")
    wav = tmp_path / "clip.wav"
    wav.write_bytes(b"RIFF" + struct.pack("<I", 36 + pcm16.nbytes) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", pcm16.nbytes) +
                    pcm16.tobytes())
    for etype in (1, 0):  # EngineType::EncDec, EngineType::Monolith (whisper.h:199-204)
        e = pkg.create_engine(pkg.EngineType.EncDec if etype == 1 else pkg.EngineType.Monolith, prefix, vocab, True)
        want_buf, want_file = e.transcribe(pcm), e.transcribe(str(wav))
        e.close()
        r = subprocess.run([exe, str(etype), prefix, vocab, "1", str(tmp_path / "clip.f32"), str(wav)],
                           capture_output=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.decode("utf-8", errors="replace").splitlines()
        assert lines[0] == "handle 1" and lines[-1] == "leaked_utf_chars 0"
        assert lines[1] == "buffer: " + want_buf and lines[2] == "file: " + want_file
