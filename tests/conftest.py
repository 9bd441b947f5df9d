import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import __graft_entry__ as ge  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    if not os.path.exists(os.path.join(ROOT, "whisper.tflite_amd", "lib", "libwhisper-tflite.so")):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    if not os.path.exists(os.path.join(ROOT, "oracle", "libwt_oracle.so")):
        ge.build()
    return ge.load_oracle()


@pytest.fixture(scope="session")
def assets(pkg, tmp_path_factory):
    """Synthetic weight/vocab files, generated once per session: name -> (prefix, vocab)."""
    d = tmp_path_factory.mktemp("assets")
    made = {}

    def get(arch, seed=0):
        key = (arch, seed)
        if key not in made:
            made[key] = ge._assets(str(d), arch, seed)
        return made[key]

    return get


def synth_pcm(kind, n, seed):
    """Same generator as tools/gen_golden.py (the golden files record kind/n/seed)."""
    from gen_golden import synth_pcm as f
    return f(kind, n, seed)


def argmax_last(x):
    x = np.asarray(x)
    return int(len(x) - 1 - np.argmax(x[::-1]))


class DevBuf:
    """Caller-owned device buffer for the *_dev entry points (what bench.py gets from torch).  Plain HIP through
    ctypes: torch's bundled HIP runtime cannot be initialised in a process where the engine's libamdhip64 is
    already live, so GPU tests never touch torch.cuda."""

    _hip = None

    def __init__(self, a):
        import ctypes
        if DevBuf._hip is None:
            DevBuf._hip = ctypes.CDLL("libamdhip64.so")
        a = np.ascontiguousarray(a)
        self.p = ctypes.c_void_p()
        assert DevBuf._hip.hipMalloc(ctypes.byref(self.p), ctypes.c_size_t(max(a.nbytes, 4))) == 0
        assert DevBuf._hip.hipMemcpy(self.p, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), 1) == 0

    def data_ptr(self):
        return self.p.value

    def free(self):
        if self.p:
            DevBuf._hip.hipFree(self.p)
            self.p = None
