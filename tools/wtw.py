"""numpy reader for .wtw weight files (layout: whisper.tflite_amd/csrc/wtw_format.h)."""
import struct

import numpy as np

DIM_KEYS = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
            "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")


def read_wtw(path):
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    magic, version, n_tensors, table_off = struct.unpack_from("<IIII", buf, 0)
    assert magic == 0x31575457 and version == 1, "not a .wtw file"
    dims = dict(zip(DIM_KEYS, struct.unpack_from("<10i", buf, 16)))
    tensors = {}
    for i in range(n_tensors):
        e = table_off + i * 128
        name = bytes(buf[e:e + 80]).split(b"\0", 1)[0].decode()
        dtype, ndim = struct.unpack_from("<II", buf, e + 80)
        shape = struct.unpack_from("<4I", buf, e + 88)[:ndim]
        off, nbytes = struct.unpack_from("<QQ", buf, e + 104)
        assert dtype == 0
        tensors[name] = np.frombuffer(buf, dtype=np.float32, count=nbytes // 4, offset=off).reshape(shape)
    return dims, tensors
