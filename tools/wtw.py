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


def write_wtw(path, dims, tensors, seed=0):
    """Writes a .wtw file from a dict name -> float32 array (same layout as csrc/weights_gen.cpp emits):
    128-byte header, 128-byte table entries, every payload 256-byte aligned."""
    names = list(tensors)
    table_off = 128
    payload_off = table_off + 128 * len(names)
    payload_off = (payload_off + 255) // 256 * 256
    offs, pos = [], payload_off
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        offs.append(pos)
        pos = (pos + a.nbytes + 255) // 256 * 256
    file_bytes = pos
    buf = bytearray(file_bytes)
    struct.pack_into("<IIII", buf, 0, 0x31575457, 1, len(names), table_off)
    struct.pack_into("<10i", buf, 16, *[int(dims[k]) for k in DIM_KEYS])
    struct.pack_into("<QQQ", buf, 56, payload_off, file_bytes, seed)
    for i, n in enumerate(names):
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        e = table_off + i * 128
        nb = n.encode()
        assert len(nb) < 80
        buf[e:e + len(nb)] = nb
        shape = list(a.shape) + [0] * (4 - a.ndim)
        struct.pack_into("<II4I", buf, e + 80, 0, a.ndim, *shape)
        struct.pack_into("<QQ", buf, e + 104, offs[i], a.nbytes)
        buf[offs[i]:offs[i] + a.nbytes] = a.tobytes()
    with open(path, "wb") as f:
        f.write(buf)


def adversarial_weights(src_wtw, dst_wtw, ln_gain=30.0, heavy=True, v_row_scale=1.0, seed=1234):
    """Random-init weights with the statistics a trained checkpoint can have and N(0, 1/fan_in) does not: LayerNorm
    gains `ln_gain` x larger on six channels of every encoder LayerNorm, LayerNorm shifts on others, heavy-tailed rows
    (2 % of the entries of every encoder Linear 8..40 x larger), and optionally one output channel of layer 0's value
    projection `v_row_scale` x larger with the matching out-projection column that much smaller (the rescaling
    symmetry a trained network is free to use).  Used by tests/test_gpu_boundary.py and bench.py's outlier_weights leg."""
    dims, t = read_wtw(src_wtw)
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in t.items():
        a = np.array(v, dtype=np.float32)
        if (k.startswith("encoder.") and k.endswith("_ln.weight")) or k == "encoder.ln_post.weight":
            a[rng.choice(a.size, 6, replace=False)] *= ln_gain
        elif k.startswith("encoder.") and (k.endswith("_ln.bias") or k == "encoder.ln_post.bias"):
            a[rng.choice(a.size, 6, replace=False)] += rng.uniform(-3, 3, 6).astype(np.float32)
        elif heavy and k.startswith("encoder.blocks.") and k.endswith(".weight") and a.ndim == 2:
            mask = rng.random(a.shape) < 0.02
            a[mask] *= rng.uniform(8, 40, int(mask.sum())).astype(np.float32)
        out[k] = a
    if v_row_scale != 1.0:
        out["encoder.blocks.0.attn.value.weight"][77] *= v_row_scale
        out["encoder.blocks.0.attn.value.bias"][77] *= v_row_scale
        out["encoder.blocks.0.attn.out.weight"][:, 77] /= v_row_scale
    write_wtw(dst_wtw, dims, out)


def trained_like_weights(src_wtw, dst_wtw, seed=4321, massive=(23, 187), massive_gain=40.0, ln_outliers=0.02,
                         ln_outlier_range=(6.0, 20.0), row_sigma=0.5):
    """Random-init weights re-shaped towards the statistics published for trained transformer checkpoints, in EVERY
    encoder layer (adversarial_weights above perturbs single places):
      * LayerNorm gains log-normal (sigma 0.35) with `ln_outliers` of the channels `ln_outlier_range` (6..20) x larger,
        shifts N(0, 0.15^2);
      * every Linear / Conv weight with log-normal row norms (sigma `row_sigma` = 0.5) and heavy-tailed entries (a Student-t with 4
        degrees of freedom, rescaled to the row's former norm): a few entries per row 5..10 sigma out;
      * "massive activations": the residual-stream channels `massive` carry values `massive_gain` x the others in all
        layers — the rows of conv2 / attention-out / fc2 that write them and their biases are scaled up, so every
        LayerNorm sees them and every consumer's bound / typical ratio grows with them.
    The decoder keeps its weights (its kernels scale per row from the data).  Used by bench.py's trained_like_weights leg
    and tests/test_gpu_boundary.py."""
    dims, t = read_wtw(src_wtw)
    rng = np.random.default_rng(seed)
    d = int(dims["n_audio_state"])
    massive = [c % d for c in massive]
    out = {}
    for k, v in t.items():
        a = np.array(v, dtype=np.float32)
        enc = k.startswith("encoder.")
        if enc and (k.endswith("_ln.weight") or k == "encoder.ln_post.weight"):
            a *= np.exp(rng.normal(0.0, 0.35, a.size)).astype(np.float32)
            hot = rng.random(a.size) < ln_outliers
            a[hot] *= rng.uniform(ln_outlier_range[0], ln_outlier_range[1], int(hot.sum())).astype(np.float32)
        elif enc and (k.endswith("_ln.bias") or k == "encoder.ln_post.bias"):
            a += rng.normal(0.0, 0.15, a.size).astype(np.float32)
        elif enc and k.endswith(".weight") and a.ndim >= 2:
            rows = a.reshape(a.shape[0], -1)
            norm = np.linalg.norm(rows, axis=1, keepdims=True)
            heavy = rng.standard_t(4, size=rows.shape).astype(np.float32)
            heavy *= norm / np.maximum(np.linalg.norm(heavy, axis=1, keepdims=True), 1e-30)
            heavy *= np.exp(rng.normal(0.0, row_sigma, (rows.shape[0], 1))).astype(np.float32)
            a = heavy.reshape(a.shape)
        out[k] = a
    writers = ["encoder.conv2"] + [f"encoder.blocks.{l}.{m}" for l in range(int(dims["n_audio_layer"])) for m in ("attn.out", "mlp.2")]
    for w in writers:
        for c in massive:
            out[w + ".weight"][c] *= massive_gain
            out[w + ".bias"][c] = out[w + ".bias"][c] * massive_gain + (3.0 if w == "encoder.conv2" else 0.0)
    write_wtw(dst_wtw, dims, out)
