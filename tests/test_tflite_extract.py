"""CPU: the .tflite -> .wtw weight extractor (reference whisper.cpp:743-744 opens
<prefix>.encoder.tflite / <prefix>.decoder.tflite, export/generate_onnx.py:135-163 writes them with dynamic-range
quantisation).  PARITY UNPINNED — no .tflite file and no TFLite / FlatBuffers library exist here: the fixture files
come from tests/tflite_writer.py (same schema field numbers, same int8 formulas), so these tests pin the container
walk, the de-quantisation arithmetic, the layout conversions and both mapping rules, not agreement with TensorFlow."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tflite_writer as tw  # noqa: E402
from wtw import read_wtw  # noqa: E402


def make_pair(pkg, tmp_path, arch="micro", named=True, seed=5):
    """Writes <prefix>.encoder.tflite / .decoder.tflite from synthetic weights the way the reference's converter
    stores them, and returns (prefix, dims, expected) with `expected` the de-quantised tensors in torch layout."""
    src = str(tmp_path / f"src-{arch}.wtw")
    pkg.write_synthetic_weights(src, arch, seed)
    dims, t = read_wtw(src)
    expected, graphs = {}, {"encoder": [], "decoder": []}
    counter = [0]

    def add(graph, name, arr, kind):
        a = np.array(arr, np.float32)
        c = {"name": (name.split(".", 1)[1] + ":0") if named else f"const_fold_opt__{counter[0]}"}
        counter[0] += 1
        if kind == "f32":
            c.update(data=a, opcode=tw.OP_ADD if name.endswith("bias") else tw.OP_MUL)
            exp = a
        elif kind == "f16":
            h = a.astype(np.float16)
            c.update(data=h, opcode=tw.OP_ADD)
            exp = h.astype(np.float32)
        elif kind == "fc":  # FULLY_CONNECTED weights [out][in], int8 per output channel
            q, sc, exp = tw.quantize_int8(a, per_axis=0)
            c.update(data=q, scale=sc, qdim=0, opcode=tw.OP_FULLY_CONNECTED)
        elif kind == "matmul":  # MatMul right-hand side [in][out], int8 per tensor
            q, sc, deq = tw.quantize_int8(a.T.copy())
            c.update(data=q, scale=sc, opcode=tw.OP_BATCH_MATMUL)
            exp = deq.T.copy()
        elif kind == "conv":  # CONV_2D filter [out][1][k][in], int8 per output channel
            f = np.transpose(a, (0, 2, 1))[:, None, :, :].copy()
            q, sc, deq = tw.quantize_int8(f, per_axis=0)
            c.update(data=q, scale=sc, qdim=0, opcode=tw.OP_CONV_2D)
            exp = np.transpose(deq[:, 0], (0, 2, 1)).copy()
        elif kind == "gather":  # embedding table, int8 per row
            q, sc, exp = tw.quantize_int8(a, per_axis=0)
            c.update(data=q, scale=sc, qdim=0, opcode=tw.OP_GATHER)
        expected[name] = np.ascontiguousarray(exp, np.float32)
        graphs[graph].append(c)

    for name, arr in t.items():  # file order = forward order of the graphs (csrc/weights_gen.cpp build_specs)
        graph = name.split(".", 1)[0]
        if name.endswith("conv1.weight") or name.endswith("conv2.weight"):
            add(graph, name, arr, "conv")
        elif name.endswith("token_embedding.weight"):
            add(graph, name, arr, "gather")
        elif name.endswith("positional_embedding"):
            add(graph, name, arr, "f16" if graph == "decoder" else "f32")
        elif arr.ndim == 2:
            # alternate the two ways a Linear reaches the file, including for the square projections
            add(graph, name, arr, "matmul" if (".key." in name or ".mlp.0." in name or ".out." in name) else "fc")
        else:
            add(graph, name, arr, "f32")
    prefix = str(tmp_path / f"model-{arch}-{'named' if named else 'anon'}")
    tw.write_tflite(prefix + ".encoder.tflite", graphs["encoder"])
    tw.write_tflite(prefix + ".decoder.tflite", graphs["decoder"])
    return prefix, dims, expected


@pytest.mark.parametrize("named", [True, False])
def test_extractor_round_trip(pkg, tmp_path, named):
    """Every tensor of the converted file equals the de-quantised source bit for bit: int8 per-tensor and per-axis
    scales, float16 and float32 constants, CONV_2D filter layout, MatMul right-hand sides stored [in][out] (also
    square ones, told apart by the consuming operator), with tensor names that carry the parameter path and with
    anonymous names (mapping by first-use order and element count)."""
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named)
    out = str(tmp_path / "converted.wtw")
    pkg.convert_tflite(prefix, out)
    got_dims, got = read_wtw(out)
    assert got_dims == dims
    assert list(got) == list(expected)  # same tensors, same file order as the native writer
    for k, v in expected.items():
        assert got[k].shape == v.shape, k
        assert np.array_equal(got[k].view(np.uint32), v.view(np.uint32)), k


def test_extractor_recognises_the_architectures(pkg, tmp_path):
    """Dims follow from the constants' sizes: tiny.en (51864 tokens) is told from tiny (51865)."""
    for arch in ("tiny.en",):
        prefix, dims, expected = make_pair(pkg, tmp_path, arch, named=False, seed=1)
        out = str(tmp_path / f"{arch}.wtw")
        pkg.convert_tflite(prefix, out)
        got_dims, got = read_wtw(out)
        assert got_dims == dims and got_dims["n_vocab"] == 51864
        k = "decoder.blocks.3.cross_attn.out.weight"
        assert np.array_equal(got[k], expected[k])


def test_extractor_errors_are_status_codes(pkg, tmp_path):
    prefix, _, _ = make_pair(pkg, tmp_path, "micro", True)
    out = str(tmp_path / "o.wtw")
    with pytest.raises(pkg.WtError) as e:
        pkg.convert_tflite(str(tmp_path / "nope"), out)
    assert e.value.code == 2  # WT_ERR_IO
    raw = open(prefix + ".encoder.tflite", "rb").read()
    for name, blob in (("ident", raw[:4] + b"XXXX" + raw[8:]), ("trunc", raw[: len(raw) // 3]), ("tiny", raw[:10]),
                       ("rootoff", b"\xff\xff\xff\x7f" + raw[4:])):
        p = str(tmp_path / name)
        open(p + ".encoder.tflite", "wb").write(blob)
        open(p + ".decoder.tflite", "wb").write(open(prefix + ".decoder.tflite", "rb").read())
        with pytest.raises(pkg.WtError) as e:
            pkg.convert_tflite(p, out)
        assert e.value.code == 3, name  # WT_ERR_FORMAT
    # a graph that lacks a parameter: the error names it
    p = str(tmp_path / "swapped")
    open(p + ".encoder.tflite", "wb").write(open(prefix + ".decoder.tflite", "rb").read())
    open(p + ".decoder.tflite", "wb").write(open(prefix + ".decoder.tflite", "rb").read())
    with pytest.raises(pkg.WtError) as e:
        pkg.convert_tflite(p, out)
    assert e.value.code == 3
