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


def make_pair(pkg, tmp_path, arch="micro", named=True, seed=5, fuse_bias=False, fuse_ln=False, shuffle_qkv=False, tag=""):
    """Writes <prefix>.encoder.tflite / .decoder.tflite from synthetic weights the way the reference's converter
    stores them, and returns (prefix, dims, expected) with `expected` the de-quantised tensors in torch layout.
    fuse_bias: a Linear's weight and bias are inputs 1 and 2 of ONE FULLY_CONNECTED operator; fuse_ln: a LayerNorm's
    gain and shift are read by one operator; shuffle_qkv: the value projection comes first in the file, then key,
    then query (a converter is free to order them so)."""
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
    if fuse_bias or fuse_ln or shuffle_qkv:
        for gname, consts in graphs.items():
            names = [n for n in t if n.split(".", 1)[0] == gname]
            idx = {n: i for i, n in enumerate(names)}
            if shuffle_qkv:
                for n in names:
                    if n.endswith(".query.weight") and n.replace(".query.", ".value.") in idx:
                        i, j = idx[n], idx[n.replace(".query.", ".value.")]
                        consts[i], consts[j] = consts[j], consts[i]
                        names[i], names[j] = names[j], names[i]
                        idx = {m: k for k, m in enumerate(names)}
            out, skip = [], set()
            for i, n in enumerate(names):
                if i in skip:
                    continue
                partner = n[:-len("weight")] + "bias" if n.endswith(".weight") else None
                is_ln = n.endswith("_ln.weight") or n.endswith("ln_post.weight") or n.endswith("decoder.ln.weight")
                if partner in idx and ((fuse_ln and is_ln) or (fuse_bias and consts[i]["opcode"] == tw.OP_FULLY_CONNECTED)):
                    out.append([consts[i], consts[idx[partner]]])
                    skip.add(idx[partner])
                else:
                    out.append(consts[i])
            graphs[gname] = out
    prefix = str(tmp_path / f"model-{arch}-{'named' if named else 'anon'}{tag}")
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


def _same(got, expected):
    return all(got[k].shape == v.shape and np.array_equal(got[k].view(np.uint32), v.view(np.uint32)) for k, v in expected.items())


def test_extractor_operator_shapes_a_converter_may_emit(pkg, tmp_path):
    """(a) anonymous constants, a Linear's weight and bias carried by ONE FULLY_CONNECTED operator (inputs 1 and 2):
    rule 2 orders by operator, then by input position, and still maps every tensor;
    (b) named constants with q / k / v in another order: rule 1 does not depend on the order;
    (c) anonymous constants where one operator reads a LayerNorm's gain AND shift (equal size, no order to tell them
    apart): refused with WT_ERR_FORMAT naming the parameter and the operator, instead of guessed."""
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named=False, fuse_bias=True, tag="-fb")
    out = str(tmp_path / "fb.wtw")
    pkg.convert_tflite(prefix, out)
    assert _same(read_wtw(out)[1], expected)
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named=True, shuffle_qkv=True, fuse_bias=True, tag="-sh")
    out = str(tmp_path / "sh.wtw")
    pkg.convert_tflite(prefix, out)
    assert _same(read_wtw(out)[1], expected)
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named=False, fuse_ln=True, tag="-ln")
    with pytest.raises(pkg.WtError) as e:
        pkg.convert_tflite(prefix, str(tmp_path / "ln.wtw"))
    assert e.value.code == 3 and "cannot tell which constant" in str(e.value) and "attn_ln" in str(e.value)
    assert not os.path.exists(str(tmp_path / "ln.wtw"))  # nothing half-written is left behind
    # the same file WITH names converts
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named=True, fuse_ln=True, tag="-lnn")
    out = str(tmp_path / "lnn.wtw")
    pkg.convert_tflite(prefix, out)
    assert _same(read_wtw(out)[1], expected)


def test_converted_file_appears_atomically_and_a_broken_one_is_rebuilt(pkg, tmp_path):
    """wt_engine_create on a prefix that only has the reference's .tflite pair converts it into <prefix>.wtw through a
    temporary file renamed into place (ranks that start together never see a short file); a .wtw that is there but
    truncated is rebuilt from the pair rather than reported as 'not a .wtw' for ever.  No GPU here: creation itself
    ends with WT_ERR_DEVICE, AFTER the host-side conversion / validation has run."""
    prefix, dims, expected = make_pair(pkg, tmp_path, "micro", named=True, tag="-atomic")
    vocab = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(vocab, 1000)

    def create():
        try:
            pkg.Engine(prefix, vocab, True).close()
            return 0
        except pkg.WtError as e:
            return e.code

    have_gpu = create() == 0
    assert os.path.exists(prefix + ".wtw") and _same(read_wtw(prefix + ".wtw")[1], expected)
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    good = open(prefix + ".wtw", "rb").read()
    for blob in (good[: len(good) // 2], good[:100], b"garbage" * 100):
        open(prefix + ".wtw", "wb").write(blob)
        rc = create()
        assert rc == (0 if have_gpu else 5)
        assert open(prefix + ".wtw", "rb").read() == good
    # without the pair a truncated .wtw is an error that says so (WT_ERR_FORMAT), and the file is left alone
    lone = str(tmp_path / "lone")
    open(lone + ".wtw", "wb").write(good[: len(good) // 2])
    try:
        pkg.Engine(lone, vocab, True).close()
        raise AssertionError("a truncated weight file was accepted")
    except pkg.WtError as e:
        assert e.code in (3, 5)
    assert len(open(lone + ".wtw", "rb").read()) == len(good) // 2


def test_read_only_model_directory_converts_once_into_a_private_cache(pkg, tmp_path, monkeypatch):
    """A model directory that cannot be written: the converted weights go to a per-user cache directory
    ($XDG_CACHE_HOME/whisper-tflite-amd, created 0700), the second engine start REUSES the cached file instead of
    converting again, a symlink planted under the temporary file's name is not followed, and a cache directory other users
    could write is refused.  (No GPU needed: creation ends with WT_ERR_DEVICE after the host-side work.)"""
    # (as root directory permissions do not bind: the test then only checks that the ordinary path still works)
    model_dir = tmp_path / "ro-model"
    model_dir.mkdir()
    prefix, dims, expected = make_pair(pkg, model_dir, "micro", named=True, tag="-ro")
    vocab = str(tmp_path / "v.bin")
    pkg.write_synthetic_vocab(vocab, 1000)
    cache = tmp_path / "xdg"
    cache.mkdir()
    monkeypatch.setenv("XDG_CACHE_HOME", str(cache))

    def create():
        try:
            pkg.Engine(prefix, vocab, True).close()
            return 0
        except pkg.WtError as e:
            return e.code

    os.chmod(model_dir, 0o555)
    try:
        writable = os.access(str(model_dir), os.W_OK)  # root: permissions do not bind
        rc = create()
        assert rc in (0, 5)
        if writable:
            assert os.path.exists(prefix + ".wtw")
            return
        assert not os.path.exists(prefix + ".wtw")
        cdir = cache / "whisper-tflite-amd"
        files = [f for f in os.listdir(cdir) if f.endswith(".wtw")]
        assert len(files) == 1 and (os.stat(cdir).st_mode & 0o777) == 0o700
        cached = cdir / files[0]
        assert _same(read_wtw(str(cached))[1], expected)
        stamp = os.stat(cached).st_mtime_ns
        assert create() == rc and os.stat(cached).st_mtime_ns == stamp  # reused, not converted again
        # a broken cached file is replaced; a symlink planted as the temporary's name is not followed
        open(cached, "wb").write(b"junk")
        victim = tmp_path / "victim.txt"
        victim.write_text("mine")
        os.symlink(str(victim), str(cached) + ".tmp." + str(os.getpid()))  # the writer's temporary name, pre-planted
        assert create() == rc and _same(read_wtw(str(cached))[1], expected) and victim.read_text() == "mine"
        assert not [f for f in os.listdir(cdir) if ".tmp." in f]
        # a cache directory that others may write is not trusted
        os.chmod(cdir, 0o777)
        os.remove(cached)
        assert create() == 2  # WT_ERR_IO
    finally:
        os.chmod(model_dir, 0o755)
        if (cache / "whisper-tflite-amd").exists():
            os.chmod(cache / "whisper-tflite-amd", 0o700)
