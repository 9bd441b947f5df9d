"""Minimal TFLite FlatBuffer WRITER — test fixture generator for the .tflite -> .wtw extractor
(whisper.tflite_amd/csrc/tflite_extract.cpp).

There is no .tflite model, no TensorFlow and no FlatBuffers library in this environment, so the extractor's reader
is exercised on files made here: the same container rules (root offset + "TFL3" identifier, tables with vtables,
vectors, strings) and the field numbers of tensorflow/lite/schema/schema.fbs (v3) that the reader documents, with the
quantisation formulas of TFLite's dynamic-range converter (symmetric int8, per tensor or per output channel).
PARITY UNPINNED: a file written by the real converter has never been read by this code.

Only what a weight extractor looks at is written: operator codes, one subgraph with tensors (shape, type, buffer,
name, quantization), operators (opcode index, inputs, outputs) in forward order, and buffers.
"""
import struct

import numpy as np

FLOAT32, FLOAT16, INT32, INT8 = 0, 1, 2, 9
OP_CONV_2D, OP_FULLY_CONNECTED, OP_GATHER, OP_MUL, OP_ADD, OP_BATCH_MATMUL = 3, 9, 36, 18, 0, 126


class Table:
    """fields: {slot: (kind, value)} with kind in u8 / i8 / u32 / i32 / u64 / table / vec_table / vec (fmt, list) /
    bytes / string"""

    def __init__(self, **fields):
        self.fields = {int(k[1:]): v for k, v in fields.items()}  # f0=..., f1=...


_SCALARS = {"u8": ("<B", 1), "i8": ("<b", 1), "u32": ("<I", 4), "i32": ("<i", 4), "u64": ("<Q", 8)}


class _Builder:
    def __init__(self):
        self.buf = bytearray()

    def align(self, n):
        while len(self.buf) % n:
            self.buf.append(0)

    def patch_offset(self, field_pos, target_pos):
        struct.pack_into("<I", self.buf, field_pos, target_pos - field_pos)

    def write_table(self, t):
        n_slots = max(t.fields) + 1 if t.fields else 0
        # inline layout: soffset (4 bytes) then fields, widest first
        order = sorted(t.fields, key=lambda s: -(_SCALARS[t.fields[s][0]][1] if t.fields[s][0] in _SCALARS else 4))
        offs, pos = {}, 4
        for s in order:
            size = _SCALARS[t.fields[s][0]][1] if t.fields[s][0] in _SCALARS else 4
            pos = (pos + size - 1) // size * size
            offs[s] = pos
            pos += size
        table_size = pos
        vt_size = 4 + 2 * n_slots
        self.align(2)
        # place the vtable so that the table behind it starts 8-byte aligned
        while (len(self.buf) + vt_size) % 8:
            self.buf.append(0)
        vt_pos = len(self.buf)
        self.buf += struct.pack("<HH", vt_size, table_size)
        for s in range(n_slots):
            self.buf += struct.pack("<H", offs.get(s, 0))
        tpos = len(self.buf)
        self.buf += bytes(table_size)
        struct.pack_into("<i", self.buf, tpos, tpos - vt_pos)
        pending = []
        for s, (kind, val) in t.fields.items():
            fp = tpos + offs[s]
            if kind in _SCALARS:
                struct.pack_into(_SCALARS[kind][0], self.buf, fp, val)
            else:
                pending.append((fp, kind, val))
        for fp, kind, val in pending:
            if kind == "table":
                self.patch_offset(fp, self.write_table(val))
            elif kind == "string":
                self.align(4)
                p = len(self.buf)
                b = val.encode()
                self.buf += struct.pack("<I", len(b)) + b + b"\0"
                self.patch_offset(fp, p)
            elif kind == "bytes":
                self.align(4)
                while (len(self.buf) + 4) % 16:  # payload 16-byte aligned like the converter's force_align
                    self.buf.append(0)
                p = len(self.buf)
                self.buf += struct.pack("<I", len(val)) + bytes(val)
                self.patch_offset(fp, p)
            elif kind == "vec":
                fmt, items = val
                size = struct.calcsize("<" + fmt)
                self.align(4)
                while (len(self.buf) + 4) % size:
                    self.buf.append(0)
                p = len(self.buf)
                self.buf += struct.pack("<I", len(items))
                for it in items:
                    self.buf += struct.pack("<" + fmt, it)
                self.patch_offset(fp, p)
            elif kind == "vec_table":
                self.align(4)
                p = len(self.buf)
                self.buf += struct.pack("<I", len(val)) + bytes(4 * len(val))
                self.patch_offset(fp, p)
                for i, child in enumerate(val):
                    self.patch_offset(p + 4 + 4 * i, self.write_table(child))
            else:
                raise ValueError(kind)
        return tpos


def quantize_int8(w, per_axis=None):
    """TFLite dynamic-range weights: symmetric int8, scale = max|w| / 127 per tensor or per slice of `per_axis`.
    Returns (q int8, scales float32 [n], dequantised float32) with dequant = scale * q exactly as the extractor."""
    w = np.asarray(w, np.float32)
    if per_axis is None:
        scale = np.array([max(float(np.abs(w).max()), 1e-30) / 127.0], np.float32)
        q = np.clip(np.round(w / scale[0]), -127, 127).astype(np.int8)
        return q, scale, (scale[0] * q.astype(np.float32)).astype(np.float32)
    moved = np.moveaxis(w, per_axis, 0)
    scale = (np.maximum(np.abs(moved).reshape(moved.shape[0], -1).max(1), 1e-30) / 127.0).astype(np.float32)
    shp = [1] * w.ndim
    shp[per_axis] = -1
    q = np.clip(np.round(w / scale.reshape(shp)), -127, 127).astype(np.int8)
    return q, scale, (scale.reshape(shp) * q.astype(np.float32)).astype(np.float32)


def write_tflite(path, constants):
    """constants: list in FORWARD (first-use) order; an entry is a dict
         name, data (np array: float32 / float16 / int8), opcode (builtin operator that consumes it),
         scale (float32 array, int8 only), qdim (int8 per-axis only)
    or a LIST of such dicts that ONE operator reads together (a FULLY_CONNECTED with its weight at input 1 and its bias
    at input 2; a folded LayerNorm with gain and shift): the operator takes the opcode of the first.
    Every constant becomes one tensor + one buffer; an operator's inputs are the running activation and its
    constants, so the operator order is the order of the list."""
    groups = [c if isinstance(c, list) else [c] for c in constants]
    opcodes = sorted({g[0]["opcode"] for g in groups})
    op_tables = [Table(f0=("i8", min(o, 127)), f2=("i32", 1), f3=("i32", o)) for o in opcodes]
    buffers = [Table()]  # buffer 0: the empty sentinel
    tensors = [Table(f0=("vec", ("i", [1, 8])), f1=("u8", FLOAT32), f2=("u32", 0), f3=("string", "activation"))]
    operators = []
    for g in groups:
        ins = [0]
        for c in g:
            a = np.ascontiguousarray(c["data"])
            ttype = {np.dtype(np.float32): FLOAT32, np.dtype(np.float16): FLOAT16, np.dtype(np.int8): INT8}[a.dtype]
            buffers.append(Table(f0=("bytes", a.tobytes())))
            fields = dict(f0=("vec", ("i", list(a.shape))), f1=("u8", ttype), f2=("u32", len(buffers) - 1),
                          f3=("string", c["name"]))
            if ttype == INT8:
                sc = np.asarray(c["scale"], np.float32).reshape(-1)
                fields["f4"] = ("table", Table(f2=("vec", ("f", [float(x) for x in sc])), f3=("vec", ("q", [0] * len(sc))),
                                               f6=("i32", int(c.get("qdim", 0)))))
            tensors.append(Table(**fields))
            ins.append(len(tensors) - 1)
        operators.append(Table(f0=("u32", opcodes.index(g[0]["opcode"])), f1=("vec", ("i", ins)), f2=("vec", ("i", [0]))))
    sub = Table(f0=("vec_table", tensors), f1=("vec", ("i", [0])), f2=("vec", ("i", [0])), f3=("vec_table", operators),
                f4=("string", "main"))
    model = Table(f0=("u32", 3), f1=("vec_table", op_tables), f2=("vec_table", [sub]), f3=("string", "wt test fixture"),
                  f4=("vec_table", buffers))
    b = _Builder()
    b.buf += bytes(4) + b"TFL3"
    root = b.write_table(model)
    struct.pack_into("<I", b.buf, 0, root)
    with open(path, "wb") as f:
        f.write(b.buf)
