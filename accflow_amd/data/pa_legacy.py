"""Decoder for pyarrow's LEGACY object serialisation (`pa.serialize(obj).to_buffer()` / `pa.deserialize(buf)`).

The CVO LMDB values were written with it and the reference reads them back with `pa.deserialize`
(data/dataset.py:45,64; pyarrow 3.0 / 11.0 in requirements.txt:96 / environment.yml:193).  The API was deprecated in
pyarrow 2.0 and is gone from the installed pyarrow 25, but the container format is ordinary Arrow IPC, which modern
pyarrow still reads.  Layout of a serialised object (arrow/python/serialize.cc, `SerializedPyObject::WriteTo`, the
form written by pyarrow 0.15 ... 14):

    int32 num_tensors, int32 num_sparse_tensors, int32 num_ndarrays, int32 num_buffers
    [pad to 8]   an IPC stream (schema, one record batch, end-of-stream) with ONE column: a dense union over the
                 python types present - child arrays int64 (int), bool, double / float, string, binary, list<union>
                 (list / tuple / set), struct (dict: keys, vals), int32 (an index into the tensors / ndarrays /
                 buffers that follow) - wrapping the object as a one-element sequence
    [pad to 64]  num_tensors, then num_ndarrays IPC Tensor messages, each followed by padding to 64
                 num_buffers x {int64 size, bytes}

A numpy array - what every CVO data value is, (H, W, C) uint8 frames or uint16-coded flows - is therefore one int32
union entry (value 0) plus one IPC tensor.  Union TYPE CODES are the C++ enum `PythonType` (INT = 2 ... LIST = 10,
DICT = 11, TUPLE = 12, SET = 13, TENSOR = 14, NDARRAY = 15, BUFFER = 16); this decoder dispatches on the Arrow TYPE of
each child and uses the codes only to tell tuple / set from list and tensor / ndarray / buffer indices apart, so a
renumbering between pyarrow versions cannot mis-decode numbers, strings, lists or a lone array.

Parity note: no file written by a real legacy pyarrow exists offline; the decoder is exercised on streams produced by
tests/golden/make_cvo_fixture.py, which writes this layout with the installed pyarrow's IPC writer.
"""
import struct

import numpy as np

PT_LIST, PT_DICT, PT_TUPLE, PT_SET, PT_TENSOR, PT_NDARRAY, PT_BUFFER = 10, 11, 12, 13, 14, 15, 16


def _align(pos, a):
    return (pos + a - 1) // a * a


def deserialize(buf):
    """bytes-like -> python object (numbers, str, bytes, list / tuple / set / dict of those, numpy arrays)."""
    import pyarrow as pa
    buf = memoryview(buf)
    if len(buf) < 16:
        raise ValueError("not a legacy pyarrow serialised object (shorter than its 16-byte header)")
    n_tensors, n_sparse, n_ndarrays, n_buffers = struct.unpack_from("<iiii", buf, 0)
    if min(n_tensors, n_sparse, n_ndarrays, n_buffers) < 0 or n_sparse:
        raise ValueError("unsupported legacy pyarrow object (counts %s)" % ((n_tensors, n_sparse, n_ndarrays, n_buffers),))
    src = pa.BufferReader(pa.py_buffer(buf))
    src.seek(16)                                    # already 8-byte aligned
    rd = pa.ipc.open_stream(src)
    batch = rd.read_next_batch()
    try:
        rd.read_next_batch()                        # consume the end-of-stream marker
    except StopIteration:
        pass
    arrays = []
    for _ in range(n_tensors + n_ndarrays):
        arrays.append(_read_tensor(src))
    pos = _align(src.tell(), 64) if arrays else src.tell()
    buffers = []
    for _ in range(n_buffers):
        size = struct.unpack_from("<q", buf, pos)[0]
        buffers.append(bytes(buf[pos + 8:pos + 8 + size]))
        pos += 8 + size
    blobs = {"tensors": arrays[:n_tensors], "ndarrays": arrays[n_tensors:], "buffers": buffers}
    seq = _decode_union(batch.column(0), blobs)
    if len(seq) != 1:
        raise ValueError("legacy pyarrow object: expected a one-element top-level sequence, got %d" % len(seq))
    return seq[0]


def _read_tensor(src):
    """The next IPC Tensor message.  The writer pads to 64 bytes before each tensor, measured from the end of the IPC
    stream - whose end-of-stream marker is 4 bytes (0x00000000) in the streams of pyarrow < 0.15 and 8 bytes
    (0xFFFFFFFF 0x00000000) since; depending on the reader the position after the last batch may sit before or after it.
    So: the 64-byte boundary first, then every 8-byte boundary up to 128 bytes further (a tensor message starts with its
    own continuation marker / length prefix and fails fast anywhere else)."""
    import pyarrow as pa
    pos = src.tell()
    first = _align(pos, 64)
    cands = [first] + [p for p in range(_align(pos, 8), first + 129, 8) if p != first]
    err = None
    for p in cands:
        try:
            src.seek(p)
            return pa.ipc.read_tensor(src).to_numpy()
        except Exception as e:   # pyarrow raises ArrowInvalid / OSError / ArrowTypeError depending on what it hit
            err = err or e
    raise ValueError("legacy pyarrow object: no tensor message at or after byte %d (%s)" % (pos, err))


def _decode_union(arr, blobs):
    """dense union array -> list of python values"""
    import pyarrow as pa
    if not pa.types.is_union(arr.type):
        raise ValueError("legacy pyarrow object: expected a union column, got %s" % arr.type)
    codes = arr.type.type_codes
    children = {code: arr.field(i) for i, code in enumerate(codes)}
    tids = arr.type_codes.to_pylist()
    offs = arr.offsets.to_pylist()
    return [_decode_value(children[t], o, t, blobs) for t, o in zip(tids, offs)]


def _decode_value(child, i, code, blobs):
    import pyarrow as pa
    t = child.type
    if pa.types.is_null(t):
        return None
    if pa.types.is_list(t):
        # (the whole values union is decoded, then cut: slicing a dense union does not re-base its type ids / offsets)
        start, stop = child.offsets[i].as_py(), child.offsets[i + 1].as_py()
        vals = _decode_union(child.values, blobs)[start:stop]
        return tuple(vals) if code == PT_TUPLE else set(vals) if code == PT_SET else vals
    if pa.types.is_struct(t):                      # dict: struct {keys: list<union>, vals: list<union>}
        keys = _decode_value(child.field(0), i, PT_LIST, blobs)
        vals = _decode_value(child.field(1), i, PT_LIST, blobs)
        return dict(zip(keys, vals))
    if pa.types.is_int32(t):                       # index into tensors / ndarrays / buffers
        idx = child[i].as_py()
        if code == PT_BUFFER and blobs["buffers"]:
            return blobs["buffers"][idx]
        pool = blobs["ndarrays"] if (code == PT_NDARRAY or not blobs["tensors"]) else blobs["tensors"]
        if not pool:
            pool = blobs["tensors"] or blobs["buffers"]
        return pool[idx]
    v = child[i].as_py()
    if isinstance(v, float) and pa.types.is_float32(t):
        return float(np.float32(v))
    return v
