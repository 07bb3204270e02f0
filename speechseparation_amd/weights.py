"""Deterministic synthetic weights / inputs and the flat weight-file format.

The reference ships no trained weights (`.MISSING_LARGE_BLOBS`), so every parity test,
the golden fixtures and `bench.py` use weights produced by a counter-based integer hash
keyed by the state_dict key name: bit-identical in this container, in the reference
import that generates the golden vectors, and on the GPU box.  Only integer arithmetic
and exactly-representable float conversions are used, so no libm call can change a bit.

Flat weight file (`*.bsrnnw`, consumed by the C library and the LADSPA plugin; replaces
the ONNX file of speech-ladspa-onnx.cpp:73 / infer-streaming.py:74):
    char[8]  magic "BSRNNW01"
    u32      n_bands, then n_bands x u32 band widths (bins)
    u32      n_tensors
    per tensor: u32 key_len, key bytes, u32 ndim, ndim x u64 dims, f32 data (little endian)
"""
import struct
import zlib
from collections import OrderedDict

import numpy as np

from . import spec as _spec

MAGIC = b"BSRNNW01"
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(stream_id, n, lane=0):
    """n exactly-representable uniforms in [0,1) with 24 random bits each (float64)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = (np.uint64(stream_id) * np.uint64(0xD1342543DE82EF95)
                + np.uint64(lane) * np.uint64(0xA0761D6478BD642F)) & _M64
        h = _splitmix64(_splitmix64(idx ^ base) + base)
    return (h >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def key_stream_id(key, seed=0):
    return (zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF) | (int(seed) << 32)


def synth_state_dict(v=None, seed=0, lstm_gain=1.0, const_gain=1.0):
    """key -> float32 ndarray for `spec.param_spec(v)`.

    U(-b, +b) with b = 1/sqrt(fan_in) (PyTorch's default scale for Linear/LSTM), so that
    activations stay O(0.1..1) through the stack.  `lstm_gain` > 1 multiplies the LSTM
    matrices ("hot" set) so gates saturate.  The zero-width band's
    learned constant (zeros at init in the reference, bsrnn.py:16) gets U(-1,1)*const_gain
    so that the path is exercised.
    """
    out = OrderedDict()
    H = _spec.BAND_FEATURES
    for key, shape in _spec.param_spec(v).items():
        n = int(np.prod(shape)) if len(shape) else 1
        if n == 0:
            out[key] = np.zeros(shape, dtype=np.float32)
            continue
        u = _uniform01(key_stream_id(key, seed), n)
        if key.endswith("trainable_constant"):
            bound = const_gain
        elif ".rnn." in key:
            bound = lstm_gain / np.sqrt(H)
        elif key.endswith(".weight"):
            bound = 1.0 / np.sqrt(shape[1])
        else:  # Linear bias: fan_in of the matching weight
            wshape = _spec.param_spec(v)[key[:-len("bias")] + "weight"]
            bound = 1.0 / np.sqrt(wshape[1])
        out[key] = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)
    return out


def synth_waveform(rows, n_samples, seed=1234, scale=0.1, row_offset=0):
    """[rows, n_samples] float32, approx 0.1*N(0,1), independent per row.

    Irwin-Hall (sum of 12 uniforms - 6): exact in float64, no libm.  `row_offset` lets a
    rank generate exactly its shard of a larger batch (row r of the global batch is the
    same whatever the sharding).
    """
    out = np.empty((rows, n_samples), dtype=np.float32)
    for r in range(rows):
        acc = np.zeros(n_samples, dtype=np.float64)
        sid = (int(seed) << 32) | ((row_offset + r) & 0xFFFFFFFF)
        for lane in range(12):
            acc += _uniform01(sid, n_samples, lane=lane + 1)
        out[r] = ((acc - 6.0) * scale).astype(np.float32)
    return out


def synth_tensor(shape, seed, scale=1.0):
    """Generic deterministic ~N(0,1)*scale tensor (for Z / state test inputs)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for lane in range(12):
        acc += _uniform01((int(seed) << 32) | 0x5EED, n, lane=lane + 1)
    return ((acc - 6.0) * scale).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------- flat file
def save_flat(path, state_dict, v=None):
    if v is None:
        v = _spec.generate_bandsplits()[0]
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(v)))
        f.write(struct.pack("<%dI" % len(v), *v))
        f.write(struct.pack("<I", len(state_dict)))
        for key, val in state_dict.items():
            arr = np.ascontiguousarray(np.asarray(val, dtype=np.float32))
            kb = key.encode("utf-8")
            f.write(struct.pack("<I", len(kb)))
            f.write(kb)
            f.write(struct.pack("<I", arr.ndim))
            f.write(struct.pack("<%dQ" % arr.ndim, *arr.shape))
            f.write(arr.astype("<f4").tobytes())


def load_flat(path):
    """-> (band widths, OrderedDict key -> float32 ndarray)"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != MAGIC:
        raise ValueError("%s: not a BSRNNW01 weight file" % path)
    pos = 8
    (nb,) = struct.unpack_from("<I", data, pos); pos += 4
    v = list(struct.unpack_from("<%dI" % nb, data, pos)); pos += 4 * nb
    (nt,) = struct.unpack_from("<I", data, pos); pos += 4
    sd = OrderedDict()
    for _ in range(nt):
        (kl,) = struct.unpack_from("<I", data, pos); pos += 4
        key = data[pos:pos + kl].decode("utf-8"); pos += kl
        (nd,) = struct.unpack_from("<I", data, pos); pos += 4
        dims = struct.unpack_from("<%dQ" % nd, data, pos); pos += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        sd[key] = np.frombuffer(data, dtype="<f4", count=n, offset=pos).reshape(dims).copy()
        pos += 4 * n
    return v, sd
