"""CPU-side tests: the C-ABI library loads and exports exactly what include/bsrnn_hip.h
declares, the host mirror class has the reference's state_dict inventory, the flat weight
file round-trips, generators are deterministic, and the product path refuses to run without
a GPU instead of falling back to a CPU implementation.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import REPO
from speechseparation_amd import spec, weights

LIB = os.path.join(REPO, "speechseparation_amd", "lib", "libbsrnn_hip.so")


def header_symbols():
    txt = open(os.path.join(REPO, "include", "bsrnn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bsrnn_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return LIB


def test_header_symbols_are_exported(built):
    from speechseparation_amd import _native
    syms = header_symbols()
    assert syms == sorted(_native.SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (bsrnn_[a-z0-9_]+)", out))
    assert set(syms) <= exported
    lib = ctypes.CDLL(built)
    for s in syms:
        assert hasattr(lib, s)
    assert lib.bsrnn_abi_version() == 2


def test_ladspa_plugin_exports_descriptor(built):
    plug = os.path.join(os.path.dirname(built), "speech_separator_ladspa.so")
    assert os.path.exists(plug)
    out = subprocess.run(["nm", "-D", "--defined-only", plug], capture_output=True, text=True, check=True).stdout
    assert re.search(r" T ladspa_descriptor\b", out)


def test_host_class_state_dict_inventory(built):
    from speechseparation_amd.bsrnn import BSRNN
    m = BSRNN()
    sd = m.state_dict()
    ps = spec.param_spec()
    assert list(sd.keys()) == list(ps.keys())
    assert all(tuple(sd[k].shape) == tuple(ps[k]) for k in ps)
    assert sum(p.numel() for p in m.parameters()) == 7481062
    m41 = BSRNN(spec.variant_bandsplits("41"))
    assert list(m41.state_dict().keys()) == list(spec.param_spec(spec.variant_bandsplits("41")).keys())
    # strict load of a foreign dict with the reference's names works, wrong names fail
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.synth_state_dict().items()}, strict=True)
    with pytest.raises(RuntimeError):
        m.load_state_dict({"nope": torch.zeros(1)}, strict=True)


def test_no_cpu_fallback(built):
    from speechseparation_amd._native import NativeError
    from speechseparation_amd.bsrnn import BSRNN
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(NativeError):
        BSRNN()(torch.zeros((2, 2050, 3)))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "speechseparation_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"(^|\n)\s*(from|import)\s+oracle|oracle/|oracle\.", txt), os.path.join(root, f)
    for f in ("infer.py", "infer-streaming.py"):
        txt = open(os.path.join(REPO, f)).read()
        assert not re.search(r"(^|\n)\s*(from|import)\s+oracle|oracle/|oracle\.", txt), f


def test_flat_weight_file_roundtrip(tmp_path):
    sd = weights.synth_state_dict()
    p = str(tmp_path / "w.bsrnnw")
    weights.save_flat(p, sd)
    v, sd2 = weights.load_flat(p)
    assert v == spec.generate_bandsplits()[0]
    assert list(sd2.keys()) == list(sd.keys())
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)


def test_generators_are_deterministic():
    a = weights.synth_state_dict(seed=0)["lstms.1.m.rnn.weight_hh_l1"]
    b = weights.synth_state_dict(seed=0)["lstms.1.m.rnn.weight_hh_l1"]
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert abs(float(a.max()) - 0.125) < 1e-3           # U(-1/sqrt(64), 1/sqrt(64))
    w1 = weights.synth_waveform(4, 5000, seed=1234)
    w2 = weights.synth_waveform(2, 5000, seed=1234, row_offset=2)
    assert np.array_equal(w1[2:], w2)                    # a shard generates exactly its rows
    assert 0.09 < float(w1.std()) < 0.11
    # pinned first values (any change here invalidates the golden fixtures)
    assert w1[0, :3].tolist() == weights.synth_waveform(1, 3, seed=1234)[0].tolist()


def test_reference_checkpoint_interchange(tmp_path, built):
    """A checkpoint written the way the reference writes it (train.py:171: torch.save(model.state_dict(), ...)) loads into the
    host class through the safe loader, converts to the flat file of the C ABI / plugin and back without changing a bit."""
    import sys
    from collections import OrderedDict
    from speechseparation_amd import audio
    from speechseparation_amd.bsrnn import BSRNN
    sd = weights.synth_state_dict(None, seed=4)
    pth, flat, back = str(tmp_path / "model-always.pth"), str(tmp_path / "m.bsrnnw"), str(tmp_path / "back.pth")
    torch.save(OrderedDict((k, torch.from_numpy(a.copy())) for k, a in sd.items()), pth)
    m = BSRNN()
    assert audio.load_model_weights(m, pth) == pth
    got = m.state_dict()
    assert all(np.array_equal(got[k].numpy(), sd[k]) for k in sd)
    tool = os.path.join(REPO, "tools", "convert_weights.py")
    subprocess.run([sys.executable, tool, pth, flat], check=True, capture_output=True)
    v, sd2 = weights.load_flat(flat)
    assert v == spec.generate_bandsplits()[0] and list(sd2) == list(spec.param_spec()) and all(np.array_equal(sd2[k], sd[k]) for k in sd)
    subprocess.run([sys.executable, tool, flat, back], check=True, capture_output=True)
    sd3 = torch.load(back, map_location="cpu", weights_only=True)
    assert all(np.array_equal(sd3[k].numpy(), sd[k]) for k in sd)
    # a checkpoint of another architecture is refused, not half-loaded
    bad = str(tmp_path / "bad.pth")
    torch.save(OrderedDict(list((k, torch.from_numpy(a.copy())) for k, a in sd.items())[:-1]), bad)
    assert subprocess.run([sys.executable, tool, bad, flat], capture_output=True).returncode != 0


def test_io_signature_matches_the_exported_onnx_naming(built):
    """bsrnn_io_info: names and shapes of the reference's ONNX export (infer-streaming.py:74; speech-ladspa-onnx.cpp:82-111
    sizes its state from the input called "state.0").  Host-only context: no GPU needed."""
    from speechseparation_amd import _native
    lib = _native.lib
    v = spec.generate_bandsplits()[0]
    ctx = ctypes.c_void_p()
    widths = (ctypes.c_int32 * len(v))(*v)
    assert lib.bsrnn_create(-1, widths, len(v), ctypes.byref(ctx)) == 0, lib.bsrnn_last_error().decode()
    try:
        assert lib.bsrnn_io_count() == 4
        got = []
        for i in range(4):
            name, is_in, nd = ctypes.c_char_p(), ctypes.c_int32(), ctypes.c_int32()
            dims = (ctypes.c_int64 * 4)()
            assert lib.bsrnn_io_info(ctx, i, 2, ctypes.byref(name), ctypes.byref(is_in), dims, ctypes.byref(nd)) == 0
            got.append((name.value.decode(), bool(is_in.value), tuple(dims[:nd.value])))
        assert got == [("x.0", True, (2, 2050)), ("state.0", True, (4, 2, 24, 64)),
                       ("y.0", False, (2, 2050)), ("new_state.0", False, (4, 2, 24, 64))]
        assert lib.bsrnn_io_info(ctx, 4, 2, None, None, (ctypes.c_int64 * 4)(), ctypes.byref(ctypes.c_int32())) != 0
    finally:
        lib.bsrnn_destroy(ctx)


# ----------------------------------------------------------------------------- weight-file loader (host-only context)
def _host_ctx():
    """A host-only context (device -1): inventory + parameter staging + file validation, no GPU needed."""
    from speechseparation_amd import _native
    v = spec.generate_bandsplits()[0]
    widths = (ctypes.c_int32 * len(v))(*v)
    ctx = ctypes.c_void_p()
    _native.check(_native.lib.bsrnn_create(-1, widths, len(v), ctypes.byref(ctx)))
    return _native, ctx


def test_host_only_context_validates_but_does_not_compute(built, tmp_path):
    _native, ctx = _host_ctx()
    lib = _native.lib
    try:
        assert lib.bsrnn_device(ctx) == -1 and lib.bsrnn_param_count(ctx) == 288
        sd = weights.synth_state_dict(seed=0)
        path = str(tmp_path / "w.bin")
        weights.save_flat(path, sd)
        assert lib.bsrnn_load_weights_file(ctx, path.encode()) == 0          # every key, rank and shape matches the inventory
        k = "bandFCs.9.0.weight"
        back = np.empty(sd[k].shape, np.float32)
        assert lib.bsrnn_get_param(ctx, k.encode(), back.ctypes.data_as(ctypes.c_void_p), back.size) == 0
        assert np.array_equal(back, sd[k])
        x = np.zeros(4, np.float32)
        rc = lib.bsrnn_separate(ctx, x.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p), 1, 4096, None)
        assert rc == 2 and b"host-only" in lib.bsrnn_last_error()            # BSRNN_ESTATE: no CPU compute path exists
    finally:
        lib.bsrnn_destroy(ctx)


def test_weight_file_loader_rejects_malformed_files(built, tmp_path):
    """The loader trusts nothing in the file: truncation, huge or overflowing dims, a transposed matrix of the right element
    count, a wrong rank, an unknown key and a wrong band table are all refused with an error code - no exception crosses
    the C ABI and no allocation is sized from the file."""
    import struct
    _native, ctx = _host_ctx()
    lib = _native.lib
    sd = weights.synth_state_dict(seed=0)
    good = str(tmp_path / "good.bin")
    weights.save_flat(good, sd)
    blob = open(good, "rb").read()
    v = spec.generate_bandsplits()[0]
    head = 8 + 4 + 4 * len(v) + 4                                   # magic, band count, widths, tensor count
    first_key = next(iter(sd))                                      # bandFCs_pre.0.0.weight [2, 2]
    kl = len(first_key.encode())
    dims_at = head + 4 + kl + 4                                     # first tensor's dims
    EIO, ENOKEY = 4, 5

    def load(data, name):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        rc = lib.bsrnn_load_weights_file(ctx, p.encode())
        return rc, lib.bsrnn_last_error().decode()

    try:
        assert load(blob, "ok.bin")[0] == 0
        for cut in (7, head - 2, dims_at + 3, len(blob) // 2, len(blob) - 1):
            rc, msg = load(blob[:cut], "cut%d.bin" % cut)
            assert rc == EIO, (cut, rc, msg)
        # dims whose product overflows uint64 / would ask for exabytes: refused before anything is allocated
        huge = bytearray(blob)
        huge[dims_at:dims_at + 16] = struct.pack("<2Q", 1 << 40, 1 << 40)
        rc, msg = load(bytes(huge), "huge.bin")
        assert rc == EIO and "shape" in msg
        # same element count, transposed shape: must not be accepted silently
        k = "bandFCs.0.0.weight"                                     # [64, 2]
        tr = dict(sd)
        tr[k] = np.ascontiguousarray(sd[k].T)
        p = str(tmp_path / "tr.bin")
        weights.save_flat(p, tr)
        rc = lib.bsrnn_load_weights_file(ctx, p.encode())
        assert rc == EIO and "shape of 'bandFCs.0.0.weight'" in lib.bsrnn_last_error().decode()
        # wrong rank
        fl = dict(sd)
        fl[k] = sd[k].reshape(-1)
        weights.save_flat(p, fl)
        assert lib.bsrnn_load_weights_file(ctx, p.encode()) == EIO and "rank" in lib.bsrnn_last_error().decode()
        # unknown key
        uk = dict(sd)
        uk["not.a.key"] = np.zeros(3, np.float32)
        weights.save_flat(p, uk)
        rc = lib.bsrnn_load_weights_file(ctx, p.encode())
        assert rc in (EIO, ENOKEY)
        # absurd key length
        bad = bytearray(blob)
        bad[head:head + 4] = struct.pack("<I", 1 << 30)
        assert load(bytes(bad), "kl.bin")[0] == EIO
        # another band table
        weights.save_flat(p, sd, v=[1025, 0])
        assert lib.bsrnn_load_weights_file(ctx, p.encode()) == EIO
        assert lib.bsrnn_load_weights_file(ctx, str(tmp_path / "missing.bin").encode()) == EIO
    finally:
        lib.bsrnn_destroy(ctx)


def test_train_entry_point_dataset_layout(tmp_path):
    """train.py's data layout (m_dataset.samples(): <datapath>/<folder>/<clip>/mixture + speech) and its argument handling, on the CPU."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("train_entry", os.path.join(REPO, "train.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    for clip in ("b_clip", "a_clip", "incomplete"):
        d = tmp_path / "tr" / clip
        d.mkdir(parents=True)
        (d / "mixture.wav").write_bytes(b"")
        if clip != "incomplete":
            (d / "speech.wav").write_bytes(b"")
    (tmp_path / "tr" / "stray.txt").write_text("x")
    got = mod.dataset(str(tmp_path), "tr")
    assert [os.path.basename(os.path.dirname(a)) for a, _ in got] == ["a_clip", "b_clip"]
    assert all(a.endswith("mixture.wav") and b.endswith("speech.wav") for a, b in got)
    assert mod.dataset(str(tmp_path), "val") == []
    with pytest.raises(SystemExit):
        mod.main([])                                   # neither --datapath nor --synthetic
