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
    assert lib.bsrnn_abi_version() == 1


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
