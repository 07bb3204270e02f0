"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI (ctypes) and the
Python mirror class, against (1) the golden fixtures produced by the reference itself and
(2) the numpy oracle on seeded inputs, plus size-independent properties at the full
BASELINE size (R=64, T=126).

Tolerance: north_star demands <= 1e-4 max-abs (fp32) on masks / waveforms at |y|max ~ 10;
asserted here at 1e-4 with the measured error printed.  Integer bookkeeping (band table,
shapes, state layout) is bit-exact by construction and checked in the CPU tests."""
import numpy as np
import pytest
import torch

from conftest import golden

import os

pytestmark = pytest.mark.gpu
TOL = 1e-4


def tol_for(ref):
    """1e-4 max-abs (north star) for every fp32-accurate mode; the reduced-precision fp16 GEMM mode (BASELINE config 2,
    selected by tests/test_gpu_modes.py through BSRNN_TEST_RELTOL) is held to 1e-2 of the largest reference value."""
    rel = float(os.environ.get("BSRNN_TEST_RELTOL", "0"))
    return rel * float(np.abs(ref).max()) if rel > 0 else TOL


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def t2n(t):
    return t.detach().cpu().numpy()


def make_model(sd, v=None):
    from speechseparation_amd.bsrnn import BSRNN
    m = BSRNN(v).eval()
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}, strict=True)
    return m.to("cuda")


@pytest.fixture(scope="module")
def model(sd_default):
    return make_model(sd_default)


@pytest.fixture(scope="module")
def model_hot(sd_hot):
    return make_model(sd_hot)


def test_native_library_is_loaded():
    from speechseparation_amd import _native
    assert _native.lib.bsrnn_abi_version() == 2
    assert "libbsrnn_hip.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("name,which", [("fwd_T8", "model"), ("fwd_hot_T8", "model_hot")])
def test_forward_mask_vs_reference(name, which, request):
    m = request.getfixturevalue(which)
    g = golden(name)
    x = torch.from_numpy(g["x"]).cuda()
    y, mask = m.forward_with_mask(x)
    e_y, e_m = maxabs(t2n(y), g["y"]), maxabs(t2n(mask), g["mask"])
    print("%s: y err %.3e  mask err %.3e  (vs fp64 ref: %.3e)" % (name, e_y, e_m, maxabs(t2n(y), g["y64"])))
    assert e_y < tol_for(g["y"]) and e_m < tol_for(g["mask"])
    assert torch.equal(m(x), y)                     # forward == forward_with_mask, deterministic
    assert x.equal(torch.from_numpy(g["x"]).cuda())  # input not modified


def test_dual_path_vs_reference(model):
    g = golden("lstms_T24")
    z, _ = model.dual_path(torch.from_numpy(g["z"]).cuda())
    e = maxabs(t2n(z), g["z_out"])
    print("dual path err %.3e" % e)
    assert e < 2e-5


def test_dual_path_taps_hot(model_hot, sd_hot):
    """dual path alone on the reference's own Z0 of the hot-weights forward (saturating gates)."""
    g = golden("fwd_hot_T8")
    z, _ = model_hot.dual_path(torch.from_numpy(g["z0"]).cuda())
    e = maxabs(t2n(z), g["z_after_3"])
    print("hot dual path err %.3e" % e)
    assert e < 5e-5


def test_forward_recurrent_and_chunks_vs_reference(model):
    g = golden("stream6")
    x = torch.from_numpy(g["x"]).cuda()
    C, _, L = x.shape
    state = torch.zeros((4, 2, C * 12, 64), device="cuda")
    for t in range(L):
        y, state = model.forward_recurrent(x[:, :, t].contiguous(), state)
        assert maxabs(t2n(y), g["y"][:, :, t]) < TOL
        if t == 0:
            assert maxabs(t2n(state), g["state_after_first"]) < 2e-5
    assert maxabs(t2n(state), g["state_final"]) < 2e-5
    # chunked 2 + 4 frames with state carry == frame-by-frame == offline
    s = torch.zeros_like(state)
    ya, s = model.forward_chunk(x[:, :, :2].contiguous(), s)
    yb, s = model.forward_chunk(x[:, :, 2:].contiguous(), s)
    assert maxabs(t2n(torch.cat((ya, yb), 2)), g["y"]) < TOL
    assert maxabs(t2n(s), g["state_final"]) < 2e-5
    assert maxabs(t2n(model(x)), g["y_offline"]) < TOL


def test_stft_istft_separate_vs_reference(model):
    g = golden("sandwich")
    wave = torch.from_numpy(g["wave"]).cuda()
    x = model.stft(wave)
    assert tuple(x.shape) == g["x"].shape
    e_x = maxabs(t2n(x), g["x"])
    e_i = maxabs(t2n(model.istft(torch.from_numpy(g["x"]).cuda())), g["istft_of_x"])
    out = model.separate(wave)
    assert tuple(out.shape) == g["wave_out"].shape
    e_w = maxabs(t2n(out), g["wave_out"])
    print("stft err %.3e  istft err %.3e  waveform->waveform err %.3e" % (e_x, e_i, e_w))
    assert e_x < 3e-5 and e_i < 1e-5 and e_w < TOL
    # unfused path equals fused path bit for bit (same kernels, same order)
    out2 = model.istft(model(model.stft(wave)))
    assert torch.equal(out, out2)


def test_streaming_loop_vs_reference(model):
    from speechseparation_amd.bsrnn import StreamingSeparator
    g = golden("streaming_ola")
    st = StreamingSeparator(model, channels=2)
    for ci in range(g["chunks"].shape[1]):
        out = st.step(torch.from_numpy(g["chunks"][:, ci, :].copy()).cuda())
        e = maxabs(t2n(out), g["out"][:, ci, :])
        assert e < TOL, (ci, e)
    assert maxabs(t2n(st.state()), g["state_final"]) < 2e-5
    # host-buffer entry point (used by the LADSPA plugin) gives the same numbers after reset
    st.reset()
    for ci in range(g["chunks"].shape[1]):
        out = st.step(torch.from_numpy(g["chunks"][:, ci, :].copy()))
        assert maxabs(t2n(out), g["out"][:, ci, :]) < TOL


def test_bands41_variant_vs_reference():
    from speechseparation_amd import weights
    g = golden("bands41_T3")
    v = g["v"].tolist()
    m = make_model(weights.synth_state_dict(v, seed=3), v)
    e = maxabs(t2n(m(torch.from_numpy(g["x"]).cuda())), g["y"])
    print("41-band err %.3e" % e)
    assert e < TOL


def test_vs_oracle_ragged_sizes(model, sd_default):
    """Seeded inputs at sizes that do not divide the tiles: C=3 rows, T=5 frames (M=15 rows,
    N=60 band sequences, 36 time sequences)."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    wave = weights.synth_waveform(3, 4 * 1024 + 77, seed=42)
    ref = onp.separate(sd_default, wave)
    out = model.separate(torch.from_numpy(wave).cuda())
    e = maxabs(t2n(out), ref)
    print("ragged separate err %.3e" % e)
    assert e < TOL
    x = onp.stft_interleaved(wave)
    state = weights.synth_tensor((4, 2, 36, 64), seed=7, scale=0.3)
    yr, sr = onp.forward_chunked(sd_default, x, state)
    y, s = model.forward_chunk(torch.from_numpy(x).cuda(), torch.from_numpy(state).cuda())
    assert maxabs(t2n(y), yr) < TOL and maxabs(t2n(s), sr) < 2e-5


def test_full_size_properties(model):
    """BASELINE metric shape R=64, 8 s @ 16 kHz (T=126): properties that need no oracle run.
    (a) dim-0 independence: any row of the batch equals the same row run alone (justifies
        batching and data-parallel sharding);
    (b) causality + state carry: chunked streaming over the same frames equals offline;
    (c) istft(stft(w)) == w[:, :len] (COLA) and the mask path is linear in x given the mask."""
    from speechseparation_amd import weights
    wave = torch.from_numpy(weights.synth_waveform(64, 128000, seed=1234)).cuda()
    out = model.separate(wave)
    assert tuple(out.shape) == (64, 125 * 1024) and bool(torch.isfinite(out).all())
    for r in (0, 17, 63):
        alone = model.separate(wave[r:r + 1].contiguous())
        assert maxabs(t2n(alone), t2n(out[r:r + 1])) < 2e-5
    x = model.stft(wave[:4].contiguous())                       # [4,2050,126]
    y_off = model(x)
    state = torch.zeros((4, 2, 4 * 12, 64), device="cuda")
    ys = []
    for a, b in ((0, 1), (1, 40), (40, 126)):
        y, state = model.forward_chunk(x[:, :, a:b].contiguous(), state)
        ys.append(y)
    assert maxabs(t2n(torch.cat(ys, 2)), t2n(y_off)) < 2e-5
    rec = model.istft(model.stft(wave[:2].contiguous()))
    assert maxabs(t2n(rec), t2n(wave[:2, :rec.shape[1]])) < 2e-6


def test_cpu_tensor_inputs_are_staged_through_the_gpu(model):
    g = golden("fwd_T8")
    y = model(torch.from_numpy(g["x"]))
    assert not y.is_cuda and maxabs(y.numpy(), g["y"]) < TOL


def test_errors_are_loud(model):
    from speechseparation_amd._native import NativeError
    with pytest.raises(ValueError):
        model(torch.zeros((2, 2049, 4), device="cuda"))
    with pytest.raises(ValueError):
        model.forward_recurrent(torch.zeros((2, 2050), device="cuda"), torch.zeros((4, 2, 23, 64), device="cuda"))
    with pytest.raises(NativeError):
        model.stft(torch.zeros((1, 1000), device="cuda"))     # reflect padding needs n > 1024


def test_compute_mode_is_reported(model):
    from speechseparation_amd import _native
    m = _native.compute_mode()
    assert m["gemm"] in ("f32", "fp16x2", "fp16", "bf16") and m["lstm"] in ("f32", "fp16x2")
    print("compute mode:", m)
    # the MLP chains run fused unless the exact-fp32 mode or BSRNN_MLP=layers asks for per-layer launches
    want = "layers" if (m["gemm"] == "f32" or os.environ.get("BSRNN_MLP") == "layers") else "fused"
    assert model.mlp_flow() == want


@pytest.mark.parametrize("which,fix", [("model", "sd_default"), ("model_hot", "sd_hot")])
def test_precision_is_at_fp32_rounding_level(which, fix, request):
    """The split-precision modes (fp32 operands as 16-bit pieces on the f16/bf16 matrix cores) must be as
    accurate as fp32 arithmetic itself: the distance of the HIP result to the float64 evaluation of the model
    may not exceed the distance of the float32 oracle to it by more than a small factor."""
    from oracle import bsrnn_numpy as orc
    from speechseparation_amd import weights
    m, sd = request.getfixturevalue(which), request.getfixturevalue(fix)
    x = weights.synth_tensor((3, 2050, 24), seed=77, scale=1.0)
    y64 = orc.forward(sd, x, dtype=np.float64)
    y32 = orc.forward(sd, x, dtype=np.float32)
    y = t2n(m(torch.from_numpy(x).cuda()))
    e_hip, e_f32 = maxabs(y, y64), maxabs(y32, y64)
    print("forward: |hip - f64| %.2e   |f32 oracle - f64| %.2e   (max|y| %.2f)" % (e_hip, e_f32, np.abs(y64).max()))
    assert e_hip <= 3 * e_f32 + 1e-7
    z = weights.synth_tensor((3, 24, 12, 64), seed=78, scale=1.0)
    z64, _ = orc.dual_path(sd, z.astype(np.float64), None, np.float64)
    z32, _ = orc.dual_path(sd, z, None, np.float32)
    zo, _ = m.dual_path(torch.from_numpy(z).cuda())
    e_hip, e_f32 = maxabs(t2n(zo), z64), maxabs(z32, z64)
    print("dual path: |hip - f64| %.2e   |f32 oracle - f64| %.2e   (max|z| %.2f)" % (e_hip, e_f32, np.abs(z64).max()))
    assert e_hip <= 3 * e_f32 + 1e-7
