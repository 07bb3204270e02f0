"""Full-size GPU tests of the BASELINE.json configurations that the fixture tests only touch at R <= 2, T <= 8:
config 5 (41-band table, K = 42, R = 32, 48 kHz x 8 s: T = 376) and config 2 (R = 32, T = 126, 16-bit GEMM operands),
plus the fused MLP chains against the per-layer flow.  Size-independent properties (row independence, chunked ==
offline, causality) and the oracle on a corner of the same batch."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def make_model(sd, v=None):
    from speechseparation_amd.bsrnn import BSRNN
    m = BSRNN(v).eval()
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}, strict=True)
    return m.to("cuda")


def test_config5_41_bands_full_size():
    """K = 42 (41 bands + the zero-width band), R = 32 rows x 384 000 samples (T = 376): L = 42 band sequences, 1344 time
    sequences, 43 MLP chains of five geometries.  (a) any row equals the same row run alone; (b) chunked streaming over the
    same frames with state carry equals offline; (c) the first frames of two rows equal the numpy oracle run on just those
    frames (the model is causal in time: bsrnn.py:106-128)."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import spec, weights
    v = spec.variant_bandsplits("41")
    sd = weights.synth_state_dict(v, seed=3)
    m = make_model(sd, v)
    assert m.mlp_flow() == "fused"
    wave = weights.synth_waveform(32, 384000, seed=5)
    w = torch.from_numpy(wave).cuda()
    out = m.separate(w)
    assert tuple(out.shape) == (32, 375 * 1024) and bool(torch.isfinite(out).all())
    for r in (0, 13, 31):
        alone = m.separate(w[r:r + 1].contiguous())
        assert maxabs(alone.cpu().numpy(), out[r:r + 1].cpu().numpy()) < 2e-5, r
    x = m.stft(w[:3].contiguous())                                   # [3, 2050, 376]
    y_off = m(x)
    state = torch.zeros((4, 2, 3 * len(v), 64), device="cuda")
    ys = []
    for a, b in ((0, 1), (1, 130), (130, 376)):
        y, state = m.forward_chunk(x[:, :, a:b].contiguous(), state)
        ys.append(y)
    assert maxabs(torch.cat(ys, 2).cpu().numpy(), y_off.cpu().numpy()) < 2e-5
    x8 = x[:2, :, :8].cpu().numpy()
    ref = onp.forward(sd, x8, v)
    e = maxabs(y_off[:2, :, :8].cpu().numpy(), ref)
    print("41-band full size: first 8 frames of rows 0-1 vs oracle %.3e (|y|max %.3g)" % (e, np.abs(ref).max()))
    assert e < 1e-4


CODE = r'''
import sys, numpy as np, torch
from speechseparation_amd import spec, weights
from speechseparation_amd.bsrnn import BSRNN
kind, path = sys.argv[1], sys.argv[2]
v = spec.variant_bandsplits("41") if kind == "b41" else None
sd = weights.synth_state_dict(v, seed=1, lstm_gain=3.0)
m = BSRNN(v).eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to("cuda")
print("flow:", m.mlp_flow())
R, n = (32, 128000) if kind == "cfg2" else ((9, 13 * 1024 + 77) if kind == "b12" else (5, 6 * 1024 + 5))
w = torch.from_numpy(weights.synth_waveform(R, n, seed=3)).cuda()
y = m.separate(w).cpu().numpy()
x = m.stft(w[:3].contiguous())
f, mask = m.forward_with_mask(x)
s = torch.zeros((4, 2, 3 * len(m.band_widths), 64), device="cuda")
z, s = m.forward_chunk(x[:, :, :3].contiguous(), s)
np.savez(path, y=y, f=f.cpu().numpy(), mask=mask.cpu().numpy(), z=z.cpu().numpy(), s=s.cpu().numpy())
'''


def run_child(kind, env_extra, d, tag):
    path = os.path.join(d, tag + ".npz")
    env = dict(os.environ, PYTHONPATH=REPO, **env_extra)
    r = subprocess.run([sys.executable, "-c", CODE, kind, path], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout, dict(np.load(path))


@pytest.mark.parametrize("kind,no48", [("b12", True), ("b41", False)])
def test_fused_chains_equal_the_per_layer_flow_bit_for_bit(kind, no48):
    """The fused chain kernel (mlp_chain.hip, the default) multiplies the same fp16 pieces in the same k order as the
    per-layer launches (BSRNN_MLP=layers) wherever it uses the same 32 x 32 x 16 MFMA: separate(), forward (output and
    mask) and a chunk with state must be EQUAL, for the 12-band table (geometries of 32 / 64 / 128 / 256 rows per
    workgroup, ragged row counts; BSRNN_CHAIN_NO48 keeps its 768- and 384-wide bands on the 32 x 32 geometry, BSRNN_CHAIN_RAG=0 the
    514-wide band's last feature tile on one wave instead of split over the k-steps of eight) and the 41-band table (BSRNN_CHAIN_NO80:
    none of its bands on the 80-row 16 x 16 geometry)."""
    extra = {"BSRNN_CHAIN_NO48": "1", "BSRNN_CHAIN_RAG": "0"} if no48 else {"BSRNN_CHAIN_NO80": "1"}
    with tempfile.TemporaryDirectory() as d:
        out_f, fused = run_child(kind, extra, d, "fused")
        out_l, layers = run_child(kind, dict(extra, BSRNN_MLP="layers"), d, "layers")
    assert "flow: fused" in out_f and "flow: layers" in out_l
    for k in fused:
        assert np.array_equal(fused[k], layers[k]), k


def test_48_row_geometry_of_the_widest_band_is_at_rounding_level():
    """By default the 768-wide band runs 48 rows and the 384-wide band 80 rows per workgroup on v_mfma_f32_16x16x32_f16 (the same
    products, summed 32 instead of 16 per instruction) and the last two columns of the 514-wide band are summed as eight k-slices (one per
    wave): not bit-identical to the 32 x 32 x 16 per-layer kernels - but at fp32 rounding level
    (last-bit differences of Z pass through the four recurrent blocks: 5e-7 ... 3e-6 of the range observed, bound 1e-5,
    a tenth of the parity tolerance)."""
    with tempfile.TemporaryDirectory() as d:
        out_f, fused = run_child("b12", {}, d, "fused")
        out_l, layers = run_child("b12", {"BSRNN_MLP": "layers"}, d, "layers")
    assert "flow: fused" in out_f and "flow: layers" in out_l
    for k in fused:
        rel = maxabs(fused[k], layers[k]) / np.abs(layers[k]).max()
        print("%s: 48-row geometry vs per-layer flow, relative to the range %.2e" % (k, rel))
        assert rel < 1e-5, (k, rel)


@pytest.mark.parametrize("kind", ["b12", "b41"])
def test_dual_path_flows_agree_at_rounding_level(kind):
    """The recurrent blocks have three ways through the library, selected per process: the default (the band block's fc formed as two
    per-direction shares inside the second band layer and added, with the residual, by the time-axis launch's staging wave; the time
    block's fc inside the 16-wave time kernel), BSRNN_BAND_FC=gemm (the band block's fc + residual as a grouped-GEMM launch, rounds
    1-2) and BSRNN_TIME_KERNEL=v2 on top of it (round 2's 8-wave time kernel + a GEMM launch for its fc).  Same fp16x2 products, sums
    taken in a different order for the fc: everything the model returns agrees to 1e-5 of its range; the two GEMM-fc flows that
    differ only in the time kernel (bit-identical h and state by construction) agree to the same bound."""
    with tempfile.TemporaryDirectory() as d:
        _, part = run_child(kind, {}, d, "part")
        _, unpaired = run_child(kind, {"BSRNN_BAND_PAIR": "0", "BSRNN_BAND_GRID": "2d"}, d, "unpaired")
        _, fallback = run_child(kind, {"BSRNN_BAND_PAIR": "mismatch"}, d, "fallback")
        _, gemm = run_child(kind, {"BSRNN_BAND_FC": "gemm"}, d, "gemm")
        _, v2 = run_child(kind, {"BSRNN_BAND_FC": "gemm", "BSRNN_TIME_KERNEL": "v2"}, d, "v2")
    # BSRNN_BAND_PAIR=0 is round 2's flow entirely (one launch per band layer on the 2-D grid, the fc as a GEMM launch): it must be the
    # GEMM-fc flow bit for bit - with or without the pair launch the band layers do the same arithmetic on the same numbers
    # (the pair launch with the fc shares against separate launches, bit for bit: tools/band_parts_check.hip)
    for k in part:
        assert np.array_equal(gemm[k], unpaired[k]), k
    # BSRNN_BAND_PAIR=mismatch (test hook): every workgroup of the pair launch finds its partner "on another XCD"; the first call notices
    # (guard value 4), runs again with one launch per layer and the context stays on that flow - same numbers as BSRNN_BAND_PAIR=0, rc 0
    for k in part:
        assert np.array_equal(fallback[k], unpaired[k]), k
    for name, a, b in (("parts vs gemm fc", part, gemm), ("16-wave vs 8-wave time kernel", gemm, v2)):
        for k in a:
            rel = maxabs(a[k], b[k]) / np.abs(b[k]).max()
            print("%s, %s: %.2e of the range" % (name, k, rel))
            assert rel < 1e-5, (name, k, rel)


@pytest.mark.parametrize("gemm,tol", [("fp16", 1e-2), ("bf16", 1.5e-2)])
def test_config2_16bit_gemm_mode_full_size(gemm, tol):
    """BASELINE config 2: R = 32 x 8 s @ 16 kHz with 16-bit GEMM operands (BSRNN_GEMM=bf16, as the config names it, and fp16: one
    MFMA term, fp32 accumulate, the LSTMs stay fp16x2) against the fp32-accurate default on the same batch: within 1e-2 (fp16) /
    1.5e-2 (bf16: 8 significant bits, measured 3e-3 ... 5e-3; the reference's own bf16 copy is 0.7e-2 of the range from its fp32 forward) of the output range."""
    with tempfile.TemporaryDirectory() as d:
        _, ref = run_child("cfg2", {}, d, "default")
        out, low = run_child("cfg2", {"BSRNN_GEMM": gemm}, d, gemm)
    assert "flow: fused" in out
    for k in ("y", "f", "mask"):
        rel = maxabs(low[k], ref[k]) / np.abs(ref[k]).max()
        print("%s GEMM mode, %s: relative to the output range %.2e" % (gemm, k, rel))
        assert np.isfinite(low[k]).all() and rel < tol, (k, rel)
