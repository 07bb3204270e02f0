"""The alternate compute modes (BSRNN_GEMM / BSRNN_LSTM, read once per process) are held to the same parity and
precision tests as the default: each mode runs the reference-fixture and float64-precision tests in a child
process of its own."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gemm,lstm", [("f32", "f32"), ("fp16x2", "f32")])
def test_parity_in_mode(gemm, lstm):
    env = dict(os.environ, BSRNN_GEMM=gemm, BSRNN_LSTM=lstm)
    sel = "test_compute_mode_is_reported or test_forward_mask_vs_reference or test_precision_is_at_fp32_rounding_level " \
          "or test_forward_recurrent_and_chunks_vs_reference or test_stft_istft_separate_vs_reference"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-s",
                        "-k", sel, "-p", "no:cacheprovider"], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "'gemm': '%s'" % gemm in r.stdout and "'lstm': '%s'" % lstm in r.stdout


@pytest.mark.parametrize("gemm,reltol", [("fp16", "1e-2"), ("bf16", "1.5e-2")])
def test_reduced_precision_16bit_modes(gemm, reltol):
    """BSRNN_GEMM=fp16 / bf16: plain 16-bit operands, one MFMA term, in the per-band MLPs (BASELINE.json's config 2 names bf16;
    fp16 has three more mantissa bits and a range guard, bf16 the range of fp32); not fp32-accurate by design.  fp16 is held to 1e-2
    of the largest reference value, bf16 to 1.5e-2: the reference's own bf16 copy sits 8.7e-2 from its fp32 forward at |y|max ~ 12
    (0.7e-2 of the range) with EVERY layer in bf16; here only the MLP operands are (measured: printed)."""
    env = dict(os.environ, BSRNN_GEMM=gemm, BSRNN_LSTM="fp16x2", BSRNN_TEST_RELTOL=reltol)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-s",
                        "-k", "test_compute_mode_is_reported or test_forward_mask_vs_reference", "-p", "no:cacheprovider"],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "'gemm': '%s'" % gemm in r.stdout


_SEQ8_SCRIPT = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r)
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
sd = weights.synth_state_dict(None, seed=1, lstm_gain=3.0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}, strict=True); m = m.to("cuda")
out = []
for rows, n, seed in ((64, 128000, 3), (5, 40 * 1024 + 77, 4), (23, 17 * 1024, 5)):
    w = torch.from_numpy(weights.synth_waveform(rows, n, seed=seed)).cuda()
    out.append(hashlib.sha256(m.separate(w).cpu().numpy().tobytes()).hexdigest())
x = m.stft(torch.from_numpy(weights.synth_waveform(9, 33 * 1024, seed=6)).cuda())
s = torch.from_numpy(np.random.default_rng(2).standard_normal((4, 2, 9 * 12, 64)).astype(np.float32) * 0.3).cuda()
y, ns = m.forward_chunk(x[:, :, :20].contiguous(), s)
out.append(hashlib.sha256(y.cpu().numpy().tobytes() + ns.cpu().numpy().tobytes()).hexdigest())
print("HASHES " + " ".join(out))
"""


def test_time_axis_kernel_with_eight_sequences_per_workgroup_is_bit_identical():
    """lstm.hip::time_lstm_h2w8_kernel (launched by itself where four sequences per workgroup would need more than one round of workgroups) forced
    for every call (BSRNN_TIME_SEQ8=1, read once per process) against never (=0): separate() at the metric's size, ragged batches (a partial
    workgroup), a chunk with carried state in and out - the same bits."""
    hashes = []
    for v in ("0", "1"):
        env = dict(os.environ, BSRNN_TIME_SEQ8=v, BSRNN_OVERLAP="0")
        r = subprocess.run([sys.executable, "-c", _SEQ8_SCRIPT % REPO], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        hashes.append([l for l in r.stdout.splitlines() if l.startswith("HASHES ")][-1])
    assert hashes[0] == hashes[1], hashes
