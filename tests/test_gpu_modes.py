"""The alternate compute modes (BSRNN_GEMM / BSRNN_LSTM, read once per process) are held to the same parity and
precision tests as the default: each mode runs the reference-fixture and float64-precision tests in a child
process of its own."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gemm,lstm", [("f32", "f32"), ("bf16x3", "fp16x2"), ("fp16x2", "f32")])
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


def test_reduced_precision_fp16_mode():
    """BSRNN_GEMM=fp16: plain fp16 operands, one MFMA term (the 16-bit compute configuration of BASELINE.json);
    not fp32-accurate by design, held to 1e-2 of the largest reference value."""
    env = dict(os.environ, BSRNN_GEMM="fp16", BSRNN_LSTM="fp16x2", BSRNN_TEST_RELTOL="1e-2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-s",
                        "-k", "test_compute_mode_is_reported or test_forward_mask_vs_reference", "-p", "no:cacheprovider"],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "'gemm': 'fp16'" in r.stdout


def test_slab_flow_is_bit_identical():
    """BSRNN_GEMM_SLAB=1 (opt-in): the MLP intermediates travel pre-split in the weights' slab format and the consuming
    layers stage both operands by LDS-DMA (gemm_h2s_kernel).  Same pieces, same MFMA order: the separated waveform and the
    chunked forward must equal the default flow bit for bit (and it runs the reference-fixture tests)."""
    code = ("import numpy as np, torch, sys\n"
            "from speechseparation_amd import weights\n"
            "from speechseparation_amd.bsrnn import BSRNN\n"
            "sd = weights.synth_state_dict(None, seed=1, lstm_gain=3.0)\n"
            "m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to('cuda')\n"
            "w = torch.from_numpy(weights.synth_waveform(5, 9 * 1024 + 77, seed=3)).cuda()\n"
            "y = m.separate(w).cpu().numpy()\n"
            "x = m.stft(w[:2])\n"
            "s = torch.zeros((4, 2, 24, 64), device='cuda')\n"
            "z, s = m.forward_chunk(x[:, :, :3].contiguous(), s)\n"
            "np.savez(sys.argv[1], y=y, z=z.cpu().numpy(), s=s.cpu().numpy())\n")
    import tempfile
    import numpy as np
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for flag in ("0", "1"):
            path = os.path.join(d, "o%s.npz" % flag)
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, BSRNN_GEMM_SLAB=flag), cwd=REPO,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-3000:]
            outs.append({k: v for k, v in np.load(path).items()})
    for k in ("y", "z", "s"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    env = dict(os.environ, BSRNN_GEMM_SLAB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q",
                        "-k", "test_forward_mask_vs_reference or test_forward_recurrent_and_chunks_vs_reference", "-p", "no:cacheprovider"],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
