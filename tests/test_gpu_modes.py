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
