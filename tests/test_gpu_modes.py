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
