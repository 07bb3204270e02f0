"""CPU checks of the split-precision operand formats (speechseparation_amd/csrc/split_host.h): the host routines
that turn fp32 weights into the fp16x2 / bf16x3 pieces the matrix-core kernels consume."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

CSRC = os.path.join(REPO, "speechseparation_amd", "csrc")


@pytest.fixture(scope="module")
def split_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("split") / "split_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", CSRC, os.path.join(REPO, "tests", "cpp", "split_check.cpp"), "-o", exe], check=True)

    def run(values):
        a = np.ascontiguousarray(values, np.float32)
        out = subprocess.run([exe], input=a.tobytes(), stdout=subprocess.PIPE, check=True).stdout
        rec = np.frombuffer(out, dtype=np.dtype([("h", "<u2", 7), ("j", "<f4", 2)]))
        assert rec.shape[0] == a.shape[0]
        return a, rec["h"], rec["j"]
    return run


def _values():
    rng = np.random.default_rng(5)
    mags = 10.0 ** rng.uniform(-9, 4.8, 20000)
    v = (mags * rng.choice([-1.0, 1.0], mags.shape)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 6.1e-5, 0.1, 1 / 3, 1e-8], np.float32)
    return np.concatenate([v, edge])


def test_f16_conversion_matches_ieee_round_to_nearest_even(split_check):
    a, h, _ = split_check(_values())
    assert np.array_equal(h[:, 0], a.astype(np.float16).view(np.uint16))


def test_bf16_conversion_is_round_to_nearest_even(split_check):
    a, h, _ = split_check(_values())
    u = a.view(np.uint32).astype(np.uint64)
    ref = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    assert np.array_equal(h[:, 1], ref)


def test_fp16x2_pieces_carry_22_bits(split_check):
    a, h, j = split_check(_values())
    p0 = h[:, 2].view(np.float16).astype(np.float64)
    p1 = h[:, 3].view(np.float16).astype(np.float64)
    rec = p0 + p1 / 2048.0
    err = np.abs(rec - a.astype(np.float64))
    # 11 + 11 significant bits while the residual stays a normal fp16 number (|a| >= 2^-14): half an ulp of 22 bits;
    # below that the pieces sit on the absolute grid 2^-24 / 2048 = 2^-35
    bound = np.maximum(np.abs(a.astype(np.float64)) * 2.0 ** -22, 2.0 ** -35)
    assert np.all(err <= bound), float((err / bound).max())
    assert np.allclose(j[:, 0], rec.astype(np.float32), rtol=0, atol=0)


def test_fp16x2_saturates_at_the_fp16_maximum(split_check):
    a, h, _ = split_check(np.array([1e6, -3e38, 70000.0], np.float32))
    p0 = h[:, 2].view(np.float16).astype(np.float64)
    assert np.all(np.isfinite(p0)) and np.array_equal(np.abs(p0), [65504.0] * 3)


def test_bf16x3_pieces_are_exact(split_check):
    a, h, j = split_check(_values())
    pieces = (h[:, 4:7].astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    rec = pieces.sum(axis=1)
    normal = np.abs(a) >= 2.0 ** -100
    assert np.array_equal(rec[normal], a.astype(np.float64)[normal])
    assert np.array_equal(j[:, 1][normal], a[normal])
