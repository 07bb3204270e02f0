"""CPU: the oracle's restatement of the reference's validation arithmetic (oracle/metrics_torch.py; m_dataset.py:182-226,
infer.py:44-47) against closed forms.  The reference module cannot be imported here (torchaudio / torchmetrics /
soundfile absent), so these properties are what pins the restatement; SI-SDR (torchmetrics' published algorithm)
stays "parity unpinned"."""
import numpy as np
import torch

from oracle import metrics_torch as mt


def _signals(R=2, n=9 * 1024 + 5, seed=3):
    g = torch.Generator().manual_seed(seed)
    s = 0.1 * torch.randn((R, n), generator=g)
    noise = 0.05 * torch.randn((R, n), generator=g)
    return s + noise, s, noise


def test_identity_model_gives_the_mixtures_own_figures():
    mix, s, noise = _signals()
    m = mt.train_infer(lambda x: x, mix, s)                 # mask == 1: x_time is the (trimmed) mixture
    n_est = (mix.shape[1] // 1024) * 1024
    assert tuple(m["x_time"].shape) == (2, n_est)
    assert float((m["x_time"] - mix[:, :n_est]).abs().max()) < 2e-6
    want = np.mean([10 * np.log10(np.sum(s[r, :n_est].numpy() ** 2) / np.sum(noise[r, :n_est].numpy() ** 2)) for r in range(2)])
    assert abs(m["sdr"] - want) < 1e-3
    assert abs(m["l1_time"] - float(noise[:, :n_est].abs().mean())) < 1e-6
    assert abs(m["loss"] - (m["l1_time"] + m["l1_re"] + m["l1_im"])) < 1e-6
    assert m["separation_db"] > 100                          # nothing was removed: ln(power / ~0)


def test_input_sdr_keeps_the_batch_dimension_quirk():
    _, s, noise = _signals(R=2, n=4096)
    s[:, :2048] *= 0.01                                      # a quiet first half: a mean of per-position ratios notices
    mix = s + noise
    m = mt.train_infer(lambda x: x, mix, s)
    a = (s.numpy() ** 2).sum(0) + 1e-9                      # over the two ROWS, per sample position
    b = (noise.numpy() ** 2).sum(0) + 1e-9
    assert abs(m["input_sdr"] - float(np.mean(10 * np.log10(a / b)))) < 1e-3
    per_row = np.mean(10 * np.log10((s.numpy() ** 2).sum(1) / (noise.numpy() ** 2).sum(1)))
    assert abs(m["input_sdr"] - per_row) > 0.5               # and it is NOT the conventional per-row figure


def test_si_sdr_is_scale_invariant_and_matches_its_closed_form():
    _, s, noise = _signals(R=3, n=8192)
    est = s + noise
    a = float(mt.si_sdr(est, s))
    assert abs(float(mt.si_sdr(3.7 * est, s)) - a) < 1e-3
    sn, en = s.double().numpy(), est.double().numpy()
    al = (en * sn).sum(1, keepdims=True) / (sn * sn).sum(1, keepdims=True)
    want = np.mean(10 * np.log10(((al * sn) ** 2).sum(1) / ((al * sn - en) ** 2).sum(1)))
    assert abs(a - want) < 1e-3


def test_separation_db_uses_the_natural_log():
    mix, s, _ = _signals()
    m = mt.train_infer(lambda x: 0.5 * x, mix, s)            # half of everything removed: ratio 4
    assert abs(m["separation_db"] - 10 * np.log(4.0)) < 1e-3
