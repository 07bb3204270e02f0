"""GPU: bsrnn_evaluate (the reference's validation arithmetic, m_dataset.py:182-226 + infer.py:44-47) through the
C ABI against the oracle's stock-PyTorch restatement on the same weights and signals.  Tolerances: 2e-3 dB on the
decibel figures (the reference sums in fp32, the device in double), 2e-5 relative on the L1 terms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DB_TOL, REL_TOL = 2e-3, 2e-5
KEYS = ("loss", "sdr", "input_sdr", "sisdr", "l1_time", "l1_re", "l1_im", "separation_db")


def _model_and_oracle(sd):
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec
    from speechseparation_amd.bsrnn import BSRNN
    m = BSRNN().eval()
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}, strict=True)
    return m.to("cuda"), TorchCpuBSRNN(sd, spec.generate_bandsplits()[0])


def _compare(got, ref):
    for k in KEYS:
        if k in ("sdr", "input_sdr", "sisdr", "separation_db"):
            assert abs(got[k] - ref[k]) < DB_TOL, (k, got[k], ref[k])
        else:
            assert abs(got[k] - ref[k]) <= REL_TOL * abs(ref[k]), (k, got[k], ref[k])


@pytest.mark.parametrize("rows,n", [(2, 8 * 1024 + 17), (3, 16384), (1, 2049)])
def test_evaluate_matches_the_oracle(sd_default, rows, n):
    from oracle import metrics_torch as mt
    from speechseparation_amd import weights
    model, oracle = _model_and_oracle(sd_default)
    mix = torch.from_numpy(weights.synth_waveform(rows, n, seed=40 + rows))
    speech = torch.from_numpy(weights.synth_waveform(rows, n, seed=50 + rows, scale=0.07))
    ref = mt.train_infer(oracle.forward, mix, speech)
    got = model.evaluate(mix.cuda(), speech.cuda(), return_estimate=True)
    print({k: (round(got[k], 5), round(ref[k], 5)) for k in KEYS})
    assert float((got["x_time"].cpu() - ref["x_time"]).abs().max()) < 1e-4
    _compare(got, ref)


def test_high_sdr_regime_and_reference_call_shape(sd_hot):
    """Target close to the model's own output (SDR ~ 40 dB, where the noise sums are small differences), through the
    reference's call shape train_infer(model, None, (mix[None], speech[None]), L1Loss)."""
    from oracle import metrics_torch as mt
    from speechseparation_amd import metrics, weights
    model, oracle = _model_and_oracle(sd_hot)
    n = 12 * 1024
    mix = torch.from_numpy(weights.synth_waveform(2, n, seed=7))
    x_ref = oracle.separate(mix)
    speech = torch.zeros_like(mix)
    speech[:, :x_ref.shape[1]] = x_ref + 0.01 * x_ref.abs().max() * torch.from_numpy(weights.synth_waveform(2, x_ref.shape[1], seed=8, scale=1.0))
    ref = mt.train_infer(oracle.forward, mix, speech)
    loss, sdr, sdr2, sdr3 = metrics.train_infer(model, None, (mix[None].cuda(), speech[None].cuda()), torch.nn.L1Loss(reduction="mean"))
    print("sdr %.4f (oracle %.4f)  si-sdr %.4f (oracle %.4f)" % (sdr, ref["sdr"], sdr3, ref["sisdr"]))
    assert ref["sdr"] > 20
    assert abs(loss - ref["loss"]) <= 5e-5 * abs(ref["loss"])
    assert abs(sdr - ref["sdr"]) < 5e-3 and abs(sdr2 - ref["input_sdr"]) < DB_TOL and abs(sdr3 - ref["sisdr"]) < 5e-3
    with pytest.raises(NotImplementedError):
        metrics.train_infer(model, object(), (mix[None], speech[None]))


def test_full_size_properties(sd_default):
    """Metric configuration (64 rows x 128000 samples): size-independent properties instead of the oracle.
    A target equal to the estimate gives a zero time-domain L1 term and the epsilon-limited SDR; rows are independent,
    so the mean over rows of per-row calls equals the batched call."""
    from speechseparation_amd import weights
    model, _ = _model_and_oracle(sd_default)
    mix = torch.from_numpy(weights.synth_waveform(64, 128000, seed=1234)).cuda()
    first = model.evaluate(mix, mix, return_estimate=True)
    est = first["x_time"]
    assert abs(first["separation_db"] - 10 * np.log(float((mix ** 2).sum().double() / ((mix - est) ** 2).sum().double()))) < 1e-3
    target = torch.zeros_like(mix)
    target[:, :est.shape[1]] = est
    m = model.evaluate(mix, target)
    assert m["l1_time"] == 0.0
    e2 = (est.double() ** 2).sum(1)
    assert abs(m["sdr"] - float((10 * torch.log10((e2 + 1e-9) / 1e-9)).mean())) < 1e-6
    rows = [model.evaluate(mix[r:r + 1], mix[r:r + 1]) for r in (0, 31, 63)]
    sub = model.evaluate(mix[[0, 31, 63]].contiguous(), mix[[0, 31, 63]].contiguous())
    assert abs(sub["sdr"] - np.mean([r["sdr"] for r in rows])) < 1e-4
    assert abs(sub["sisdr"] - np.mean([r["sisdr"] for r in rows])) < 1e-4
