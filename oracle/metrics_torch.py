"""ORACLE -- test infrastructure, never the product path.

The validation arithmetic of the reference restated with the same stock PyTorch CPU operators, fp32:
  * `infer(model, sample)`            m_dataset.py:182-199  (STFT sandwich around model.forward, STFT of the clean signal)
  * `train_infer(model, None, ...)`   m_dataset.py:202-226  (L1 tri-loss, sdr, sdr2, SI-SDR; no discriminator)
  * the "Separation dB" report        infer.py:44-47

Pinning.  m_dataset.py itself cannot be imported in the build container (torchaudio, torchmetrics and soundfile are not
installed, and it moves tensors to 'cuda' at import time, m_dataset.py:201), so these functions follow the source text.
SI-SDR comes from a third-party package that is absent here and from /root/reference: torchmetrics
`ScaleInvariantSignalDistortionRatio()` (zero_mean=False); its published algorithm is restated in `si_sdr`
-- PARITY UNPINNED for that number (no fixture of the reference holds one).  The other quantities are pinned by closed
forms in tests/test_metrics_oracle.py (known noise levels, scale invariance, the batch-dimension quirk of `sdr2`).
"""
import numpy as np
import torch


def infer(forward, mix, speech):
    """m_dataset.py:182-199.  forward: [C,2050,T] -> [C,2050,T]; mix, speech: [C, n] float32 (the squeezed sample)."""
    win = torch.hann_window(2048)
    X = torch.stft(mix, n_fft=2048, hop_length=1024, return_complex=True, window=win)
    x = torch.stack((X.real, X.imag), dim=2)
    x = x.reshape((x.shape[0], -1, x.shape[3]))
    y = forward(x)
    y = y.reshape((y.shape[0], -1, 2, y.shape[2]))          # the reference writes 2 for shape[0] (:192)
    Y = torch.complex(y[:, :, 0, :], y[:, :, 1, :])
    x_time = torch.istft(Y, n_fft=2048, hop_length=1024, window=win)
    S = torch.stft(speech, n_fft=2048, hop_length=1024, return_complex=True, window=win)
    return Y, x_time, S, speech[:, :x_time.shape[1]]


def si_sdr(preds, target):
    """torchmetrics.functional.audio.scale_invariant_signal_distortion_ratio, zero_mean=False; mean over rows
    (what the Metric object's compute() returns for one update)."""
    eps = torch.finfo(preds.dtype).eps
    alpha = (torch.sum(preds * target, dim=-1, keepdim=True) + eps) / (torch.sum(target ** 2, dim=-1, keepdim=True) + eps)
    target_scaled = alpha * target
    noise = target_scaled - preds
    val = (torch.sum(target_scaled ** 2, dim=-1) + eps) / (torch.sum(noise ** 2, dim=-1) + eps)
    return (10 * torch.log10(val)).mean()


@torch.no_grad()
def train_infer(forward, mix, speech):
    """m_dataset.py:202-226 with discriminator=None and lossfn=L1Loss(mean) (train.py:54).
    mix, speech: [C, n].  -> dict with the reference's (loss, sdr, sdr2, sdr3) and the parts."""
    l1 = torch.nn.L1Loss(reduction="mean")
    Y, x_time, S, s_time = infer(forward, mix, speech)
    l1_time, l1_re, l1_im = l1(x_time, s_time), l1(Y.real, S.real), l1(Y.imag, S.imag)
    loss = l1_time + l1_re + l1_im
    n2s = torch.sum(torch.square(s_time), dim=1) + 1e-9
    n2d = torch.sum(torch.square(x_time - s_time), dim=1) + 1e-9
    sdr = (10 * torch.log10(n2s / n2d)).mean()
    # sample[0], sample[1] still have the DataLoader's batch dimension: [1, C, n]; dim=1 is the row axis (:219-222)
    s1, s0 = speech[None], mix[None]
    n2s = torch.sum(torch.square(s1), dim=1) + 1e-9
    n2d = torch.sum(torch.square(s1 - s0), dim=1) + 1e-9
    sdr2 = (10 * torch.log10(n2s / n2d)).mean()
    sdr3 = si_sdr(x_time, s_time)
    # infer.py:41-47 (numpy on the cpu tensors)
    w = mix[:, :x_time.shape[1]].numpy()
    sep_db = 10 * np.log(np.sum(np.square(w)) / np.sum(np.square(w - x_time.numpy())))
    return {"loss": float(loss), "sdr": float(sdr), "input_sdr": float(sdr2), "sisdr": float(sdr3),
            "l1_time": float(l1_time), "l1_re": float(l1_re), "l1_im": float(l1_im), "separation_db": float(sep_db),
            "x_time": x_time}
