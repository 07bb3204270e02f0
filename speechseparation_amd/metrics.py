"""Host mirror of the reference's validation helpers (m_dataset.py:182-226), over the HIP path.

`train_infer(model, None, sample)` keeps the reference's call shape and return tuple
`(loss, sdr, sdr2, sdr3)` so that a validation loop written against m_dataset.py (train.py:137-150) runs
unchanged; the arithmetic itself is one C-ABI call (`bsrnn_evaluate`): separation, the clean signal's STFT
and every reduction stay on the device, only the eight numbers come back.  Training (the backward pass, the
discriminator term) is out of scope: a non-None discriminator is refused rather than ignored.
"""
import torch


def evaluate(model, mix, speech, return_estimate=False):
    """mix, speech: [R, n] (or the DataLoader's [1, R, n]) -> dict of the metrics (BSRNN.evaluate)."""
    if mix.dim() == 3 and mix.shape[0] == 1:
        mix, speech = mix.squeeze(0), speech.squeeze(0)        # m_dataset.py:184-185
    return model.evaluate(mix, speech, return_estimate=return_estimate)


def train_infer(model, discriminator, sample, lossfn=None, verbose=False):
    """m_dataset.py:202-226.  sample = (waveform, waveform_speech), each [1, R, n].  lossfn must be the reference's
    L1Loss(reduction='mean') (train.py:54) or None.  Returns 0-dim tensors like the reference (its validation loop calls
    `.item()` on them, train.py:140-144); they carry no graph: inference only."""
    if discriminator is not None:
        raise NotImplementedError("the discriminator term belongs to training, which this path does not cover")
    if lossfn is not None and not (isinstance(lossfn, torch.nn.L1Loss) and lossfn.reduction == "mean"):
        raise ValueError("only the reference's L1Loss(reduction='mean') is implemented on the device")
    m = evaluate(model, sample[0], sample[1])
    if verbose:
        print(" ".join("%s=%.4f" % kv for kv in m.items()))
    return tuple(torch.tensor(m[k], dtype=torch.float64) for k in ("loss", "sdr", "input_sdr", "sisdr"))
