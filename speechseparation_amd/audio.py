"""WAV I/O and resampling for the entry points (torchaudio is not available in this image;
the reference uses torchaudio.load/save, infer.py:24,42, and torchaudio.io.StreamReader with
sample_rate=44100, infer-streaming.py:77-78)."""
import numpy as np
import torch
from scipy.io import wavfile
from scipy.signal import resample_poly


def load_wav(path):
    """-> (float32 tensor [channels, n] in [-1, 1), sample_rate), like torchaudio.load."""
    sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(data.T)), int(sr)


def save_wav(path, waveform, sr):
    """waveform [channels, n] float -> 32-bit float WAV (what torchaudio.save writes for float32)."""
    a = waveform.detach().cpu().numpy() if isinstance(waveform, torch.Tensor) else np.asarray(waveform)
    wavfile.write(path, int(sr), np.ascontiguousarray(a.T.astype(np.float32)))


def resample(waveform, sr_from, sr_to):
    if sr_from == sr_to:
        return waveform
    g = np.gcd(int(sr_from), int(sr_to))
    out = resample_poly(waveform.numpy(), sr_to // g, sr_from // g, axis=1)
    return torch.from_numpy(np.ascontiguousarray(out.astype(np.float32)))


def load_model_weights(model, path=None, synthetic_seed=None):
    """Weights for the entry points: a reference checkpoint (`model-always.pth`, a plain
    state_dict saved by train.py:167; loaded with weights_only=True), our flat file, or
    deterministic synthetic weights when none exists (no trained weights ship with the reference)."""
    from . import weights
    if synthetic_seed is not None:
        sd = weights.synth_state_dict(model.band_widths, seed=synthetic_seed)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        return "synthetic(seed=%d)" % synthetic_seed
    if path.endswith(".bsrnnw"):
        v, sd = weights.load_flat(path)
        if v != list(model.band_widths):
            raise ValueError("%s holds band table %s" % (path, v))
        model.load_state_dict({k: torch.from_numpy(a) for k, a in sd.items()})
    else:
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return path
