#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (the reference lives at /root/reference, read-only, and
never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is the reference's own output and what is restated:
  * every model tensor (y, mask, Z taps, states) comes from importing
    /root/reference/bsrnn.py and calling BSRNN.forward / forward_recurrent / lstms with
    our deterministic synthetic weights loaded through its own load_state_dict;
  * infer.py and infer-streaming.py cannot be imported (torchaudio / m_dataset are absent
    in this image), so their STFT sandwich (infer.py:29-37) and streaming loop
    (infer-streaming.py:116-145) are re-typed here with the same stock torch calls around
    the imported model.
Fixtures are data only: inputs and expected outputs as .npz.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from speechseparation_amd import spec, weights  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def load_ref(v=None, **kw):
    import bsrnn
    if v is not None:
        bsrnn.generate_bandsplits = lambda: (list(v), [0] * len(v))
    else:
        import importlib
        importlib.reload(bsrnn)
    m = bsrnn.BSRNN().eval()
    sd = weights.synth_state_dict(v, **kw)
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}, strict=True)
    return bsrnn, m


def sandwich_in(wave):
    win = torch.hann_window(2048)
    X = torch.stft(wave, n_fft=2048, hop_length=1024, return_complex=True, window=win)     # infer.py:31
    x = torch.stack((X.real, X.imag), dim=2)                                                # :32
    return x.reshape((x.shape[0], x.shape[1] * 2, x.shape[3]))                              # :33


def sandwich_out(y):
    win = torch.hann_window(2048)
    y = y.reshape((2, -1, 2, y.shape[2]))                                                   # :35
    Y = torch.complex(y[:, :, 0, :], y[:, :, 1, :])                                         # :36
    return torch.istft(Y, n_fft=2048, hop_length=1024, window=win)                          # :37


def taps_forward(m, x):
    """forward with hooks: mask = cat(pre_i + post_i) (bsrnn.py:425), Z after each lstms[j]."""
    taps = {}
    pre, post = {}, {}
    hs = []
    for i in range(len(m.bandFCs_pre)):
        hs.append(m.bandFCs_pre[i].register_forward_hook(lambda mod, a, out, i=i: pre.__setitem__(i, out)))
        hs.append(m.bandFCs_back_post[i].register_forward_hook(lambda mod, a, out, i=i: post.__setitem__(i, out)))
    for j in range(4):
        hs.append(m.lstms[j].register_forward_hook(lambda mod, a, out, j=j: taps.__setitem__("z_after_%d" % j, out.contiguous().numpy().copy())))
    hs.append(m.lstms.register_forward_hook(lambda mod, a, out: taps.__setitem__("z0", a[0].contiguous().numpy().copy())))
    y = m(x)
    for h in hs:
        h.remove()
    mask = torch.cat([pre[i] + post[i] for i in range(len(pre))], 2).permute(0, 2, 1)
    taps["mask"] = mask.contiguous().numpy().copy()
    taps["y"] = y.numpy().copy()
    return taps


def main():
    out = {}
    # ---- band table (integer, bit-exact) -------------------------------------------------
    import bsrnn as _b
    v, w = _b.generate_bandsplits()
    np.savez(os.path.join(HERE, "bandsplits.npz"), v=np.array(v, np.int64), w=np.array(w, np.int64))

    # ---- forward, default-scale weights, R=2, T=8 ---------------------------------------
    wave = torch.from_numpy(weights.synth_waveform(2, 7 * 1024, seed=1234))
    x = sandwich_in(wave)
    for name, kw in (("fwd_T8", dict(seed=0)), ("fwd_hot_T8", dict(seed=1, lstm_gain=3.0))):
        _, m = load_ref(None, **kw)
        t = taps_forward(m, x)
        m64 = m.double()
        y64 = m64(x.double()).numpy()
        m.float()
        np.savez(os.path.join(HERE, name + ".npz"), x=x.numpy(), y64=y64.astype(np.float64), **t)
        print(name, "x", tuple(x.shape), "|y|max", np.abs(t["y"]).max(), "fp32-fp64", np.abs(t["y"] - y64).max())

    # ---- dual-path alone on random Z [2,24,12,64] ---------------------------------------
    _, m = load_ref(None, seed=0)
    z = torch.from_numpy(weights.synth_tensor((2, 24, 12, 64), seed=77, scale=0.5))
    zo = m.lstms(z).contiguous().numpy()
    np.savez(os.path.join(HERE, "lstms_T24.npz"), z=z.numpy(), z_out=zo)

    # ---- streaming model steps: 6 frames of forward_recurrent from zero state ------------
    wave_s = torch.from_numpy(weights.synth_waveform(2, 7 * 1024, seed=99))
    xs = sandwich_in(wave_s)[:, :, 1:7]                      # frames 1..6 [2,2050,6]
    state = torch.zeros((4, 2, 24, 64))
    ys, st1 = [], None
    for t in range(xs.shape[2]):
        y, state = m.forward_recurrent(xs[:, :, t].contiguous(), state)
        ys.append(y.numpy().copy())
        if t == 0:
            st1 = state.numpy().copy()
    y_off = m(xs).numpy()                                    # offline forward on the same frames
    np.savez(os.path.join(HERE, "stream6.npz"), x=xs.numpy(), y=np.stack(ys, 2), state_after_first=st1,
             state_final=state.numpy().copy(), y_offline=y_off)
    print("stream vs offline", np.abs(np.stack(ys, 2) - y_off).max())

    # ---- infer.py sandwich: waveform -> waveform ------------------------------------------
    wave2 = torch.from_numpy(weights.synth_waveform(2, 8 * 1024 + 300, seed=5))      # ragged tail (n % hop != 0)
    x2 = sandwich_in(wave2)
    y2 = m(x2)
    w_out = sandwich_out(y2)
    np.savez(os.path.join(HERE, "sandwich.npz"), wave=wave2.numpy(), x=x2.numpy(), wave_out=w_out.numpy(),
             istft_of_x=sandwich_out(x2).numpy())
    print("sandwich", tuple(wave2.shape), "->", tuple(w_out.shape))

    # ---- infer-streaming.py loop (lines 84-147) restated around the imported model --------
    fft_window = 2048
    stft_window = torch.hann_window(fft_window)
    current_waveform = torch.zeros((2, fft_window))
    state = torch.zeros((4, 2, 24, 64))
    previous_speech = [torch.zeros((2, fft_window))] * 8
    chunks = torch.from_numpy(weights.synth_waveform(2, 5 * 1024, seed=321)).reshape(2, 5, 1024)
    outs = []
    for ci in range(5):
        waveform = chunks[:, ci, :]
        waveform = torch.cat((current_waveform[:, 1024:], waveform), 1)
        current_waveform = waveform
        xx = torch.fft.rfft(waveform * stft_window)
        xx = torch.stack((xx.real, xx.imag), dim=2)
        xx = xx.reshape((2, -1))
        xx, state = m.forward_recurrent(xx, state)
        xx = xx.reshape((2, -1, 2))
        xx = torch.complex(xx[:, :, 0], xx[:, :, 1])
        waveform = torch.fft.irfft(xx)
        previous_speech = [waveform, previous_speech[0]]
        sum_of_window = torch.zeros(1024)
        new_samples = torch.zeros((2, 1024))
        current_range = (0, 1024)
        for wf in previous_speech:
            win = stft_window[current_range[0]:current_range[1]]
            sum_of_window = sum_of_window + win
            new_samples = new_samples + wf[:, current_range[0]:current_range[1]]
            current_range = (current_range[0] + 1024, current_range[1] + 1024)
        new_samples = new_samples / sum_of_window
        outs.append(new_samples.numpy().copy())
    np.savez(os.path.join(HERE, "streaming_ola.npz"), chunks=chunks.numpy(), out=np.stack(outs, 1),
             state_final=state.numpy().copy())

    # ---- 41-band variant (monkeypatched band table), R=1, T=3 ----------------------------
    v41 = spec.variant_bandsplits("41")
    _, m41 = load_ref(v41, seed=3)
    x41 = sandwich_in(torch.from_numpy(weights.synth_waveform(1, 2 * 1024, seed=8)))
    y41 = m41(x41).numpy()
    np.savez(os.path.join(HERE, "bands41_T3.npz"), v=np.array(v41, np.int64), x=x41.numpy(), y=y41)
    print("bands41 params", sum(p.numel() for p in m41.parameters()))
    load_ref(None)   # restore module


def named_sizes():
    """Round 3: fixtures at the sizes BASELINE.json / SURVEY 8(c) name.  Inputs regenerate from the seeded
    generators (and, for spectra, from the same stock torch.stft call as infer.py:29-33), so only outputs are stored, plus a
    thin subsample of each regenerated input to prove the regeneration."""
    # ---- config 1: infer.py on one 4 s MONO 16 kHz mixture: 64 000 samples, duplicated to two rows (infer.py:26-27),
    #      T = 63 frames, output 63 488 samples (infer.py:29-37)
    _, m = load_ref(None, seed=0)
    mono = torch.from_numpy(weights.synth_waveform(1, 64000, seed=41))
    wave = torch.cat((mono, mono), 0)                                                       # infer.py:26-27
    x = sandwich_in(wave)
    y = m(x)
    w_out = sandwich_out(y)
    assert tuple(x.shape) == (2, 2050, 63) and tuple(w_out.shape) == (2, 63488)
    # "Separation dB" of infer.py:44-47 on the same tensors (natural log, as the reference prints it)
    sep = 10 * torch.log(torch.mean(torch.square(wave[:, :w_out.shape[1]])) / torch.mean(torch.square(wave[:, :w_out.shape[1]] - w_out)))
    np.savez(os.path.join(HERE, "cfg1_sandwich.npz"), wave_out=w_out.numpy(), x_sub=x.numpy()[:, ::41, ::7].copy(),
             y_sub=y.numpy()[:, ::41, ::7].copy(), separation_db=np.array([float(sep)], np.float64))
    print("cfg1 sandwich", tuple(wave.shape), "->", tuple(w_out.shape), "separation dB", float(sep))

    # ---- forward at R = 2 (two different rows), T = 63
    wave2 = torch.from_numpy(weights.synth_waveform(2, 64000, seed=42))
    x2 = sandwich_in(wave2)
    t = taps_forward(m, x2)
    y64 = m.double()(x2.double()).numpy()
    m.float()
    np.savez(os.path.join(HERE, "fwd_T63.npz"), y=t["y"], x_sub=x2.numpy()[:, ::41, ::7].copy(), mask_sub=t["mask"][:, ::41, ::7].copy(),
             z_after_3_sub=t["z_after_3"][:, ::7].copy(), y64_sub=y64[:, ::41, ::7].copy())
    print("fwd_T63 |y|max", np.abs(t["y"]).max(), "fp32-fp64", np.abs(t["y"] - y64).max())

    # ---- 41-band table, hot weights (LSTM matrices x 3), R = 2, T = 8, with the dual-path taps
    v41 = spec.variant_bandsplits("41")
    _, m41 = load_ref(v41, seed=4, lstm_gain=3.0)
    x41 = sandwich_in(torch.from_numpy(weights.synth_waveform(2, 7 * 1024, seed=9)))
    t41 = taps_forward(m41, x41)
    np.savez(os.path.join(HERE, "bands41_hot_T8.npz"), v=np.array(v41, np.int64), x_sub=x41.numpy()[:, ::41, :].copy(), **t41)
    print("bands41 hot", tuple(x41.shape), "|y|max", np.abs(t41["y"]).max())
    load_ref(None)   # restore module


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "named":
        named_sizes()
    else:
        main()
        named_sizes()
