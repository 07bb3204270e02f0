"""Diagnostic: offline forward run to run, and two chunks of 256 frames with state carry against offline, bit for bit, at R = 2, 8, 64
(how the missing barrier of the band kernels' first step was found: NOTEBOOK.md R3.9)."""
import numpy as np, torch, sys
import os
sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, sys_path)
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
sd = weights.synth_state_dict(None, seed=0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to("cuda")
for R in (2, 8, 64):
    wave = weights.synth_waveform(R, 511 * 1024 + 9, seed=60 + R)
    x = m.stft(torch.from_numpy(wave).cuda())
    y1 = m(x).clone(); y2 = m(x).clone()
    print("R", R, "offline run-to-run max diff", float((y1 - y2).abs().max()))
    state = torch.zeros((4, 2, R * 12, 64), device="cuda")
    ys = []
    for a in (0, 256):
        y, state = m.forward_chunk(x[:, :, a:a + 256].contiguous(), state)
        ys.append(y)
    yc = torch.cat(ys, 2)
    d = (yc - y1).abs()
    print("   chunks vs offline max", float(d.max()), "first chunk", float(d[:, :, :256].max()), "second", float(d[:, :, 256:].max()))
    # where
    idx = torch.nonzero(d > 0)
    if len(idx):
        print("   rows with differences:", sorted(set(idx[:, 0].tolist()))[:20], "first frame", int(idx[:, 2].min()))


