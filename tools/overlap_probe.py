"""Diagnostic of the overlapped dual path under uneven load: contexts with different BSRNN_OVERLAP settings, same inputs, matrix products on
a second torch stream starting and stopping; every differing call is localised in the spectrogram domain (rows, first frame, bands).
    python tools/overlap_probe.py [rows] [frames] [calls] [mode ...]        modes: 0 1 band mask pub (default: 0 1)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from speechseparation_amd import spec, weights
from test_gpu_overlap import make_model

R = int(sys.argv[1]) if len(sys.argv) > 1 else 48
T = int(sys.argv[2]) if len(sys.argv) > 2 else 101
CALLS = int(sys.argv[3]) if len(sys.argv) > 3 else 60
modes = sys.argv[4:] or ["0", "1"]
sd = weights.synth_state_dict(None, seed=1, lstm_gain=3.0)
ser = make_model(sd, "0")
w = torch.from_numpy(weights.synth_waveform(R, (T - 1) * 1024 + 11, seed=7)).cuda()
x = ser.stft(w)
ref = ser(x).cpu().numpy()                      # [R, 2050, T]
v = spec.generate_bandsplits()[0]
edges = np.cumsum([0] + [2 * a for a in v])
a = torch.randn(4096, 4096, device="cuda")
side = torch.cuda.Stream()
import ctypes
from speechseparation_amd import _native
def peek(m, which, n):
    buf = np.empty(n, np.float32)
    _native.check(_native.lib.bsrnn_debug_peek(m._ctx, which, buf.ctypes.data_as(ctypes.c_void_p), n))
    return buf
NZ = R * T * 12 * 64
for mode in modes:
    m = make_model(sd, mode)
    bad = 0
    m(x); torch.cuda.synchronize()
    z1_ref, hb1_ref = peek(m, 1, NZ).reshape(R, T, 12, 64), peek(m, 2, 2 * NZ).reshape(R, T, 12, 128)
    def where(name, a, b):
        d = np.abs(a.astype(np.float64) - b)
        if not d.max():
            print("      %s: identical" % name); return
        idx = np.argwhere(d > 0)
        rows = sorted(set(idx[:, 0].tolist())); frs = sorted(set(idx[:, 1].tolist())); bds = sorted(set(idx[:, 2].tolist()))
        first = idx[idx[:, 1] == frs[0]]
        print("      %s: %d elements differ, max %.3e; rows %s frames %d..%d (%d) bands %s; at first frame %d: bands %s units %s" % (
            name, len(idx), d.max(), rows[:6], frs[0], frs[-1], len(frs), bds, frs[0], sorted(set(first[:, 2].tolist())),
            sorted(set(first[:, 3].tolist()))[:40]))
    for i in range(CALLS):
        if i % 3 != 2 and os.environ.get("PROBE_LOAD", "1") != "0":
            with torch.cuda.stream(side):
                for _ in range(1 + i % 4):
                    a = torch.tanh(a @ a * 1e-3)
        o = m(x).cpu().numpy()
        if not np.array_equal(o, ref):
            bad += 1
            d = np.abs(o.astype(np.float64) - ref)          # [R, 2050, T]
            rows = np.nonzero(d.max(axis=(1, 2)))[0]
            for r in rows[:3]:
                fr = np.nonzero(d[r].max(axis=0))[0]
                f0 = fr[0]
                bands = [b for b in range(len(v)) if v[b] and d[r, edges[b]:edges[b + 1], f0].max() > 0]
                print("mode %s call %d row %d: frames %d..%d (%d differ), first frame %d: bands %s max %.2e; overall max %.2e (|ref| max %.2e)" % (
                    mode, i, r, fr[0], fr[-1], len(fr), f0, bands, d[r, :, f0].max(), d[r].max(), np.abs(ref[r]).max()))
            if len(rows) > 3:
                print("   ... %d rows in all: %s" % (len(rows), rows.tolist()[:40]))
            if os.environ.get("PROBE_TERSE") == "1": continue
            where("Z1 (output of the first time-axis launch)", peek(m, 1, NZ).reshape(R, T, 12, 64), z1_ref)
            where("HB1 (fc shares of the second band block)", peek(m, 2, 2 * NZ).reshape(R, T, 12, 128), hb1_ref)
    torch.cuda.synchronize()
    print("mode %s: %d / %d calls differ; overlap_state %d" % (mode, bad, CALLS, m.overlap_state()))
