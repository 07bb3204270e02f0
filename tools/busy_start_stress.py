#!/usr/bin/env python3
"""Stress of the moment a second process starts using the GPU (where tests/test_gpu_coresident.py failed once in a while): quiet
result first, then N times {start a load process, run separate() 4 times beside it, stop it}; counts calls whose result differs."""
import os, subprocess, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sd = weights.synth_state_dict(None, seed=0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to("cuda")
w = torch.from_numpy(weights.synth_waveform(65, 16 * 1024 + 9, seed=31)).cuda()
x = m.stft(w)
quiet = m.separate(w).cpu().numpy()
quiet_f = m(x).cpu().numpy()
env = dict(os.environ, PYTHONPATH=REPO)
bad = badf = calls = 0
for r in range(rounds):
    load = subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "coresident_check.py"), "load", "150000"], env=env, cwd=REPO,
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    load.stdout.readline()
    for i in range(4):
        s = m.separate(w).cpu().numpy(); f = m(x).cpu().numpy(); calls += 1
        if not np.array_equal(s, quiet):
            bad += 1
            d = np.abs(s - quiet); idx = np.argwhere(d > 0)
            print("  round %d call %d: separate max diff %.3g in rows %s" % (r, i, d.max(), sorted(set(idx[:, 0].tolist()))[:8]))
        if not np.array_equal(f, quiet_f):
            badf += 1
            d = np.abs(f - quiet_f); idx = np.argwhere(d > 0)
            print("  round %d call %d: forward max diff %.3g in rows %s" % (r, i, d.max(), sorted(set(idx[:, 0].tolist()))[:8]))
    load.kill(); load.wait()
print("%s: separate %d / %d calls differ, forward %d / %d" % (os.environ.get("BSRNN_BAND_FC", "part"), bad, calls, badf, calls))
