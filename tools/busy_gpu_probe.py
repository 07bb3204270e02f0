#!/usr/bin/env python3
"""Diagnostic twin of tests/test_gpu_coresident.py::test_results_do_not_depend_on_a_busy_gpu: quiet results first, then the same calls
beside a second process; on a mismatch say where and how large, and whether the busy-GPU results agree with each other."""
import os, subprocess, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
sd = weights.synth_state_dict(None, seed=0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to("cuda")
w = torch.from_numpy(weights.synth_waveform(65, 16 * 1024 + 9, seed=31)).cuda()
quiet_stft = m.stft(w).cpu().numpy()
quiet = [m.separate(w).cpu().numpy() for _ in range(3)]
print("quiet runs equal:", all(np.array_equal(q, quiet[0]) for q in quiet))
quiet_istft = m.istft(torch.from_numpy(quiet_stft).cuda()).cpu().numpy()
env = dict(os.environ, PYTHONPATH=REPO)
load = subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "coresident_check.py"), "load", "150000"], env=env, cwd=REPO,
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
line = load.stdout.readline()
print("load:", line.strip())
xs = torch.from_numpy(quiet_stft).cuda()
busy = []
for i in range(10):
    a = np.array_equal(m.stft(w).cpu().numpy(), quiet_stft)
    b = np.array_equal(m.istft(xs).cpu().numpy(), quiet_istft)
    s = m.separate(w).cpu().numpy()
    busy.append(s)
    if not (a and b and np.array_equal(s, quiet[0])):
        d = np.abs(s - quiet[0]); idx = np.argwhere(d > 0)
        print("iteration %d: stft %s istft %s separate max diff %.3g, %d elements, rows %s, first sample %d last %d" % (
            i, a, b, d.max(), len(idx), sorted(set(idx[:, 0].tolist()))[:12], idx[:, 1].min() if len(idx) else -1, idx[:, 1].max() if len(idx) else -1))
print("busy runs equal each other:", all(np.array_equal(q, busy[0]) for q in busy), "; equal quiet:", np.array_equal(busy[0], quiet[0]))
load.kill(); load.wait()
