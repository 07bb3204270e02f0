#!/usr/bin/env python3
"""What the two event brackets of bench.py's timed loop cost the stream: K calls of separate() with and without them, alternating."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
sd = weights.synth_state_dict(None, seed=0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to("cuda")
w = torch.from_numpy(weights.synth_waveform(64, 128000, seed=1234)).cuda()
out = torch.empty((64, 125 * 1024), device="cuda")
m.set_range_policy("deferred")
for _ in range(100): m.separate(w, out=out)
def run(prof, K=200):
    m.set_profiling(prof, "cuda:0"); m.stage_times(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): m.separate(w, out=out)
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0) / K
    m.stage_times(reset=True)
    return dt
for r in range(3):
    a = run(False); b = run(("bandsplit_mlp", "mask_mlp")); c = run(True)
    print("no brackets %.4f ms | the two chain brackets %.4f ms | all stages %.4f ms" % (a, b, c))
