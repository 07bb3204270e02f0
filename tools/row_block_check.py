# Bit-exactness of a 130-row batch against its two row blocks, repeated (used to bisect the BSRNN_PARTS hazard):
#   PYTHONPATH=. BSRNN_PARTS=2 python tools/row_block_check.py 40
# For every dirty run it also reports whether a second device-to-host copy of the same device tensor differs from the
# first (stale lines at copy time) and the sizes of the dirty runs of samples (cache-line granularity?).
import numpy as np, torch, sys
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
sd = weights.synth_state_dict(None, seed=0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to('cuda')
wave = weights.synth_waveform(130, 16 * 1024 + 9, seed=31)
w = torch.from_numpy(wave).cuda()
h0 = m.separate(w[:65].contiguous()).cpu().numpy(); h1 = m.separate(w[65:].contiguous()).cpu().numpy()
halves = np.concatenate([h0, h1], 0)
bad = 0; reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for rep in range(reps):
    y = m.separate(w)
    a = y.cpu().numpy()
    if np.abs(a - halves).max() > 0:
        bad += 1
        torch.cuda.synchronize()
        b = y.cpu().numpy()
        d = (a != halves)
        runs = []
        for r in np.nonzero(d.any(1))[0][:4]:
            idx = np.nonzero(d[r])[0]
            splits = np.split(idx, np.nonzero(np.diff(idx) > 1)[0] + 1)
            runs.append([(int(s[0]), len(s)) for s in splits][:4])
        print("rep %d dirty: second copy equals first: %s; second copy clean: %s; dirty runs (start, length) %s" % (rep, np.array_equal(a, b), np.array_equal(b, halves), runs))
print("dirty runs: %d / %d" % (bad, reps))
