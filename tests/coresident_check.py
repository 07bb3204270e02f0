# Which entry point gives run-to-run different results when another process keeps the GPU busy?
#   PYTHONPATH=. python tests/coresident_check.py load 400000 &   (background load: separate() in a loop)
#   PYTHONPATH=. python tests/coresident_check.py stft|istft|forward|separate|stream|evaluate 60
# Test infrastructure (imports the oracle to tell which of two differing results is the right one).
import sys, numpy as np, torch
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
kind, reps = sys.argv[1], int(sys.argv[2])
sd = weights.synth_state_dict(None, seed=0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to('cuda')
w = torch.from_numpy(weights.synth_waveform(65, 16 * 1024 + 9, seed=31)).cuda()
if kind.startswith("load"):          # "load" = separate(); "load:stft" / "load:forward" / "load:dual" = one kernel family only
    what = kind.partition(":")[2] or "separate"
    xl = m.stft(w)
    zl = torch.zeros((65, 17, 12, 64), device="cuda")
    body = {"separate": lambda: m.separate(w), "stft": lambda: m.stft(w), "forward": lambda: m(xl), "dual": lambda: m.dual_path(zl)}[what]
    body(); torch.cuda.synchronize()
    print("load ready", flush=True)      # the test waits for this line: from here on the GPU is busy
    for _ in range(reps):
        body()
    torch.cuda.synchronize(); print("load done"); sys.exit(0)
x = m.stft(w)
def stream_run():
    from speechseparation_amd.bsrnn import StreamingSeparator
    st = StreamingSeparator(m, channels=2)
    return torch.cat([st.step(w[:2, i * 1024:(i + 1) * 1024].contiguous()) for i in range(12)], 1)
def evaluate_run():
    r = m.evaluate(w[:8], 0.5 * w[:8])
    return torch.tensor([r[k] for k in sorted(r)], dtype=torch.float64)
fn = {"stft": lambda: m.stft(w), "istft": lambda: m.istft(x), "forward": lambda: m(x), "separate": lambda: m.separate(w),
      "stream": stream_run, "evaluate": evaluate_run}[kind]
ref = fn().cpu().numpy()
from oracle import bsrnn_numpy as onp
wn = w.cpu().numpy()
truth = {"stft": lambda: onp.stft_interleaved(wn), "separate": lambda: onp.separate(sd, wn)}.get(kind)
truth = truth() if truth else None
if truth is not None:
    print("first result vs oracle: max err %.3g" % np.abs(ref - truth).max())
bad = 0
for _ in range(reps):
    out = fn().cpu().numpy()
    if not np.array_equal(out, ref):
        bad += 1
        if bad <= 3:
            d = np.abs(out - ref)
            idx = np.argwhere(d > 0)
            if truth is not None:
                print("  this run vs oracle: max err %.3g" % np.abs(out - truth).max())
            print("  max diff %.3g (max |ref| %.3g), %d elements differ; first %s last %s; distinct last-axis indices %s" % (
                d.max(), np.abs(ref).max(), len(idx), idx[0].tolist(), idx[-1].tolist(), sorted(set(idx[:, -1].tolist()))[:20]))
print("%s: differing runs %d / %d" % (kind, bad, reps))
