"""Streaming L = 1 step in a loop (for rocprofv3 --kernel-trace --stats): which kernels make up one 1024-sample chunk."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN, StreamingSeparator
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = BSRNN().eval()
m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in weights.synth_state_dict(None, seed=0).items()})
m = m.to("cuda:0")
st = StreamingSeparator(m, channels=C)
chunk = torch.from_numpy(weights.synth_waveform(C, 1024, seed=3)).cuda()
for _ in range(20):
    st.step(chunk)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    st.step(chunk)
torch.cuda.synchronize()
print("C=%d: %.1f us per step" % (C, (time.perf_counter() - t0) / n * 1e6))
