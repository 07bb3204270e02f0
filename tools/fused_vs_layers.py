"""Fused MLP chains (default) against the per-layer flow (BSRNN_MLP=layers): same pieces, same k order - how close?
Runs both flows in child processes on the same inputs and compares separate(), forward (y and mask) and a streaming chunk.
    python tools/fused_vs_layers.py [rows] [frames]"""
import os, subprocess, sys, tempfile
import numpy as np

CODE = r'''
import sys, numpy as np, torch
from speechseparation_amd import weights
from speechseparation_amd.bsrnn import BSRNN
R, T = int(sys.argv[2]), int(sys.argv[3])
sd = weights.synth_state_dict(None, seed=1, lstm_gain=3.0)
m = BSRNN().eval(); m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}); m = m.to('cuda')
w = torch.from_numpy(weights.synth_waveform(R, (T - 1) * 1024 + 77, seed=3)).cuda()
y = m.separate(w).cpu().numpy()
x = m.stft(w[:3])
f, mask = m.forward_with_mask(x)
s = torch.zeros((4, 2, 3 * 12, 64), device='cuda')
z, s = m.forward_chunk(x[:, :, :3].contiguous(), s)
print("flow:", m.mlp_flow())
np.savez(sys.argv[1], y=y, f=f.cpu().numpy(), mask=mask.cpu().numpy(), z=z.cpu().numpy(), s=s.cpu().numpy())
'''
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 10)
outs = {}
with tempfile.TemporaryDirectory() as d:
    for flow in ("fused", "layers"):
        path = os.path.join(d, flow + ".npz")
        env = dict(os.environ, PYTHONPATH=repo)
        if flow == "layers":
            env["BSRNN_MLP"] = "layers"
        r = subprocess.run([sys.executable, "-c", CODE, path, str(R), str(T)], env=env, cwd=repo, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
        assert ("flow: " + flow) in r.stdout, r.stdout          # the child really ran the flow it was asked for
        outs[flow] = dict(np.load(path))
for k in outs["fused"]:
    a, b = outs["fused"][k], outs["layers"][k]
    d = np.abs(a.astype(np.float64) - b)
    print("%-5s shape %-18s max|ref| %.3g  max|fused - layers| %.3g  differing %d / %d  %s" % (
        k, a.shape, np.abs(b).max(), d.max(), int((d > 0).sum()), d.size, "BIT-IDENTICAL" if np.array_equal(a, b) else ""))
