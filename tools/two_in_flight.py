"""Measurement: two independent batches in flight - two model contexts on two streams, `separate()` calls alternating - against one
context on one stream (bench.py's timed loop).  Same workload as bench.py per call (R = 64 x 128000 samples).
    python tools/two_in_flight.py [steps=200]
Prints ms per step (= per batch) for: one context; two contexts, calls alternating (each context with the default overlapped dual path,
and with BSRNN_OVERLAP=0 set for both)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from speechseparation_amd import spec, weights


def run(n_ctx, steps, rows=64, samples=128000):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    models = [bench.build_model(dev)[0] for _ in range(n_ctx)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]
    wave = torch.from_numpy(weights.synth_waveform(rows, samples, seed=1234)).to(dev)
    T = spec.n_frames(samples)
    outs = [torch.empty((rows, (T - 1) * 1024), device=dev) for _ in range(n_ctx)]
    for m in models:
        m.set_range_policy("deferred")
    torch.cuda.synchronize()

    def loop(k):
        for i in range(k):
            j = i % n_ctx
            with torch.cuda.stream(streams[j]):
                models[j].separate(wave, out=outs[j])
    loop(80)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for j, m in enumerate(models):
        with torch.cuda.stream(streams[j]):
            m.sync()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    return dt / steps * 1e3, same, [m.overlap_state() for m in models]


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    torch.set_grad_enabled(False)
    for n in (1, 2, 3, 1, 2):
        ms, same, st = run(n, steps)
        print("contexts in flight %d: %.4f ms per batch of 64 rows (%.2f M row-frames/s), outputs of the contexts equal: %s, overlap state %s"
              % (n, ms, 64 * 126 / ms / 1e3, same, st), flush=True)
