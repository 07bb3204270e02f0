#!/usr/bin/env python3
"""Measurements for the BASELINE.json configs other than the headline one (bench.py):
config 2 (batch 32 x 8 s, here fp32), config 3 (streaming: L = 1 per-call latency and L = 256
chunks with state carry), config 5 (41-band table, 48 kHz x 8 s, batch 32).  One JSON line each."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechseparation_amd import spec, weights  # noqa: E402
from speechseparation_amd.bsrnn import BSRNN, StreamingSeparator  # noqa: E402


def model_for(v=None, seed=0):
    m = BSRNN(v).eval()
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in weights.synth_state_dict(v, seed=seed).items()})
    return m.to("cuda:0")


def timed(fn, warm=3, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    torch.set_grad_enabled(False)
    m = model_for()
    from speechseparation_amd import _native
    mode = _native.compute_mode()
    if mode["gemm"] in ("fp16", "bf16"):     # config 2 as named: 16-bit compute (BSRNN_GEMM=bf16 / fp16: plain 16-bit GEMM operands, fp32 accumulate / I/O)
        w = torch.from_numpy(weights.synth_waveform(32, 128000, seed=1234)).cuda()
        out = torch.empty((32, 125 * 1024), device="cuda")
        dt = timed(lambda: m.separate(w, out=out))
        print(json.dumps({"config": "2: offline R=32 x 8 s @16 kHz, 16-bit GEMM operands (BSRNN_GEMM=%s)" % mode["gemm"],
                          "ms_per_batch": round(dt * 1e3, 4), "row_frames_per_s": round(32 * 126 / dt, 1)}))
        return
    # config 2 in the fp32-accurate default mode
    w = torch.from_numpy(weights.synth_waveform(32, 128000, seed=1234)).cuda()
    out = torch.empty((32, 125 * 1024), device="cuda")
    dt = timed(lambda: m.separate(w, out=out))
    print(json.dumps({"config": "2: offline R=32 x 8 s @16 kHz, fp32-accurate default (fp16x2 split)", "ms_per_batch": round(dt * 1e3, 4), "row_frames_per_s": round(32 * 126 / dt, 1)}))
    # config 3: streaming
    for C in (2, 64):
        st = StreamingSeparator(m, channels=C)
        chunk = torch.from_numpy(weights.synth_waveform(C, 1024, seed=3)).cuda()
        dt = timed(lambda: st.step(chunk), warm=5, reps=200)
        print(json.dumps({"config": "3: streaming L=1 (one 1024-sample chunk per call), C=%d" % C, "us_per_step": round(dt * 1e6, 1),
                          "realtime_budget_us_at_44k1": 23220, "row_frames_per_s": round(C / dt, 1)}))
    for C in (2, 64):
        x = m.stft(torch.from_numpy(weights.synth_waveform(C, 256 * 1024, seed=4)).cuda())[:, :, :256].contiguous()
        state = torch.zeros((4, 2, C * 12, 64), device="cuda")
        dt = timed(lambda: m.forward_chunk(x, state), warm=2, reps=10)
        print(json.dumps({"config": "3: chunked streaming L=256 frames per call with state carry, C=%d" % C, "ms_per_chunk": round(dt * 1e3, 3),
                          "row_frames_per_s": round(C * 256 / dt, 1)}))
    # config 5: 41 bands (+ the zero-width band), 48 kHz x 8 s = 384000 samples, T = 376, batch 32
    v41 = spec.variant_bandsplits("41")
    m41 = model_for(v41, seed=3)
    w = torch.from_numpy(weights.synth_waveform(32, 384000, seed=5)).cuda()
    out = torch.empty((32, 375 * 1024), device="cuda")
    dt = timed(lambda: m41.separate(w, out=out), warm=2, reps=5)
    print(json.dumps({"config": "5: 41-band table (K=42), R=32 x 8 s @48 kHz (T=376), fp32", "ms_per_batch": round(dt * 1e3, 3),
                      "row_frames_per_s": round(32 * 376 / dt, 1)}))


if __name__ == "__main__":
    main()
