#!/usr/bin/env python3
"""Measurement: one iteration of the reference's training loop (train.py:97-115, batch_size 1: train_infer -> backward ->
AdamW step -> zero_grad) on the library's training kernels, beside the same iteration with stock torch on the host CPU
(oracle/bsrnn_torch_cpu.py, the checker: a bounded number of iterations).  One JSON line per configuration.
    python tools/train_step_bench.py [--rows 2] [--seconds 8] [--steps 10] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechseparation_amd import spec, train, weights  # noqa: E402
from speechseparation_amd.bsrnn import BSRNN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay measurement (train.GraphedTrainStep)")
    a = ap.parse_args()
    n = int(a.seconds * 16000)
    sd = weights.synth_state_dict(None, seed=0)
    mix = torch.from_numpy(weights.synth_waveform(a.rows, n, seed=1))
    speech = torch.from_numpy(weights.synth_waveform(a.rows, n, seed=2))
    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(v, copy=True)) for k, v in sd.items()})
    m = m.to("cuda:0")
    opt = train.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2)
    mg, sg = mix.cuda(), speech.cuda()
    for _ in range(2):
        train.train_step(m, opt, mg, sg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = train.train_step(m, opt, mg, sg)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    # phases of one iteration, each closed by a device synchronise (so they add up to more than the pipelined step)
    ph = {}
    def lap(name, t):
        torch.cuda.synchronize()
        ph[name] = round(1e3 * (time.perf_counter() - t), 2)
        return time.perf_counter()
    t = time.perf_counter()
    lo, _ = train.train_loss(m, mg, sg)
    t = lap("forward+loss", t)
    lo.backward()
    t = lap("backward", t)
    opt.step(); opt.zero_grad()
    t = lap("adamw", t)
    T = 1 + n // 1024
    line = {"config": "train step (train.py:97-115, batch_size 1): %d rows x %.0f s @16 kHz (T=%d), exact-fp32 training kernels" % (a.rows, a.seconds, T),
            "ms_per_step": round(ms, 2), "row_frames_per_s": round(a.rows * T / (ms * 1e-3), 1), "loss_after": round(float(loss), 5), "phases_ms": ph}
    if not a.no_graph:
        # the same iteration as one hipGraph replay (fresh model and optimizer, same start: the losses must agree with the eager run's)
        m2 = BSRNN().train()
        m2.load_state_dict({k: torch.from_numpy(np.array(v, copy=True)) for k, v in sd.items()})
        m2 = m2.to("cuda:0")
        opt2 = train.AdamW(m2.parameters(), lr=1e-3, weight_decay=1e-2, capturable=True)
        step = train.GraphedTrainStep(m2, opt2, a.rows, n, warmup=1)
        for _ in range(2):                                   # one eager step, then capture + first replay
            step(mg, sg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            lg = step(mg, sg)
        torch.cuda.synchronize()
        gms = 1e3 * (time.perf_counter() - t0) / a.steps
        line["graph_ms_per_step"] = round(gms, 2)
        line["graph_row_frames_per_s"] = round(a.rows * T / (gms * 1e-3), 1)
        line["graph_loss_after"] = round(float(lg), 5)       # after 2 + steps iterations, as loss_after
    if not a.no_cpu:
        from oracle.bsrnn_torch_cpu import TorchCpuBSRNN   # checker / CPU baseline only
        ref = TorchCpuBSRNN(sd, spec.generate_bandsplits()[0])
        params = ref.trainable()
        opt_ref = torch.optim.AdamW([p for p in params.values() if p.numel() > 0], lr=1e-3, weight_decay=1e-2)
        win = torch.hann_window(2048)
        l1 = torch.nn.L1Loss(reduction="mean")
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            X = torch.stft(mix, n_fft=2048, hop_length=1024, return_complex=True, window=win)
            y = ref.forward_differentiable(torch.stack((X.real, X.imag), dim=2).reshape(a.rows, 2050, -1))
            yc = y.reshape(a.rows, -1, 2, y.shape[2])
            Y = torch.complex(yc[:, :, 0, :], yc[:, :, 1, :])
            xt = torch.istft(Y, n_fft=2048, hop_length=1024, window=win)
            S = torch.stft(speech, n_fft=2048, hop_length=1024, return_complex=True, window=win)
            lo = l1(xt, speech[:, :xt.shape[1]]) + l1(Y.real, S.real) + l1(Y.imag, S.imag)
            lo.backward()
            opt_ref.step()
            opt_ref.zero_grad()
            times.append(time.perf_counter() - t0)
        line["cpu_ms_per_step"] = round(1e3 * sorted(times)[1], 1)
        line["cpu_cores"] = torch.get_num_threads()
        line["gpu_over_cpu"] = round(line["cpu_ms_per_step"] / ms, 1)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
