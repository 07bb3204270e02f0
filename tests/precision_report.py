#!/usr/bin/env python3
"""Measurement: how far is each compute mode of the HIP path from the exact answer?

Runs BSRNN.forward on seeded inputs (default weights and the "hot" set with 3x LSTM gain) through the
C-ABI in the mode given by the environment (BSRNN_GEMM / BSRNN_LSTM) and prints max-abs differences to
the numpy oracle evaluated in float32 and in float64.  The float64 run is the exact answer up to 1e-15;
the float32 oracle's own distance to it is the rounding noise any fp32 implementation carries.

    BSRNN_GEMM=f32 BSRNN_LSTM=f32 python tests/precision_report.py
    python tests/precision_report.py          # product default (fp16x2 split)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import bsrnn_numpy as orc          # noqa: E402  (test infrastructure, used as the checker)
from speechseparation_amd import weights      # noqa: E402
from speechseparation_amd.bsrnn import BSRNN   # noqa: E402


def main():
    C, T = 3, 24
    mode = "gemm=%s lstm=%s" % (os.environ.get("BSRNN_GEMM", "fp16x2"), os.environ.get("BSRNN_LSTM", "fp16x2"))
    for label, kw in (("default weights", dict(seed=0)), ("hot weights (3x LSTM gain)", dict(seed=1, lstm_gain=3.0))):
        sd = weights.synth_state_dict(None, **kw)
        x = weights.synth_tensor((C, 2050, T), seed=77, scale=1.0)
        y32 = orc.forward(sd, x, dtype=np.float32)
        y64 = orc.forward(sd, x, dtype=np.float64)
        m = BSRNN().eval()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        m = m.to("cuda:0")
        y = m(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
        print("%-28s %-26s max|y| %.3f  |hip - oracle32| %.2e  |hip - oracle64| %.2e  |oracle32 - oracle64| %.2e"
              % (label, mode, np.abs(y64).max(), np.abs(y - y32).max(), np.abs(y - y64).max(), np.abs(y32 - y64).max()))


if __name__ == "__main__":
    main()
