#!/usr/bin/env python3
"""Measurement: distance of the dual-path step (4 recurrent blocks) of a library build to the float64 oracle,
beside the float32 oracle's own distance (same inputs as tests/test_gpu_parity.py::test_precision_is_at_fp32_rounding_level).
    BSRNN_HIP_LIB=build/ab/variant.so python tools/precision_dual_path.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import bsrnn_numpy as orc          # noqa: E402  (checker)
from speechseparation_amd import weights      # noqa: E402
from speechseparation_amd.bsrnn import BSRNN   # noqa: E402

for label, kw in (("default", dict(seed=0)), ("hot", dict(seed=1, lstm_gain=3.0))):
    sd = weights.synth_state_dict(None, **kw)
    m = BSRNN().eval()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to("cuda:0")
    for seed in (78, 79, 80):
        z = weights.synth_tensor((3, 24, 12, 64), seed=seed, scale=1.0)
        z64, _ = orc.dual_path(sd, z.astype(np.float64), None, np.float64)
        z32, _ = orc.dual_path(sd, z, None, np.float32)
        zo, _ = m.dual_path(torch.from_numpy(z).cuda())
        zo = zo.cpu().numpy()
        print("%s %-8s seed %d  |hip - f64| %.2e   |f32 oracle - f64| %.2e   max|z| %.2f"
              % (os.environ.get("BSRNN_HIP_LIB", "in-tree"), label, seed, np.abs(zo - z64).max(), np.abs(z32 - z64).max(), np.abs(z64).max()))
