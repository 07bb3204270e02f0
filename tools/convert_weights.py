#!/usr/bin/env python3
"""Weight interchange between the reference's checkpoints and the flat file of the C ABI / LADSPA plugin.

    convert_weights.py model-always.pth model-always.bsrnnw      # train.py:171 state_dict -> flat file (bsrnn_load_weights_file)
    convert_weights.py model-always.bsrnnw model-always.pth      # and back: a plain state_dict the reference's load_state_dict takes

The checkpoint is read with torch.load(weights_only=True) (nothing in the file is executed); keys and shapes are checked
against the model's inventory (288 tensors for the default band table) before anything is written.
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechseparation_amd import spec, weights  # noqa: E402


def check_inventory(sd, v):
    want = spec.param_spec(v)
    missing, extra = [k for k in want if k not in sd], [k for k in sd if k not in want]
    if missing or extra:
        raise SystemExit("state_dict does not match the band table %s: missing %s, unexpected %s" % (v, missing[:4], extra[:4]))
    for k, shape in want.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise SystemExit("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), tuple(shape)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--bands", default=None, help="band-table variant (spec.variant_bandsplits), default: generate_bandsplits()")
    args = ap.parse_args(argv)
    if args.src.endswith(".bsrnnw"):
        v, sd = weights.load_flat(args.src)
        check_inventory(sd, v)
        torch.save(OrderedDict((k, torch.from_numpy(np.array(a))) for k, a in sd.items()), args.dst)
    else:
        v = spec.variant_bandsplits(args.bands) if args.bands else spec.generate_bandsplits()[0]
        sd = torch.load(args.src, map_location="cpu", weights_only=True)
        sd = OrderedDict((k, t.detach().to(torch.float32).numpy()) for k, t in sd.items())
        check_inventory(sd, v)
        weights.save_flat(args.dst, OrderedDict((k, sd[k]) for k in spec.param_spec(v)), v)
    print("wrote", args.dst)


if __name__ == "__main__":
    main()
