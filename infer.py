#!/usr/bin/env python3
"""Offline separation entry point -- same command line and outputs as the reference's infer.py
(--input/--output; writes the separated file, prints "Separation dB", writes the five re-mix
files mix_{100,90,50,20,-100}.wav), with the model on the MI355X HIP path.

Reference flow (infer.py:17-79): load model-always.pth -> load audio, mono duplicated to two
rows (:26-27) -> STFT (:29-33) -> BSRNN.forward (:34) -> iSTFT (:35-37) -> save -> report.
Here STFT -> forward -> iSTFT is one fused device call (`BSRNN.separate`), numerically the same
sandwich.  Extra flags: --weights (checkpoint or flat file), --synthetic-weights SEED (no
trained weights ship with the reference), --device, --outdir for the re-mix files.
"""
import argparse
import os

import numpy as np
import torch

from speechseparation_amd import audio
from speechseparation_amd.bsrnn import BSRNN

# (file suffix, gain of the non-dialog residue mixed back in; None = residue only)   infer.py:49-79
REMIXES = (("100", 0.0), ("90", 0.3), ("50", 0.5), ("20", 0.8), ("-100", None))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Infer the BSRNN model")
    ap.add_argument("--input", type=str, required=True, help="Input file")
    ap.add_argument("--output", type=str, required=True, help="Output file")
    ap.add_argument("--weights", type=str, default="model-always.pth")
    ap.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED")
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--outdir", type=str, default=".")
    args = ap.parse_args(argv)

    torch.set_grad_enabled(False)
    model = BSRNN().eval()
    audio.load_model_weights(model, args.weights, args.synthetic_weights)
    model = model.to(args.device)

    waveform, sr = audio.load_wav(args.input)
    if waveform.shape[0] == 1:                      # mono -> two identical rows
        waveform = torch.cat((waveform, waveform), 0)
    orig_peak = waveform.max().item()

    dialog = model.separate(waveform.to(args.device)).cpu()
    waveform = waveform[:, :dialog.shape[1]]
    audio.save_wav(args.output, dialog, sr)

    signal_power = np.sum(np.square(waveform.numpy()))
    remaining_power = np.sum(np.square((waveform - dialog).numpy()))
    print("Separation dB", 10 * np.log(signal_power / remaining_power))     # natural log, as the reference prints it

    residue = waveform - dialog
    for tag, gain in REMIXES:
        mix = residue if gain is None else dialog + gain * residue
        mix = mix * orig_peak / mix.max()
        audio.save_wav(os.path.join(args.outdir, "mix_%s.wav" % tag), mix, sr)


if __name__ == "__main__":
    main()
