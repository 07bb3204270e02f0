#!/usr/bin/env python3
"""Validation pass over (mixture, clean speech) file pairs -- the `with torch.no_grad()` block of the reference's
train.py:132-150 without the training around it: `train_infer(model, None, sample, l1loss)` per pair, then the averages
the reference prints ("Validation Loss", "Validation SDR") plus the SI-SDR it logs.  Separation, both STFTs and every
reduction run on the MI355X (one bsrnn_evaluate call per pair).

    validate.py --pairs mix1.wav speech1.wav [mix2.wav speech2.wav ...] [--weights model-always.pth]
"""
import argparse

import torch

from speechseparation_amd import audio, metrics
from speechseparation_amd.bsrnn import BSRNN


def main(argv=None):
    ap = argparse.ArgumentParser(description="Validate the BSRNN model on (mixture, speech) pairs")
    ap.add_argument("--pairs", type=str, nargs="+", required=True, metavar="WAV", help="mixture and clean file, alternating")
    ap.add_argument("--weights", type=str, default="model-always.pth")
    ap.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED")
    ap.add_argument("--device", type=str, default="cuda:0")
    args = ap.parse_args(argv)
    if len(args.pairs) % 2:
        ap.error("--pairs needs an even number of files")

    torch.set_grad_enabled(False)
    model = BSRNN().eval()
    audio.load_model_weights(model, args.weights, args.synthetic_weights)
    model = model.to(args.device)
    l1loss = torch.nn.L1Loss(reduction="mean")

    val_loss = val_sdr = val_sdr2 = val_sisdr = 0.0
    n_pairs = len(args.pairs) // 2
    for mix_path, speech_path in zip(args.pairs[0::2], args.pairs[1::2]):
        mix, _ = audio.load_wav(mix_path)
        speech, _ = audio.load_wav(speech_path)
        if mix.shape[0] == 1:                                       # mono -> two identical rows, as infer.py:26-27
            mix, speech = torch.cat((mix, mix), 0), torch.cat((speech, speech), 0)
        n = min(mix.shape[1], speech.shape[1])
        sample = (mix[None, :, :n].to(args.device), speech[None, :, :n].to(args.device))
        loss, sdr, sdr2, sisdr = metrics.train_infer(model, None, sample, l1loss)
        val_loss += loss.item(); val_sdr += sdr.item(); val_sdr2 += sdr2.item(); val_sisdr += sisdr.item()
    print("Validation Loss", val_loss / n_pairs, "Validation SDR", val_sdr / n_pairs)
    print("Validation input SDR", val_sdr2 / n_pairs, "Validation SI-SDR", val_sisdr / n_pairs)


if __name__ == "__main__":
    main()
