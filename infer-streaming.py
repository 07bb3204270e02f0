#!/usr/bin/env python3
"""Streaming separation entry point -- same command line as the reference's
infer-streaming.py (--input/--output/--name), with the whole per-chunk loop
(infer-streaming.py:104-147: slide buffer, rfft, forward_recurrent, irfft, 2-slot overlap-add)
resident on the MI355X: one `StreamingSeparator.step` per 1024-sample chunk at 44.1 kHz.

The reference also dumps the traced model for its LADSPA plugin (hello.onnx, :74); the
counterpart here is the flat weight file `hello.bsrnnw` that speech_separator_ladspa.so loads
(the model name is recorded next to it in hello.name).  ONNX export itself is out of scope
for this round.
"""
import argparse
import time

import torch

from speechseparation_amd import audio
from speechseparation_amd.bsrnn import BSRNN, StreamingSeparator

STREAM_RATE = 44100      # infer-streaming.py:78
CHUNK = 1024


def main(argv=None):
    ap = argparse.ArgumentParser(description="Infer the BSRNN model")
    ap.add_argument("--input", type=str, required=True, help="Input file")
    ap.add_argument("--output", type=str, required=True, help="Output file")
    ap.add_argument("--name", type=str, default="bsrnn", help="Model name")
    ap.add_argument("--weights", type=str, default="model-always.pth")
    ap.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED")
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--export", type=str, default="hello.bsrnnw")
    args = ap.parse_args(argv)

    torch.set_grad_enabled(False)
    model = BSRNN().eval()
    audio.load_model_weights(model, args.weights, args.synthetic_weights)
    model = model.to(args.device)
    if args.export:
        model.save_flat(args.export)
        with open(args.export.rsplit(".", 1)[0] + ".name", "w") as f:
            f.write("model=%s\n" % args.name)
        print("Flat weight file dumped")

    waveform, sr = audio.load_wav(args.input)
    waveform = audio.resample(waveform, sr, STREAM_RATE)
    if waveform.shape[0] == 1:
        waveform = torch.cat((waveform, waveform), 0)
    waveform = waveform[:2].to(args.device)

    sep = StreamingSeparator(model, channels=2, device=args.device)
    n_chunks = waveform.shape[1] // CHUNK                 # a short tail chunk ends the stream (:108)
    outs = []
    t = time.time()
    for i in range(n_chunks):
        outs.append(sep.step(waveform[:, i * CHUNK:(i + 1) * CHUNK].contiguous()))
    torch.cuda.synchronize()
    elapsed = time.time() - t
    out = torch.cat(outs, 1) if outs else torch.zeros((2, 0))
    audio.save_wav(args.output, out, STREAM_RATE)
    print(f"Elapsed {elapsed}")


if __name__ == "__main__":
    main()
