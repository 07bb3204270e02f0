#!/usr/bin/env python3
"""Training entry point -- the loop of the reference's train.py (:55-176) with every parameterised layer's forward and backward on
the MI355X training kernels (speechseparation_amd/train.py).

Reference flow per epoch: for every training clip `train_infer` (STFT -> BSRNN -> iSTFT, L1 tri-loss, SDR; m_dataset.py:182-226),
`toBackward.backward()` (the loss, or -SDR with --loss_sdr), an optimizer step every `batch_size` clips (AdamW lr 1e-3,
weight_decay 1e-2; :50), then a validation pass without gradients and the checkpoints `model.pth` (best validation loss) /
`model-always.pth` (+ `optimizer*.pth`).  Same flags: --datapath --mini --batch_size --resume --loss_sdr.

Data: <datapath>/{train,val}/<clip>/{mixture,speech}.wav - the folders the reference trains from (`samples(args.datapath, 'train')` /
`'val'`, train.py:58,74; DnR layout of m_dataset.samples(), :8-19), as 16-bit / float WAV (the reference's .flac needs torchaudio, absent
here); a `tr/` folder (DnR v2's name) is used when there is no `train/`; --train-folder / --val-folder override, or --synthetic N clips of seeded noise (no dataset ships with the reference).
Not carried over: wandb logging, the discriminator, the RIR / re-mix augmentations, ReduceLROnPlateau (commented out in the
reference loop) and its batch-size growth heuristic.  Under `torchrun` (one process per GPU) the clips are split over the ranks
and the gradients averaged before every optimizer step (data-parallel, RCCL).
"""
import argparse
import os
import random

import torch

from speechseparation_amd import audio, train as hip_train, weights
from speechseparation_amd.bsrnn import BSRNN


def dataset(datapath, folder):
    """[(mixture path, speech path)] of m_dataset.samples(datapath, folder)."""
    base = os.path.join(datapath, folder)
    if not os.path.isdir(base):
        return []
    out = []
    for name in sorted(os.listdir(base)):
        d = os.path.join(base, name)
        if os.path.isdir(d) and os.path.exists(os.path.join(d, "mixture.wav")) and os.path.exists(os.path.join(d, "speech.wav")):
            out.append((os.path.join(d, "mixture.wav"), os.path.join(d, "speech.wav")))
    return out


def load_pair(item, device):
    """(mix, speech) [2, n] on the device; mono files are duplicated to two rows (m_dataset.load_waveform, :57-61)."""
    if isinstance(item, int):                                   # synthetic clip number `item`
        n = SYNTH_SAMPLES
        return (torch.from_numpy(weights.synth_waveform(2, n, seed=1000 + 2 * item)).to(device),
                torch.from_numpy(weights.synth_waveform(2, n, seed=1001 + 2 * item)).to(device))
    pair = []
    for path in item:
        w, _ = audio.load_wav(path)
        pair.append((torch.cat((w, w), 0) if w.shape[0] == 1 else w[:2]).to(device))
    n = min(pair[0].shape[1], pair[1].shape[1])
    return pair[0][:, :n].contiguous(), pair[1][:, :n].contiguous()


SYNTH_SAMPLES = 4 * 16000


def main(argv=None):
    global SYNTH_SAMPLES
    ap = argparse.ArgumentParser(description="Train a BSRNN model")
    ap.add_argument("--datapath", type=str, default=None, help="Path to the dataset")
    ap.add_argument("--mini", action="store_true", help="Use a small dataset")
    ap.add_argument("--batch_size", type=int, default=1, help="Batch size")
    ap.add_argument("--resume", action="store_true", help="Reload model")
    ap.add_argument("--train-folder", type=str, default=None, help="training split under --datapath (default: train, else tr)")
    ap.add_argument("--val-folder", type=str, default="val", help="validation split under --datapath")
    ap.add_argument("--loss_sdr", action="store_true", help="Use SDR as loss")
    ap.add_argument("--epochs", type=int, default=1, help="epochs to run (the reference loops until interrupted)")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N", help="N synthetic training clips (and N // 4 + 1 for validation)")
    ap.add_argument("--seconds", type=float, default=4.0, help="length of the synthetic clips")
    ap.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED", help="initial weights from the seeded generator instead of torch's init")
    ap.add_argument("--graph", action="store_true", help="replay each iteration as one hipGraph (speechseparation_amd.train.GraphedTrainStep): "
                    "single process, --batch_size 1, one graph per clip length (at most 4 lengths, other clips run launch by launch)")
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--outdir", type=str, default=".")
    args = ap.parse_args(argv)
    SYNTH_SAMPLES = int(args.seconds * 16000)

    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = torch.device(args.device or "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    if args.synthetic:
        train_set, val_set = list(range(args.synthetic)), list(range(args.synthetic, args.synthetic + args.synthetic // 4 + 1))
    elif args.datapath:
        # the reference's live path reads `train` and `val` (train.py:58,74); DnR's own split is called `tr`
        tf = args.train_folder or ("train" if os.path.isdir(os.path.join(args.datapath, "train")) else "tr")
        train_set, val_set = dataset(args.datapath, tf), dataset(args.datapath, args.val_folder)
    else:
        ap.error("give --datapath DIR or --synthetic N")
    if args.mini:
        random.Random(0).shuffle(train_set)
        train_set, val_set = train_set[:10], val_set[:10]
    if not train_set:
        raise SystemExit("no training clips found under %s/{train,tr}/*/{mixture,speech}.wav (WAV copies of the reference's .flac; "
                         "--train-folder names another split)" % args.datapath)

    model = BSRNN().train()
    if args.synthetic_weights is not None:
        audio.load_model_weights(model, None, args.synthetic_weights)
    if args.resume:
        model.load_state_dict(torch.load(os.path.join(args.outdir, "model.pth"), weights_only=True))
    model = model.to(device)
    use_graph = args.graph and world == 1 and args.batch_size == 1
    if args.graph and not use_graph and rank == 0:
        print("--graph needs a single process and --batch_size 1: running launch by launch")
    optimizer = hip_train.AdamW(model.parameters(), lr=0.001, weight_decay=0.01, capturable=use_graph)
    graphs = {}                                                  # clip shape -> GraphedTrainStep
    if args.resume and os.path.exists(os.path.join(args.outdir, "optimizer.pth")):
        optimizer.load_state_dict(torch.load(os.path.join(args.outdir, "optimizer.pth"), weights_only=True))

    best = 1.0e30
    for epoch in range(args.epochs):
        if rank == 0:
            print("Epoch", epoch)
        order = list(range(len(train_set)))
        random.Random(epoch).shuffle(order)                      # DataLoader(shuffle=True)
        order = order[: len(order) // world * world][rank::world]     # the same number of clips (and collectives) on every rank
        epoch_loss, epoch_sdr, batch_i = 0.0, 0.0, 0
        for idx in order:
            mix, speech = load_pair(train_set[idx], device)
            if use_graph and (tuple(mix.shape) in graphs or len(graphs) < 4):
                if tuple(mix.shape) not in graphs:
                    graphs[tuple(mix.shape)] = hip_train.GraphedTrainStep(model, optimizer, mix.shape[0], mix.shape[1], loss_sdr=args.loss_sdr)
                step = graphs[tuple(mix.shape)]
                loss = step(mix, speech)                         # loss, backward, optimizer step, zero_grad: one graph launch
                batch_i += 1
                epoch_loss += float(loss)
                epoch_sdr += float(step.last_sdr)
                continue
            loss, x_time = hip_train.train_loss(model, mix, speech)
            sdr = hip_train.sdr(x_time, speech[:, :x_time.shape[1]])
            (-sdr if args.loss_sdr else loss).backward()
            batch_i += 1
            epoch_loss += float(loss.detach())
            epoch_sdr += float(sdr.detach())
            if batch_i % args.batch_size == 0:
                if world > 1:
                    from speechseparation_amd.dist import all_reduce_gradients
                    all_reduce_gradients(model.parameters())
                optimizer.step()
                optimizer.zero_grad()
        if world > 1:
            from speechseparation_amd.dist import all_reduce_gradients
            all_reduce_gradients(model.parameters())
        optimizer.step()                                         # train.py:122-123: the tail of the epoch
        optimizer.zero_grad()
        if rank == 0:
            print("Epoch", epoch, "Loss", epoch_loss / max(1, batch_i), "sdr", epoch_sdr / max(1, batch_i))

        # validation without gradients: the inference path (bsrnn_evaluate = infer + train_infer's arithmetic on the device)
        model.eval()
        val_loss = val_sdr = 0.0
        with torch.no_grad():
            for item in val_set:
                mix, speech = load_pair(item, device)
                mtr = model.evaluate(mix, speech)
                val_loss += mtr["loss"]
                val_sdr += mtr["sdr"]
        model.train()
        if rank == 0:
            if val_set:
                print("Validation Loss", val_loss / len(val_set), "Validation SDR", val_sdr / len(val_set))
            state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            if val_set and val_loss < best:
                print("...Saving")
                best = val_loss
                torch.save(state, os.path.join(args.outdir, "model.pth"))
                torch.save(optimizer.state_dict(), os.path.join(args.outdir, "optimizer.pth"))
            torch.save(state, os.path.join(args.outdir, "model-always.pth"))
            torch.save(optimizer.state_dict(), os.path.join(args.outdir, "optimizer-always.pth"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
