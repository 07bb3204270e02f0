"""Data-parallel sharding of utterance rows over the GPUs of one node.

The separation path has no exchange step (every op of BSRNN.forward is row-wise in dim 0,
bsrnn.py:394-395 with merge_channels=False), so multi-GPU is plain
row sharding: one process per GPU, rank g of G owns a contiguous block of rows, weights are
replicated (every rank builds the same state_dict; 29.9 MB), and torch.distributed (RCCL on
the GPUs, gloo in CPU tests) is used only OUTSIDE the path: start/stop barriers, max-over-ranks
timing, and an optional all-gather of the outputs for parity checks.
"""
import torch


def shard_rows(total_rows, world_size, rank):
    """Contiguous block [lo, hi) of `total_rows` for `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(total_rows, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def separate_sharded(separate_fn, waveform, world_size, rank):
    """Run `separate_fn` on this rank's rows of the global [R, n] batch; returns (lo, hi, out)."""
    lo, hi = shard_rows(waveform.shape[0], world_size, rank)
    return lo, hi, separate_fn(waveform[lo:hi].contiguous())


def gather_rows(local_out, total_rows, group=None):
    """All-gather row shards back into the global [R, ...] tensor (off the timed path)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_rows(total_rows, world, r) for r in range(world)]
    maxrows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxrows,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
    pad[:local_out.shape[0]] = local_out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def max_over_ranks(seconds, device="cpu", group=None):
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
