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


# ------------------------------------------------------------------------------------------ data-parallel training
# Training has the one real exchange step of this code base: every rank runs the step on its own clips (train.py:97-115 is
# batch_size 1 per iteration; DP = one clip per GPU and iteration) and the gradients are averaged before the optimizer step.
# 7.48 M parameters = 29.9 MB of fp32 gradients: they are flattened into a few large buckets (default 16 MB: xGMI rings are
# per-link bound, a handful of large all-reduces beats 285 small ones) in the model's parameter order, reduced with
# torch.distributed (backend nccl = RCCL on the GPUs, gloo in the CPU tests) and scattered back in place.
def gradient_buckets(params, bucket_bytes=16 << 20):
    """Lists of parameters (those with a gradient) whose gradients together fill about `bucket_bytes`, in parameter order."""
    buckets, cur, size = [], [], 0
    for p in params:
        if p.grad is None or p.numel() == 0:
            continue
        cur.append(p)
        size += p.numel() * p.grad.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    return buckets


def all_reduce_gradients(params, group=None, bucket_bytes=16 << 20, average=True):
    """Sum (or average) the .grad of `params` over the ranks, bucketed; every rank must hold gradients for the same
    parameters.  Returns the number of all-reduce calls made."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    calls = 0
    for bucket in gradient_buckets(list(params), bucket_bytes):
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        if flat.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal on one GPU / CPU collectives: staged through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        calls += 1
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return calls
