"""world_size-2 gloo test of the multi-GPU bookkeeping (CPU): row sharding, shard-exact input
generation, output gather and the max-over-ranks timing reduce.  The per-shard work here is
the oracle's STFT (row-wise like the real path); the HIP path itself is covered by -m gpu."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speechseparation_amd import weights
from speechseparation_amd.dist import gather_rows, max_over_ranks, separate_sharded, shard_rows


def test_shard_rows_partition():
    for total in (1, 5, 64, 512, 7):
        for world in (1, 2, 3, 8):
            blocks = [shard_rows(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard_rows(512, 8, 3) == (192, 256)          # config 4: 64 rows per GPU


def _worker(rank, world, port, total_rows, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import bsrnn_numpy as onp
    lo, hi = shard_rows(total_rows, world, rank)
    wave = torch.from_numpy(weights.synth_waveform(hi - lo, n, seed=1234, row_offset=lo))   # exactly this rank's rows
    fn = lambda w: torch.from_numpy(onp.stft_interleaved(w.numpy()))
    lo2, hi2, out = separate_sharded(fn, torch.from_numpy(weights.synth_waveform(total_rows, n, seed=1234)), world, rank)
    assert (lo, hi) == (lo2, hi2)
    assert torch.equal(out, fn(wave))                    # shard generated locally == slice of the global batch
    full = gather_rows(out, total_rows)
    t = max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        q.put((full.numpy(), t))
    dist.destroy_process_group()


def test_two_rank_gather_matches_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total_rows, n = 5, 3000                              # odd row count: shards of 3 and 2
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total_rows, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, t = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import bsrnn_numpy as onp
    ref = onp.stft_interleaved(weights.synth_waveform(total_rows, n, seed=1234))
    assert np.array_equal(full, ref)
    assert t == 2.0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two rank processes itself (the driver calls it
    exactly like that).  BSRNN_BENCH_PLUMBING=1: gloo rendezvous, barrier and max-reduce only - no kernels, no GPU."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["BSRNN_BENCH_PLUMBING"] = "1"
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 3
    assert d["max_over_ranks"] == 2.0                    # rank 1's value won the MAX reduce
    assert d["per_rank"] == [1.0, 2.0]                   # every rank's own figure reaches rank 0 (all_gather)


def test_bench_refuses_a_world_that_contradicts_gpus():
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSRNN_BENCH_PLUMBING="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechseparation_amd.dist import all_reduce_gradients
    torch.manual_seed(0)
    shapes = [(256, 64), (256,), (0,), (768, 768), (64, 128), (2, 2)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    gen = torch.Generator().manual_seed(100 + rank)
    for p in params:
        p.grad = torch.randn(p.shape, generator=gen)
    calls = all_reduce_gradients(params, bucket_bytes=1 << 20)   # 2.4 MB of gradients -> 2 buckets
    if rank == 0:
        q.put(([p.grad.numpy().copy() for p in params], calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce_averages_in_buckets():
    """Data-parallel training's exchange step (dist.all_reduce_gradients) on two gloo ranks: bucketed, averaged, in place."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    grads, calls = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shapes = [(256, 64), (256,), (0,), (768, 768), (64, 128), (2, 2)]
    want = []
    for r in range(2):
        gen = torch.Generator().manual_seed(100 + r)
        want.append([torch.randn(s, generator=gen) for s in shapes])
    assert calls == 2
    for g, a, b in zip(grads, want[0], want[1]):
        assert np.allclose(g, ((a + b) / 2).numpy(), rtol=0, atol=1e-7)
