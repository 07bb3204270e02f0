"""Multi-rank rehearsal on the one-GPU box: `bench.py --gpus 2` with BSRNN_BENCH_REHEARSE=1 starts two rank
processes that share cuda:0 and talk over gloo, so the launcher, the per-rank sharding, the kernels and the collectives
around the timed region all run in the driver's GPU tier.  Not a measurement (both ranks share one GPU); a scaling curve
needs the driver's 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_rank_rehearsal_on_one_gpu():
    env = dict(os.environ, BSRNN_BENCH_REHEARSE="1", PYTHONPATH=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rows", "8", "--samples", "32000",
           "--no-cpu-baseline", "--no-exact-f32", "--no-train-step"]
    r = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert d["config"]["global_rows"] == 16 and d["config"]["rows_per_gpu"] == 8 and d["scaling"] == "weak"
    assert len(d["per_rank_ms"]) == 2 and all(v > 0 for v in d["per_rank_ms"])
    assert d["per_rank_ms_min_max"] == [min(d["per_rank_ms"]), max(d["per_rank_ms"])]
    assert abs(d["ms_per_step"] - max(d["per_rank_ms"])) < 1e-3     # value = the slowest rank's time
    T = 1 + 32000 // 1024
    assert abs(d["value"] - 16 * T / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert "REHEARSAL" in d["data"]
    assert "roofline" in d and "cpu_baseline" not in d
