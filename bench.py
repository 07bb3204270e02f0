#!/usr/bin/env python3
"""Headline benchmark: separated row-frames/s (+ RTF) of the offline separation sandwich
waveform -> STFT -> BSRNN.forward -> iSTFT -> waveform on synthetic mixtures.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric config): per GPU R = 64 rows x 8 s @ 16 kHz
(128 000 samples, T = 126 frames), K = 12 bands, fp32, weights and waveforms from the
deterministic generators (there are no trained weights / datasets).  Inputs are resident in
HBM before the timed region.  N > 1: one process per GPU (torch.distributed, RCCL), every rank
separates its own 64-row shard of a 64*N-row batch -- the path has no exchange step, so the
only collectives are the timing barrier / max-reduce ("weak" scaling).

One JSON line on rank 0 with the driver's contract fields plus
  roofline      the dominant kernel family (by HIP-event time measured live in the timed
                region, on the launch stream): achieved algorithmic rate vs gfx950 peak
  cpu_baseline  the reference's CPU path (stock torch CPU ops, oracle/bsrnn_torch_cpu.py) timed
                on this box's host cores on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# work-unit constants per row-frame (SURVEY.md section 8(d) / BASELINE.md section 4), K = 12, fp32
BYTES_DUAL_PATH = 24576          # 4 blocks x (read + write of 12 x 64 fp32)
BYTES_PIPELINE = 96312           # every stage reads its input once, writes its output once
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* (f32 in), dense
DOMINANT = ("bandsplit_mlp", "mask_mlp")
# what the arithmetic type really is, per BSRNN_GEMM mode (the LSTMs follow BSRNN_LSTM: fp16x2 unless "f32")
DTYPE_LABEL = {
    "f32": "f32 (exact fp32 MFMA)",
    "fp16x2": "f32 io/accumulate, fp16x2-split operands (22-bit operands, 3 f16 MFMA terms; a2*b2 term dropped)",
    "fp16": "f16 operands, f32 accumulate/io (REDUCED PRECISION configuration; the fp16 one-term mode: 3 more mantissa bits than the bf16 BASELINE config 2 names, range guard)",
    "bf16": "bf16 operands in the per-band MLPs, f32 accumulate/io, LSTMs fp16x2 (REDUCED PRECISION configuration: BASELINE config 2 as named)",
}


def exact_macs(v):
    """MACs per row-frame of each stage family for band table v (as the kernels execute them:
    fc_in folded into W_ih of layer 0)."""
    H = 64
    K = len(v)
    pre = sum(2 * (2 * w) ** 2 for w in v if w)
    fc = sum(2 * w * max(2 * w, H) + max(2 * w, H) * H + H * H for w in v if w)
    back = sum(H * 2 * H + 2 * H * max(2 * w, 2 * H) + max(2 * w, 2 * H) * 2 * w for w in v if w)
    post = pre
    band = 2 * K * 2 * (4 * H * (H + H) + 4 * H * (2 * H + H))           # 2 blocks x K steps x 2 dirs x (layer0 + layer1)
    time_ = 2 * K * 2 * 4 * H * (H + H)                                   # 2 blocks x K seqs x 2 layers
    return {"bandsplit_mlp": pre + fc, "mask_mlp": back + post, "band_lstm": band, "band_fc": 2 * K * 2 * H * H,
            "time_lstm": time_, "time_fc": 2 * K * H * H}


def mlp_flow(gemm):
    """'fused' (mlp_chain.hip: one launch per chain, intermediates in LDS) unless the per-layer flow was asked for
    (BSRNN_MLP=layers) or the exact-fp32 mode runs (per-layer fp32 MFMA kernels)."""
    return "layers" if (gemm == "f32" or os.environ.get("BSRNN_MLP") == "layers") else "fused"


def gemm_activation_bytes(v, flow="layers"):
    """Algorithmic activation bytes per row-frame of the MLP launches (fp32: every launch reads its input once and writes
    its output once).  Per-layer flow: ten launches, the last one also reads the residual and the spectrum it multiplies.
    Fused flow: BandSplit reads the spectrum and writes the residual P and Z; MaskEstimation reads Z, P and the spectrum
    and writes the masked spectrum (8200 + 8200 + 3072, and 3072 + 8200 + 8200 + 8200 B at K = 12)."""
    H = 64
    a = [2 * w for w in v if w]
    m = [max(x, H) for x in a]
    p = [max(x, 2 * H) for x in a]
    n = len(a)
    if flow == "fused":
        return 4 * ((sum(a) + sum(a) + n * H) + (n * H + 3 * sum(a)))
    cols = (sum(a) + sum(a)) + (sum(a) + sum(a)) + (sum(a) + sum(m)) + (sum(m) + n * H) + (n * H + n * H)        # PRE0 PRE2 FC0 FC2 FC4
    cols += (n * H + n * 2 * H) + (n * 2 * H + sum(p)) + (sum(p) + sum(a)) + (sum(a) + sum(a)) + (sum(a) + 3 * sum(a))   # BACK0 BACK2 BACK4 POST0 POST2
    return 4 * cols


def host_cores():
    """Threads for the CPU baseline (the ONE policy): this process's share of the box = CPU affinity
    capped by the cgroup quota and by 16, the CPU share of one GPU slot on the bench pool.  (`os.cpu_count()` reports the host's 256 hardware threads from inside a 16-CPU cgroup;
    asking torch for all of them thrashes.)  BSRNN_CPU_THREADS overrides."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    n = min(n, 16)              # the bench pool's CPU share of one GPU slot (what gpurun grants a one-GPU box), also where no quota is visible
    if os.environ.get("BSRNN_CPU_THREADS"):
        n = int(os.environ["BSRNN_CPU_THREADS"])
    return max(1, n)


# The ten grouped-GEMM launches of the two bracketed stages, by kernel instance (template arguments as rocprofv3
# prints them) -> launches per step.  gemm_h2_kernel<epilogue, tile (1 = 128x64, 2 = 128x128), ablation>;
# gemm_split_kernel<epilogue, pieces, tile, ...>.
FUSED_LAUNCH_MIX = {        # mlp_chain_kernel<chain (0 BandSplit, 1 MaskEstimation), MFMA terms>
    "fp16x2": (("mlp_chain_kernel<0, 3>", 1), ("mlp_chain_kernel<1, 3>", 1)),
    "fp16": (("mlp_chain_kernel<0, 1>", 1), ("mlp_chain_kernel<1, 1>", 1)),
    "bf16": (("mlp_chain_kernel<0, -1>", 1), ("mlp_chain_kernel<1, -1>", 1)),
}
GEMM_LAUNCH_MIX = {
    "f32": (("gemm_f32_kernel<1,", 8), ("gemm_f32_kernel<0,", 1), ("gemm_f32_kernel<3,", 1)),
    "fp16x2": (("gemm_h2_kernel<1, 2,", 6), ("gemm_h2_kernel<1, 1,", 2), ("gemm_h2_kernel<0, 1,", 1), ("gemm_h2_kernel<3, 2,", 1)),
    "fp16": (("gemm_h2_kernel<1, 2, 0, 1>", 6), ("gemm_h2_kernel<1, 1, 0, 1>", 2), ("gemm_h2_kernel<0, 1, 0, 1>", 1), ("gemm_h2_kernel<3, 2, 0, 1>", 1)),
    "bf16": (("gemm_h2_kernel<1, 2,", 6), ("gemm_h2_kernel<1, 1,", 2), ("gemm_h2_kernel<0, 1,", 1), ("gemm_h2_kernel<3, 2,", 1)),
}
# matrix-pipe roofline of the grouped GEMM per mode: (kernel, peak in algorithmic TFLOP/s, how it is derived)
GEMM_ROOF = {
    "f32": ("gemm_f32_kernel", 157.3, "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32, exact fp32)"),
    "fp16x2": ("gemm_h2_kernel", 2500.0 / 3, "f16 MFMA dense peak 2500 TFLOP/s / 3 MFMA terms per fp32-accurate product "
               "(a1b1 + a1b2 + a2b1, fp32 accumulate)"),
    "fp16": ("gemm_h2_kernel<fp16, 1 term>", 2500.0, "f16 MFMA dense peak; REDUCED PRECISION (plain fp16 operands, ~5e-4 relative per product): "
             "not the fp32-accurate default"),
    "bf16": ("mlp_chain_kernel<bf16, 1 term>", 2500.0, "bf16 MFMA dense peak; REDUCED PRECISION (plain bf16 operands, ~4e-3 relative per product): "
             "not the fp32-accurate default"),
}


def profiled_traffic(gemm, flow="layers"):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN_traffic.json, written by tools/summarize_profile.py from separate FETCH_SIZE and
    WRITE_SIZE passes with the gfx950 x2 fetch correction).  The bracketed stages launch the LEAKY
    epilogue 8x, LINEAR 1x and MASK 1x per step."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot = 0.0
    mix = FUSED_LAUNCH_MIX[gemm] if flow == "fused" else GEMM_LAUNCH_MIX[gemm]
    for key, n in mix:
        hit = [v for k, v in d.items() if k.startswith(key)]
        if not hit:
            return None, None
        tot += n * (hit[0]["fetched_bytes"] + hit[0]["written_bytes"])
    return tot / sum(n for _, n in mix), os.path.basename(files[-1])


def build_model(device):
    from speechseparation_amd import weights
    from speechseparation_amd.bsrnn import BSRNN
    sd = weights.synth_state_dict(seed=0)
    m = BSRNN().eval()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m.to(device), sd


def cpu_baseline(sd, rows, n_samples, budget_s=12.0, max_iters=7):
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, weights
    cores = host_cores()
    torch.set_num_threads(cores)
    m = TorchCpuBSRNN(sd, spec.generate_bandsplits()[0])
    wave = torch.from_numpy(weights.synth_waveform(rows, n_samples, seed=1234))
    m.separate(wave)                      # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < max_iters and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        m.separate(wave)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    T = 1 + n_samples // 1024
    return {"value": rows * T / med, "unit": "row-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d iteration(s) of the full workload (%d rows x %d samples, T=%d), median; stock torch CPU ops "
                      "(nn.LSTM/MKLDNN, addmm, torch.stft/istft) = the reference's CPU path restated" % (len(times), rows, n_samples, T),
            "seconds_per_pass": med}


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes (one per GPU) and relay
    rank 0's JSON line.  Runs before this process has made any HIP call (torch.cuda.device_count() does not initialise
    the runtime on this image); the children are ordinary subprocesses, nothing is exec'ed in place."""
    import subprocess
    rehearse = os.environ.get("BSRNN_BENCH_REHEARSE") == "1" or os.environ.get("BSRNN_BENCH_PLUMBING") == "1"
    if not rehearse:
        have = torch.cuda.device_count()
        if n > have:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) visible (BSRNN_BENCH_REHEARSE=1 shares cuda:0 for a rehearsal)" % (n, have))
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    for ln in out0.splitlines():              # stdout carries the JSON line only; anything a library printed there goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.exit("bench.py: rank(s) failed: %s" % ", ".join("rank %d rc %d" % b for b in bad))


def exact_f32_record(args):
    """The same workload on the library's exact-fp32 kernels (v_mfma_f32_*_f32, bit-exact fp32 fma chains), measured in a
    child process started before this one touches the GPU (the compute mode is read once per process)."""
    import subprocess
    env = dict(os.environ, BSRNN_GEMM="f32", BSRNN_LSTM="f32", BSRNN_BENCH_CHILD="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(10, args.steps // 2)), "--warmup", str(args.warmup),
           "--rows", str(args.rows), "--samples", str(args.samples), "--no-cpu-baseline", "--no-exact-f32", "--no-train-step", "--no-in-flight"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        return {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "compute": {k: d["compute"][k] for k in ("gemm", "lstm")},
                "note": "BSRNN_GEMM=f32 BSRNN_LSTM=f32: exact fp32 MFMA kernels of the same library, same workload, separate process"}
    except Exception as e:      # the sub-record is a report item, not part of `value`
        return {"error": "%s: %s" % (type(e).__name__, e)}


def train_step_record(args):
    """One iteration of the reference's training loop (train.py:97-115) on the library's training kernels for the same batch
    shape, measured in a child process before this one touches the GPU (tools/train_step_bench.py).
    A report item, not part of `value`."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "tools", "train_step_bench.py"), "--rows", str(args.rows), "--seconds", str(args.samples / 16000.0),
           "--steps", "10", "--no-cpu"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        return {"ms_per_step": d["ms_per_step"], "row_frames_per_s": d["row_frames_per_s"], "phases_ms": d.get("phases_ms"),
                "graph_ms_per_step": d.get("graph_ms_per_step"), "graph_row_frames_per_s": d.get("graph_row_frames_per_s"),
                "note": "forward + L1 tri-loss + backward + AdamW of m_dataset.train_infer / train.py:97-115 on the exact-fp32 training kernels, "
                        "gradients verified against torch.autograd (tests/test_gpu_train.py); separate process; ms_per_step = the loop driven "
                        "from Python (one launch per kernel), graph_ms_per_step = the same iteration captured once and replayed as one hipGraph "
                        "(train.GraphedTrainStep)"}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def plumbing_line(args, world, dist):
    """BSRNN_BENCH_PLUMBING=1 (CPU test of the launcher and the collectives around the timed region; no kernels run,
    nothing is measured): rendezvous over gloo, barrier, max-reduce, one JSON line from rank 0."""
    t = torch.tensor([1.0 + int(os.environ.get("RANK", "0"))], dtype=torch.float64)
    rccl_ranks = 1
    per_rank = [float(t.item())]
    if dist is not None:
        dist.barrier()
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)                  # the same collectives, in the same order, as the measured path
        per_rank = [float(a.item()) for a in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rccl_ranks = dist.get_world_size()
        if rccl_ranks != args.gpus:
            sys.exit("bench.py: the process group has %d ranks, --gpus says %d" % (rccl_ranks, args.gpus))
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"metric": "separated row-frames/sec, batch64 8s@16kHz", "value": None, "unit": "row-frames/s", "n_gpus": world,
                          "rccl_ranks": rccl_ranks, "backend": "gloo", "steps": args.steps, "warmup": args.warmup, "max_over_ranks": float(t.item()),
                          "per_rank": per_rank,
                          "data": "PLUMBING TEST: no kernels run, nothing measured"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=64, help="rows per GPU")
    ap.add_argument("--samples", type=int, default=128000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the exact-fp32 sub-record (a child process)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step sub-record (a child process)")
    ap.add_argument("--no-in-flight", action="store_true", help="skip the two / three batches in flight report item")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher: become the launcher (N rank processes), before any GPU call
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let bench.py start the ranks itself)" % (args.gpus, world))
    dist = None
    # Rehearsal on a one-GPU box (not a measurement): BSRNN_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo for
    # the barrier / max-reduce, so the whole multi-rank script path can be exercised without an 8-GPU node.
    rehearse = os.environ.get("BSRNN_BENCH_REHEARSE") == "1"
    plumbing = os.environ.get("BSRNN_BENCH_PLUMBING") == "1"
    if rehearse:
        local_rank = 0
    # exact-fp32 sub-record: a child process, run to completion before this process initialises the GPU
    f32_rec = train_rec = None
    if world == 1 and not args.no_exact_f32 and not plumbing:
        f32_rec = exact_f32_record(args)
    if world == 1 and not args.no_train_step and not plumbing:
        train_rec = train_step_record(args)           # (child processes run before this one touches the GPU)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse or plumbing:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if plumbing:
        plumbing_line(args, world, dist)
        return
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.set_grad_enabled(False)

    from speechseparation_amd import spec, weights
    from speechseparation_amd.dist import shard_rows
    model, sd = build_model(device)
    lo, hi = shard_rows(args.rows * world, world, rank)          # contiguous row block of the global batch
    wave = torch.from_numpy(weights.synth_waveform(hi - lo, args.samples, seed=1234, row_offset=lo)).to(device)
    T = spec.n_frames(args.samples)
    out = torch.empty((hi - lo, (T - 1) * 1024), device=device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Device ramp, part of setup like the model build: the first ~40 ms of work after an idle GPU run 3-4 % slower
    # (clocks and caches; measured on one box: 1.235 ms/step with 3 warm-up steps, 1.195 with 30), so a fixed number
    # of untimed steps runs before the caller's W warm-up steps whatever W is.  The timed region itself carries a
    # fixed ~0.4 ms (pipeline fill after the barrier, final synchronize): 1.22 ms/step at K = 20, 1.20 at K = 100.
    PREWARM = 64
    # The timed loop opts out of the default range policy ("exact": every call waits for its kernels and checks the fp16x2
    # range guard): K back-to-back asynchronous calls are what is measured.  The guard word is read once after
    # the run (model.sync()): a violation would fail the run instead of going unnoticed.
    model.set_range_policy("deferred")
    for _ in range(PREWARM):
        model.separate(wave, out=out)
    for _ in range(args.warmup):
        model.separate(wave, out=out)
    # timed region: only the dominant kernel family (the grouped fp32-MFMA GEMM launches of the
    # two per-band MLP chains) is bracketed with HIP events, on the launch stream
    # (every SAMPLE-th step of the timed region carries the brackets: an event pair costs the stream ~2.5 us, tools/bracket_cost.py;
    #  the durations are averaged over the bracketed launches only)
    SAMPLE = 4
    model.set_profiling(False, device)
    model.stage_times(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % SAMPLE == 0:
            model.set_profiling(DOMINANT, device)
        elif i % SAMPLE == 1:
            model.set_profiling(False, device)
        model.separate(wave, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    dom_stages = model.stage_times(reset=True)
    n_sampled = (args.steps + SAMPLE - 1) // SAMPLE
    # per-stage table: a few extra steps with every stage bracketed (not part of `value`)
    model.set_profiling(True, device)
    n_extra = 5
    for _ in range(n_extra):
        model.separate(wave, out=out)
    stages = model.stage_times(reset=True)
    model.set_profiling(False, device)
    model.sync()                                  # raises if any step of the run left the fp16x2 range
    # With the dual path overlapped (the default for this workload) the mask chain is LAUNCHED beside the second time-axis launch and
    # its workgroups wait for their frames: its launch duration then contains that wait.  The same kernels one after the other
    # (a second context under BSRNN_OVERLAP=0, same weights and input): what the kernels take when nothing runs beside them.
    alone = None
    if model.overlap_state(device) == 1:
        keep = os.environ.get("BSRNN_OVERLAP")
        os.environ["BSRNN_OVERLAP"] = "0"
        try:
            m2, _ = build_model(device)
            m2.set_range_policy("deferred")
            for _ in range(8):
                m2.separate(wave, out=out)
            m2.set_profiling(DOMINANT, device)
            m2.stage_times(reset=True)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            n_alone = 20
            for _ in range(n_alone):
                m2.separate(wave, out=out)
            torch.cuda.synchronize()
            alone_ms = 1e3 * (time.perf_counter() - ta) / n_alone
            st2 = m2.stage_times(reset=True)
            m2.set_profiling(True, device)        # ... and the per-stage table of that flow (every stage bracketed)
            for _ in range(n_extra):
                m2.separate(wave, out=out)
            stages_alone = m2.stage_times(reset=True)
            m2.set_profiling(False, device)
            m2.sync()
            alone = {"ms_per_step": round(alone_ms, 4), "stages_ms": {k: round(st2[k][0] / n_alone, 4) for k in DOMINANT}, "all": stages_alone}
            del m2
        finally:
            if keep is None:
                os.environ.pop("BSRNN_OVERLAP", None)
            else:
                os.environ["BSRNN_OVERLAP"] = keep
    # Two and three independent batches in flight: further contexts (same weights, same input, own outputs) on streams of their own, the
    # calls alternating - how a server with more than one request queue would drive the library.  A report item beside `value` (which is
    # ONE context on ONE stream): launches of different batches fill each other's tails and the CUs a time-axis launch leaves idle.
    in_flight = None
    if world == 1 and not args.no_in_flight:
        ctxs = [model] + [build_model(device)[0] for _ in range(2)]
        strs = [torch.cuda.current_stream(device)] + [torch.cuda.Stream(device) for _ in range(2)]
        outs = [out] + [torch.empty_like(out) for _ in range(2)]
        for m_ in ctxs[1:]:
            m_.set_range_policy("deferred")
        in_flight = {}
        for nctx in (2, 3):
            def loop(k):
                for i in range(k):
                    j = i % nctx
                    with torch.cuda.stream(strs[j]):
                        ctxs[j].separate(wave, out=outs[j])
            loop(12 * nctx)
            torch.cuda.synchronize()
            tq = time.perf_counter()
            n_if = max(30, args.steps)
            loop(n_if)
            torch.cuda.synchronize()
            ms_if = 1e3 * (time.perf_counter() - tq) / n_if
            for j in range(nctx):
                with torch.cuda.stream(strs[j]):
                    ctxs[j].sync()                # (raises on a range-guard violation)
            in_flight[str(nctx)] = {"ms_per_step": round(ms_if, 4), "row_frames_per_s": round((hi - lo) * T / (ms_if * 1e-3), 1), "steps": n_if,
                                    "outputs_equal": bool(all(torch.equal(outs[0], o) for o in outs[1:nctx]))}
        del ctxs, outs
    # the reference's own operator on the same batch: BSRNN.forward on [R, 2050, T] (two layout transposes that `separate`
    # does not pay, no STFT / iSTFT) - a report item beside `value`
    xspec = model.stft(wave)
    for _ in range(5):
        model(xspec)
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    n_fwd = max(5, args.steps // 4)
    for _ in range(n_fwd):
        model(xspec)
    torch.cuda.synchronize()
    fwd_ms = 1e3 * (time.perf_counter() - tf0) / n_fwd
    model.sync()
    del xspec
    # the same call under the default policy (synchronise + guard check per call): what a caller who does not opt out sees
    model.set_range_policy("exact")
    torch.cuda.synchronize()
    te0 = time.perf_counter()
    n_ex = max(5, args.steps // 4)
    for _ in range(n_ex):
        model.separate(wave, out=out)
    torch.cuda.synchronize()
    exact_ms = 1e3 * (time.perf_counter() - te0) / n_ex
    my_ms = 1e3 * elapsed / args.steps
    per_rank_ms = [my_ms]
    if dist is not None:
        cpu_or_dev = "cpu" if rehearse else device
        t = torch.tensor([elapsed], device=cpu_or_dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [1e3 * float(a.item()) / args.steps for a in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if dist.get_world_size() != args.gpus:
            sys.exit("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    frames_total = args.rows * world * T * args.steps
    value = frames_total / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        rf = (hi - lo) * T                                   # row-frames one launch processes on this GPU
        macs = exact_macs(spec.generate_bandsplits()[0])
        stages_overlapped = None
        if alone is not None:                     # the per-stage table describes the kernels: taken from the flow without overlap
            stages_overlapped = {k: round(ms / n_extra, 4) for k, (ms, n) in stages.items() if n}
            stages = alone["all"]
        per_step = {k: (ms / n_extra) for k, (ms, n) in stages.items() if n}
        fam = {}
        for name, ms in per_step.items():
            fam[name] = {"ms_per_step": round(ms, 4)}
            if name in macs:
                fam[name]["tflops"] = round(2 * macs[name] * rf / (ms * 1e-3) / 1e12, 2)
        # dominant kernel = gemm_f32_kernel; its 10 launches per step inside the two MLP chains are
        # what the timed-region events bracket (5 launches per bracket)
        dom_ms_step = sum(dom_stages[k][0] for k in DOMINANT) / n_sampled
        dom_flop_step = 2 * sum(macs[k] for k in DOMINANT) * rf
        from speechseparation_amd import _native
        cmode = _native.compute_mode()
        flow = mlp_flow(cmode["gemm"])
        # launches of the dominant family per step: every bracket of the two MLP stages holds 1 (fused) or 5 (per-layer)
        # launches, and bsrnn_separate runs the batch as `blocks` concurrent row blocks (2 from 64 rows on), each with its own
        n_brackets = sum(dom_stages[k][1] for k in DOMINANT) / float(n_sampled)
        blocks = max(1, int(round(n_brackets / 2.0)))
        n_launch = int(round(n_brackets)) * (1 if flow == "fused" else 5)
        kname, peak, basis = GEMM_ROOF[cmode["gemm"]]
        if flow == "fused":
            kname = "mlp_chain_kernel"
        traffic, traffic_src = profiled_traffic(cmode["gemm"], flow)
        achieved = dom_flop_step / (dom_ms_step * 1e-3) / 1e12
        act_bytes = gemm_activation_bytes(spec.generate_bandsplits()[0], flow)
        roofline = {"kernel": "%s (the per-band MLP chains BandSplit + MaskEstimation, %s: %d launches/step)" % (
                        kname, "fused, intermediates in LDS" if flow == "fused" else "one grouped launch per layer", n_launch) +
                              (" over %d concurrent row blocks of %d rows" % (blocks, (hi - lo) // blocks) if blocks > 1 else ""),
                    "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "row_blocks": blocks, "traffic_source": traffic_src, "avg_launch_ms": round(dom_ms_step / n_launch, 4), "launches_per_step": n_launch,
                    "flop_per_launch_avg": dom_flop_step / n_launch, "peak_basis": basis,
                    "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "hbm_view": {"algorithmic_bytes_per_launch_avg": act_bytes * rf / n_launch,
                                 "achieved_GBs": round(act_bytes * rf / (dom_ms_step * 1e-3) / 1e9, 1),
                                 "frac_of_hbm_peak": round(act_bytes * rf / (dom_ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "note": "the same launches against HBM: activations in + out per launch, weights excluded"},
                    "note": "achieved = algorithmic flops (2 x MACs of the Linear layers x row-frames, fp32 semantics) / launch time; "
                            "HIP events on the launch stream inside the timed region, on every %d-th of its steps (%d launches timed)" % (SAMPLE, n_sampled * n_launch)}
        roofline["dual_path_flow"] = {0: "one launch after the other (BSRNN_OVERLAP=0)", 1: "overlapped: second band block beside the first time-axis launch, mask chain beside the second (auxiliary stream, per-workgroup waits)",
                                      2: "overlap switched off after a consumer's wait expired"}.get(model.overlap_state(device), "?")
        roofline["per_stage_ms"] = {k: round(dom_stages[k][0] / n_sampled, 4) for k in DOMINANT}
        if alone is not None:
            a_ms = sum(alone["stages_ms"].values())
            roofline["kernels_alone"] = {"ms_per_step_of_that_flow": alone["ms_per_step"], "stages_ms": alone["stages_ms"],
                                         "achieved": round(dom_flop_step / (a_ms * 1e-3) / 1e12, 2), "frac": round(dom_flop_step / (a_ms * 1e-3) / 1e12 / peak, 4),
                                         "note": "the same two launches with nothing beside them (second context, BSRNN_OVERLAP=0, 20 bracketed steps): in the overlapped flow "
                                                 "the mask chain's launch starts beside the second time-axis launch and its duration contains its workgroups' waits for their frames"}
        dp_ms = sum(per_step.get(k, 0.0) for k in ("band_lstm", "band_fc", "time_lstm", "time_fc"))
        dp_gbs = BYTES_DUAL_PATH * rf / (dp_ms * 1e-3) / 1e9 if dp_ms else 0.0
        dual = {"bound": "hbm", "achieved": round(dp_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(dp_gbs / PEAK_HBM_GBS, 4), "ms_per_step": round(dp_ms, 4),
                "tflops": round(2 * sum(macs[k] for k in ("band_lstm", "band_fc", "time_lstm", "time_fc")) * rf / (dp_ms * 1e-3) / 1e12, 2) if dp_ms else 0.0,
                "note": "north-star accounting of the dual-path step: 24576 algorithmic B/row-frame; the step is fp32-compute/latency bound (SURVEY 7.3-1)"}
        line = {
            "metric": "separated row-frames/sec, batch64 8s@16kHz",
            "value": round(value, 1), "unit": "row-frames/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist is not None else 1, "backend": (dist.get_backend() if dist is not None else "none"),
            "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM,
            "ms_per_step": round(ms_per_step, 4), "per_rank_ms": [round(v, 4) for v in per_rank_ms],
            "per_rank_ms_min_max": [round(min(per_rank_ms), 4), round(max(per_rank_ms), 4)],
            "per_rank_ms_max_over_min": round(max(per_rank_ms) / min(per_rank_ms), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "range_guard": "deferred in the timed loop (explicit opt-out of the default per-call synchronise-and-check, include/bsrnn_hip.h); "
                           "guard word read after the run: clean.  The same call under the default policy: %.4f ms per call" % exact_ms,
            "forward_api": {"ms_per_call": round(fwd_ms, 4), "row_frames_per_s": round((hi - lo) * T / (fwd_ms * 1e-3), 1),
                            "note": "BSRNN.forward on the [R, 2050, T] spectrum (the reference's operator, bsrnn.py:385): model only, with the two "
                                    "layout transposes of that boundary; no STFT / iSTFT; this rank's %d rows" % (hi - lo)},
            "dtype": DTYPE_LABEL[cmode["gemm"]], "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks share cuda:0 over gloo; not a measurement)",
            "config": {"workload": "offline separate (STFT->BSRNN.forward->iSTFT), %d rows/GPU x %d samples @16 kHz (T=%d), K=12 bands, fp32"
                                   % (args.rows, args.samples, T),
                       "rows_per_gpu": args.rows, "global_rows": args.rows * world, "frames": T, "parallelism": "dp%d (row shards, no in-path collective)" % world},
            "rtf": round(elapsed / args.steps / (args.rows * args.samples / 16000.0), 8),
            "pipeline_hbm": {"achieved_GBs": round(BYTES_PIPELINE * rf / (ms_per_step * 1e-3) / 1e9, 1), "peak_GBs": PEAK_HBM_GBS},
            "compute": dict(cmode, note="fp32 in/out/state/accumulate; products of fp32 operands split into 16-bit pieces on the "
                                        "f16/bf16 matrix cores unless 'f32' (error at fp32 rounding level, DESIGN.md)"),
            "roofline": roofline, "roofline_dual_path": dual, "stages": fam,
        }
        if stages_overlapped is not None:
            line["stages_note"] = ("`stages` and `roofline_dual_path` are the launches one after the other (second context, BSRNN_OVERLAP=0: %.4f ms per step); "
                                   "in the overlapped flow that `value` measures a consumer launch's bracket contains its waits" % alone["ms_per_step"])
            line["stages_overlapped_flow_ms"] = stages_overlapped
        if in_flight is not None:
            line["batches_in_flight"] = dict(in_flight, note="NOT `value`: the same step for 2 / 3 independent batches of %d rows in flight (one context and stream "
                                             "each, calls alternating; ms per step = per batch); `value` is one context on one stream" % (hi - lo))
        if f32_rec is not None:
            line["exact_f32"] = f32_rec
        if train_rec is not None:
            line["train_step"] = train_rec
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(sd, args.rows, args.samples)
            line["cpu_baseline"] = cb
            line["gpu_over_cpu"] = round(value / cb["value"], 1)
            # BASELINE config 1 (infer.py on one 4 s mono 16 kHz mixture: 2 rows x 64 000 samples, T = 63), same policy
            c1 = cpu_baseline(sd, 2, 64000, budget_s=4.0, max_iters=21)
            line["cpu_baseline_config1"] = {k: c1[k] for k in ("value", "unit", "cores", "kind", "sample", "seconds_per_pass")}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
