"""GPU: training step, part 1 - the recurrent layers' forward-with-saves and backward through time (csrc/lstm_train.hip,
bsrnn_lstm_train_forward / _backward) against torch.autograd on stock nn.LSTM on the CPU, the operator the reference's
train step differentiates (bsrnn.py:66-72 inside train.py:97-115).  Exact-fp32 kernels: outputs to 2e-6, gradients to 1e-4
of their largest element (sums over up to N*L = 3 000 rows in a different order than torch's)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def _reference(N, L, IN, bidir, seed, gain=1.0):
    torch.manual_seed(seed)
    lstm = torch.nn.LSTM(IN, 64, num_layers=1, batch_first=True, bidirectional=bidir)
    with torch.no_grad():
        for p in lstm.parameters():
            p.mul_(gain)
    x = torch.randn(N, L, IN, requires_grad=True)
    dh = torch.randn(N, L, 128 if bidir else 64)
    h, _ = lstm(x)
    (h * dh).sum().backward()
    return lstm, x, dh, h


@pytest.mark.parametrize("N,L,IN,bidir,gain", [
    (37, 12, 64, True, 1.0),       # band-axis layer 0: 12 bands per sequence, partial tile of sequences
    (64, 12, 128, True, 3.0),      # band-axis layer 1 (input = both directions of layer 0), saturating gates
    (24, 126, 64, False, 1.0),     # time-axis layer: C K = 24 sequences of T = 126 frames
    (1, 1, 64, True, 1.0),         # one sequence, one step
])
def test_lstm_layer_forward_and_backward_match_autograd(N, L, IN, bidir, gain):
    from speechseparation_amd import train
    lstm, x, dh, h_ref = _reference(N, L, IN, bidir, seed=N + L, gain=gain)
    w_ih, w_hh, b_ih, b_hh = [t.detach().cuda() for t in train.stack_direction_weights(lstm, 0)]
    xg = x.detach().cuda()
    h, gates, cells = train.lstm_layer_forward(xg, w_ih, w_hh, b_ih + b_hh)
    assert float((h.cpu() - h_ref.detach()).abs().max()) < 2e-6
    dx, dw_ih, dw_hh, db = train.lstm_layer_backward(xg, h, gates, cells, dh.cuda(), w_ih, w_hh)
    sfx = ["", "_reverse"][: 2 if bidir else 1]
    ref = lambda name: torch.stack([getattr(lstm, name + "_l0" + s).grad for s in sfx])     # noqa: E731
    errs = {"dx": _rel(dx, x.grad), "dw_ih": _rel(dw_ih, ref("weight_ih")), "dw_hh": _rel(dw_hh, ref("weight_hh")),
            "db": _rel(db, ref("bias_ih"))}
    print("N=%d L=%d IN=%d ndir=%d: relative gradient errors %s" % (N, L, IN, len(sfx), {k: "%.1e" % v for k, v in errs.items()}))
    assert max(errs.values()) < 1e-4, errs
    assert torch.equal(ref("bias_ih"), ref("bias_hh"))          # one db serves both bias vectors


def test_autograd_function_stands_in_for_nn_lstm():
    """Two stacked layers (the 2-layer BLSTM of a band block, bsrnn.py:66-72) through LstmLayerFunction: same loss gradient for
    every parameter and for the input as stock nn.LSTM."""
    from speechseparation_amd import train
    torch.manual_seed(5)
    lstm = torch.nn.LSTM(64, 64, num_layers=2, batch_first=True, bidirectional=True)
    x = torch.randn(48, 12, 64, requires_grad=True)
    target = torch.randn(48, 12, 128)
    (lstm(x)[0] - target).abs().mean().backward()               # an L1 loss, as the reference's (m_dataset.py:211-216)
    want = {n: p.grad.clone() for n, p in lstm.named_parameters()}
    want_dx = x.grad.clone()

    dev = torch.device("cuda:0")
    params = [[t.detach().to(dev).requires_grad_(True) for t in train.stack_direction_weights(lstm, layer)] for layer in (0, 1)]
    xg = x.detach().to(dev).requires_grad_(True)
    y = xg
    for w in params:
        y = train.LstmLayerFunction.apply(y, *w)
    (y - target.to(dev)).abs().mean().backward()
    assert _rel(xg.grad, want_dx) < 1e-4
    for layer, w in enumerate(params):
        for name, t in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), w):
            for d, sfx in enumerate(("", "_reverse")):
                assert _rel(t.grad[d], want["%s_l%d%s" % (name, layer, sfx)]) < 1e-4, (name, layer, sfx)


def test_backward_is_bit_reproducible_and_validates_arguments():
    from speechseparation_amd import train
    from speechseparation_amd._native import NativeError
    torch.manual_seed(1)
    x = torch.randn(100, 12, 64, device="cuda")
    w_ih, w_hh, b = torch.randn(2, 256, 64, device="cuda") * 0.1, torch.randn(2, 256, 64, device="cuda") * 0.1, torch.zeros(2, 256, device="cuda")
    h, g, c = train.lstm_layer_forward(x, w_ih, w_hh, b)
    dh = torch.randn_like(h)
    a = train.lstm_layer_backward(x, h, g, c, dh, w_ih, w_hh)
    b2 = train.lstm_layer_backward(x, h, g, c, dh, w_ih, w_hh)
    assert all(torch.equal(u, v) for u, v in zip(a, b2))
    assert train.lstm_layer_backward(x, h, g, c, dh, w_ih, w_hh, need_dx=False)[0] is None
    with pytest.raises(ValueError):
        train.lstm_layer_forward(x, w_ih[:, :, :32], w_hh, b)
    with pytest.raises(NativeError):
        train.lstm_layer_forward(torch.randn(4, 3, 96, device="cuda"), torch.randn(1, 256, 96, device="cuda"), w_hh[:1], b[:1])
    with pytest.raises(ValueError):
        train.lstm_layer_forward(x.cpu(), w_ih, w_hh, b)


@pytest.mark.parametrize("M,K,N,leaky", [(200, 2, 2, True), (777, 96, 64, True), (1000, 768, 768, True), (130, 64, 128, False), (64, 514, 514, False)])
def test_linear_forward_and_backward_match_autograd(M, K, N, leaky):
    """nn.Linear (+ LeakyReLU) of the band MLPs on a column block of wider rows, against torch on the CPU."""
    from speechseparation_amd import train
    torch.manual_seed(M + K)
    wide = torch.randn(M, K + 13)
    x = wide[:, 5:5 + K].clone().requires_grad_(True)
    lin = torch.nn.Linear(K, N)
    dy = torch.randn(M, N)
    y_ref = lin(x)
    if leaky:
        y_ref = torch.nn.functional.leaky_relu(y_ref)
    (y_ref * dy).sum().backward()

    wg = wide.cuda()
    xg = wg[:, 5:5 + K].requires_grad_(True)            # row stride K + 13
    w, b = lin.weight.detach().cuda().requires_grad_(True), lin.bias.detach().cuda().requires_grad_(True)
    y = train.LinearFunction.apply(xg, w, b, leaky)
    (y * dy.cuda()).sum().backward()
    assert _rel(y, y_ref) < 2e-6
    errs = {"dx": _rel(xg.grad, x.grad), "dw": _rel(w.grad, lin.weight.grad), "db": _rel(b.grad, lin.bias.grad)}
    print("M=%d K=%d N=%d leaky=%s: relative gradient errors %s" % (M, K, N, leaky, {k: "%.1e" % v for k, v in errs.items()}))
    assert max(errs.values()) < 1e-4, errs


def test_training_forward_and_backward_of_the_whole_model_match_autograd():
    """forward_train = BSRNN.forward with every parameterised layer on the library's training kernels: the output and the
    gradient of an L1 loss (m_dataset.py:211-216 uses L1 terms) w.r.t. all 288 parameter tensors against torch.autograd on
    the CPU restatement of the reference (oracle/bsrnn_torch_cpu.py), C = 2 rows x T = 8 frames, the 'hot' weight set."""
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, train, weights
    from speechseparation_amd.bsrnn import BSRNN
    v = spec.generate_bandsplits()[0]
    sd = weights.synth_state_dict(None, seed=1, lstm_gain=3.0)
    x = torch.from_numpy(weights.synth_tensor((2, 2050, 8), seed=5, scale=1.0))
    target = torch.from_numpy(weights.synth_tensor((2, 2050, 8), seed=6, scale=1.0))

    ref = TorchCpuBSRNN(sd, v)
    params = ref.trainable()
    y_ref = ref.forward_differentiable(x)
    (y_ref - target).abs().mean().backward()

    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    y = train.forward_train(m, x.cuda())
    (y - target.cuda()).abs().mean().backward()
    assert float((y.detach().cpu() - y_ref.detach()).abs().max()) < 1e-4
    worst = ("", 0.0)
    n = 0
    for name, p in m.named_parameters():
        g_ref = params[name].grad
        if p.numel() == 0:
            continue
        assert p.grad is not None and g_ref is not None, name
        e = _rel(p.grad, g_ref) if float(g_ref.abs().max()) > 0 else float(p.grad.abs().max())
        worst = max(worst, (name, e), key=lambda t: t[1])
        n += 1
    print("%d parameter tensors, worst relative gradient error %.2e (%s)" % (n, worst[1], worst[0]))
    assert n >= 280 and worst[1] < 1e-3, worst


def test_istft_backward_is_the_transpose_of_torch_istft():
    from speechseparation_amd import train
    torch.manual_seed(3)
    R, T = 3, 7
    y = torch.randn(R, 2050, T, requires_grad=True)
    g = torch.randn(R, (T - 1) * 1024)
    yc = y.reshape(R, 1025, 2, T)
    w = torch.istft(torch.complex(yc[:, :, 0, :], yc[:, :, 1, :]), n_fft=2048, hop_length=1024, window=torch.hann_window(2048))
    (w * g).sum().backward()
    yg = y.detach().cuda().requires_grad_(True)
    wg = train.IstftFunction.apply(yg)
    (wg * g.cuda()).sum().backward()
    assert _rel(wg, w) < 2e-6
    # torch gives the imaginary parts of bins 0 and 1024 no gradient either (irfft ignores them)
    assert float(y.grad[:, 1, :].abs().max()) == 0.0 and float(yg.grad[:, 1, :].abs().max()) == 0.0
    print("iSTFT backward: relative error %.1e" % _rel(yg.grad, y.grad))
    assert _rel(yg.grad, y.grad) < 1e-5


def test_train_loss_and_its_gradient_match_the_reference_train_step():
    """m_dataset.train_infer (without discriminator) end to end: STFT -> BSRNN -> iSTFT -> L1 tri-loss and loss.backward(),
    library kernels on the GPU against torch on the CPU (torch.stft / istft + the CPU restatement of the model)."""
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, train, weights
    from speechseparation_amd.bsrnn import BSRNN
    v = spec.generate_bandsplits()[0]
    sd = weights.synth_state_dict(None, seed=0)
    mix = torch.from_numpy(weights.synth_waveform(2, 8 * 1024, seed=11))
    speech = torch.from_numpy(weights.synth_waveform(2, 8 * 1024, seed=12))

    ref = TorchCpuBSRNN(sd, v)
    params = ref.trainable()
    win = torch.hann_window(2048)

    def spec_of(wv):
        X = torch.stft(wv, n_fft=2048, hop_length=1024, return_complex=True, window=win)
        return X

    X = spec_of(mix)
    x = torch.stack((X.real, X.imag), dim=2).reshape(2, 2050, -1)
    y = ref.forward_differentiable(x)
    yc = y.reshape(2, -1, 2, y.shape[2])
    Y = torch.complex(yc[:, :, 0, :], yc[:, :, 1, :])
    x_time = torch.istft(Y, n_fft=2048, hop_length=1024, window=win)
    S = spec_of(speech)
    l1 = torch.nn.L1Loss(reduction="mean")
    loss_ref = l1(x_time, speech[:, :x_time.shape[1]]) + l1(Y.real, S.real) + l1(Y.imag, S.imag)
    loss_ref.backward()

    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    loss, xt = train.train_loss(m, mix.cuda(), speech.cuda())
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    assert _rel(xt, x_time) < 1e-5
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        if p.numel() == 0:
            continue
        g_ref = params[name].grad
        e = _rel(p.grad, g_ref) if float(g_ref.abs().max()) > 0 else float(p.grad.abs().max())
        worst = max(worst, (name, e), key=lambda t: t[1])
    print("loss %.6f (reference %.6f); worst relative gradient error %.2e (%s)" % (float(loss), float(loss_ref), worst[1], worst[0]))
    assert worst[1] < 1e-3, worst


def test_three_train_steps_follow_torch_adamw():
    """train.py:97-115 with batch_size 1: three iterations (loss, backward, AdamW(lr 1e-3, weight_decay 1e-2), zero_grad) on the
    library against the same three iterations with torch on the CPU: losses and the parameters afterwards."""
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, train, weights
    from speechseparation_amd.bsrnn import BSRNN
    v = spec.generate_bandsplits()[0]
    sd = weights.synth_state_dict(None, seed=0)
    mix = torch.from_numpy(weights.synth_waveform(2, 4 * 1024, seed=21))
    speech = torch.from_numpy(weights.synth_waveform(2, 4 * 1024, seed=22))
    ref = TorchCpuBSRNN(sd, v)
    params = ref.trainable()
    names = [k for k in params if params[k].numel() > 0]
    opt_ref = torch.optim.AdamW([params[k] for k in names], lr=1e-3, weight_decay=1e-2)
    win = torch.hann_window(2048)
    l1 = torch.nn.L1Loss(reduction="mean")
    ref_losses = []
    for _ in range(3):
        X = torch.stft(mix, n_fft=2048, hop_length=1024, return_complex=True, window=win)
        y = ref.forward_differentiable(torch.stack((X.real, X.imag), dim=2).reshape(2, 2050, -1))
        yc = y.reshape(2, -1, 2, y.shape[2])
        Y = torch.complex(yc[:, :, 0, :], yc[:, :, 1, :])
        xt = torch.istft(Y, n_fft=2048, hop_length=1024, window=win)
        S = torch.stft(speech, n_fft=2048, hop_length=1024, return_complex=True, window=win)
        loss = l1(xt, speech[:, :xt.shape[1]]) + l1(Y.real, S.real) + l1(Y.imag, S.imag)
        loss.backward()
        opt_ref.step()
        opt_ref.zero_grad()
        ref_losses.append(float(loss))

    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    opt = train.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2)
    losses = [float(train.train_step(m, opt, mix.cuda(), speech.cuda())) for _ in range(3)]
    print("losses", losses, "reference", ref_losses)
    assert losses[2] < losses[0]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-4 * abs(b)
    got = dict(m.named_parameters())
    worst = max((float((got[k].detach().cpu() - params[k].detach()).abs().max()), k) for k in names)
    print("largest parameter difference after three steps: %.2e (%s)" % worst)
    # AdamW's first steps move every parameter by ~lr whatever the gradient's size: sign-level noise of tiny gradients is amplified
    assert worst[0] < 2e-3


def test_graphed_train_step_replays_the_eager_iterations():
    """train.GraphedTrainStep: the iteration of train.py:97-115 captured once into a hipGraph (STFTs, forward, loss, backward, AdamW with
    its step count on the device, zero_grad) and replayed, against train.train_step driven from Python: five iterations with a
    different clip each (1 eager warm-up, capture + replay, three replays), then a learning-rate change through set_lr and one more.
    Same kernels in the same order: losses and parameters agree to fp32 rounding of the bias corrections (device pow vs host pow);
    the optimizer's torch-format state_dict carries the step count of the replays."""
    from speechseparation_amd import train, weights
    from speechseparation_amd.bsrnn import BSRNN
    sd = weights.synth_state_dict(None, seed=0)
    n = 6 * 1024
    clips = [(torch.from_numpy(weights.synth_waveform(2, n, seed=40 + i)).cuda(), torch.from_numpy(weights.synth_waveform(2, n, seed=60 + i)).cuda())
             for i in range(6)]

    def fresh(capturable):
        m = BSRNN().train()
        m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
        m = m.to("cuda:0")
        return m, train.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, capturable=capturable)

    m1, o1 = fresh(False)
    eager = [float(train.train_step(m1, o1, a, b)) for a, b in clips[:5]]
    o1.set_lr(3e-4)
    eager.append(float(train.train_step(m1, o1, *clips[5])))

    m2, o2 = fresh(True)
    step = train.GraphedTrainStep(m2, o2, 2, n, warmup=1)
    graphed = [float(step(a, b)) for a, b in clips[:5]]
    assert step.graph is not None and step.calls == 5
    o2.set_lr(3e-4)
    graphed.append(float(step(*clips[5])))
    print("eager  ", eager)
    print("graphed", graphed)
    for a, b in zip(graphed, eager):
        assert abs(a - b) <= 2e-6 * abs(b), (graphed, eager)
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    worst = max((float((p1[k] - p2[k]).abs().max()), k) for k in p1 if p1[k].numel())
    print("largest parameter difference eager vs graphed after six steps: %.2e (%s)" % worst)
    assert worst[0] < 2e-6
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert set(s1["state"]) == set(s2["state"])
    assert all(float(s2["state"][i]["step"]) == 6.0 for i in s2["state"])
    assert s2["param_groups"][0]["lr"] == pytest.approx(3e-4)
    with pytest.raises(ValueError):
        step(clips[0][0][:, :4096], clips[0][1][:, :4096])                  # another clip shape than the captured one
    with pytest.raises(ValueError):
        train.GraphedTrainStep(m1, o1, 2, n)                                # optimizer without the device-side step count


@pytest.mark.parametrize("leaky", [True, False])
def test_grouped_linear_matches_autograd(leaky):
    """The same layer of several bands in grouped launches (bsrnn_linear_group_train_*): different widths, inputs that are column
    blocks of one wide row buffer, one input that needs no gradient."""
    from speechseparation_amd import train
    M = 333
    dims = [(2, 2), (6, 64), (96, 96), (514, 514), (64, 128)]
    for seed in range(9, 60):
        # LeakyReLU's derivative jumps at 0: data whose pre-activations all stay clear of it, so that a rounding-level
        # difference between the two evaluations cannot pick the other branch (seed 9 has one of 171 162 values at 1e-7)
        torch.manual_seed(seed)
        wide = torch.randn(M, sum(k for k, _ in dims) + 7)
        lins = [torch.nn.Linear(k, n) for k, n in dims]
        xs, pos = [], 3
        for k, _ in dims:
            xs.append(wide[:, pos:pos + k].clone().requires_grad_(True))
            pos += k
        with torch.no_grad():
            clear = min(float(lin(x).abs().min()) for lin, x in zip(lins, xs))
        if clear > 1e-5:
            break
    assert clear > 1e-5
    dys = [torch.randn(M, n) for _, n in dims]
    act = torch.nn.functional.leaky_relu if leaky else (lambda t: t)
    loss = sum((act(lin(x)) * dy).sum() for lin, x, dy in zip(lins, xs, dys))
    loss.backward()

    wg = wide.cuda()
    xg, pos = [], 3
    for i, (k, _) in enumerate(dims):
        t = wg[:, pos:pos + k]
        xg.append(t.requires_grad_(True) if i != 1 else t)          # job 1: no input gradient wanted
        pos += k
    ws = [lin.weight.detach().cuda().requires_grad_(True) for lin in lins]
    bs = [lin.bias.detach().cuda().requires_grad_(True) for lin in lins]
    ys = train.GroupedLinearFunction.apply(leaky, len(dims), *xg, *ws, *bs)
    sum((y * dy.cuda()).sum() for y, dy in zip(ys, dys)).backward()
    for i, (lin, x) in enumerate(zip(lins, xs)):
        assert _rel(ys[i], act(lin(x))) < 2e-6
        assert _rel(ws[i].grad, lin.weight.grad) < 1e-4 and _rel(bs[i].grad, lin.bias.grad) < 1e-4
        if i != 1:
            assert _rel(xg[i].grad, x.grad) < 1e-4


def _dp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)        # both ranks share cuda:0 here: gradients staged through the host
    from speechseparation_amd import train, weights
    from speechseparation_amd.bsrnn import BSRNN
    sd = weights.synth_state_dict(None, seed=0)
    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    opt = train.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2)
    mix = torch.from_numpy(weights.synth_waveform(2, 3 * 1024, seed=31 + rank)).cuda()       # every rank its own clip
    speech = torch.from_numpy(weights.synth_waveform(2, 3 * 1024, seed=41 + rank)).cuda()
    loss = train.train_step(m, opt, mix, speech)
    out = {k: p.detach().cpu().numpy() for k, p in m.named_parameters() if p.numel() > 0}
    q.put((rank, float(loss), out))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_train_step_averages_the_gradients_of_the_ranks():
    """Two processes (one clip each, gloo rendezvous, both on cuda:0): after train_step both hold the same parameters, equal to one
    process that averages the two clips' gradients itself before the same AdamW step."""
    import socket
    import torch.multiprocessing as mp
    from speechseparation_amd import train, weights
    from speechseparation_amd.bsrnn import BSRNN
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (lo, out)) for r, lo, out in (q.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for k in got[0][1]:
        assert np.array_equal(got[0][1][k], got[1][1][k]), k             # the ranks stay in lockstep

    sd = weights.synth_state_dict(None, seed=0)
    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    opt = train.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2)
    grads = None
    for r in range(2):
        mix = torch.from_numpy(weights.synth_waveform(2, 3 * 1024, seed=31 + r)).cuda()
        speech = torch.from_numpy(weights.synth_waveform(2, 3 * 1024, seed=41 + r)).cuda()
        loss, _ = train.train_loss(m, mix, speech)
        assert abs(float(loss) - got[r][0]) < 1e-6 * abs(float(loss))
        loss.backward()
        g = [p.grad.clone() if p.grad is not None else None for p in m.parameters()]
        opt.zero_grad()
        grads = g if grads is None else [a + b if a is not None else None for a, b in zip(grads, g)]
    for p, g in zip(m.parameters(), grads):
        if g is not None:
            p.grad = g / 2
    opt.step()
    worst = max(float(np.abs(p.detach().cpu().numpy() - got[0][1][k]).max()) for k, p in m.named_parameters() if p.numel() > 0)
    print("data-parallel step vs single process with averaged gradients: largest parameter difference %.2e" % worst)
    assert worst < 1e-5


def test_sdr_loss_gradient_matches_the_reference_option():
    """`train.py --loss_sdr`: back-propagating -SDR of the time-domain estimate (m_dataset.py:217-220) instead of the L1 tri-loss."""
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, train, weights
    from speechseparation_amd.bsrnn import BSRNN
    v = spec.generate_bandsplits()[0]
    sd = weights.synth_state_dict(None, seed=0)
    mix = torch.from_numpy(weights.synth_waveform(2, 5 * 1024, seed=51))
    speech = torch.from_numpy(weights.synth_waveform(2, 5 * 1024, seed=52))
    ref = TorchCpuBSRNN(sd, v)
    params = ref.trainable()
    win = torch.hann_window(2048)
    X = torch.stft(mix, n_fft=2048, hop_length=1024, return_complex=True, window=win)
    y = ref.forward_differentiable(torch.stack((X.real, X.imag), dim=2).reshape(2, 2050, -1))
    yc = y.reshape(2, -1, 2, y.shape[2])
    xt = torch.istft(torch.complex(yc[:, :, 0, :], yc[:, :, 1, :]), n_fft=2048, hop_length=1024, window=win)
    st = speech[:, :xt.shape[1]]
    sdr_ref = (10 * torch.log10((st.square().sum(1) + 1e-9) / ((xt - st).square().sum(1) + 1e-9))).mean()
    (-sdr_ref).backward()

    m = BSRNN().train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    _, x_time = train.train_loss(m, mix.cuda(), speech.cuda())
    val = train.sdr(x_time, speech.cuda()[:, :x_time.shape[1]])
    (-val).backward()
    assert abs(float(val.detach()) - float(sdr_ref.detach())) < 1e-4
    worst = max(_rel(p.grad, params[n].grad) for n, p in m.named_parameters() if p.numel() > 0 and float(params[n].grad.abs().max()) > 0)
    print("SDR %.4f dB (reference %.4f); worst relative gradient error %.2e" % (float(val.detach()), float(sdr_ref.detach()), worst))
    assert worst < 1e-3


def test_train_entry_point_runs_epochs_validates_and_checkpoints(tmp_path):
    """train.py (the reference's command line): two epochs on synthetic clips, validation through the inference path with the
    UPDATED weights, checkpoints in the reference's file names, --resume."""
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, PYTHONPATH=REPO)
    cmd = [sys.executable, os.path.join(REPO, "train.py"), "--synthetic", "3", "--seconds", "1", "--epochs", "2", "--batch_size", "2",
           "--synthetic-weights", "0", "--outdir", str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    losses = [float(l.split()[3]) for l in p.stdout.splitlines() if l.startswith("Epoch") and "Loss" in l]
    vals = [float(l.split()[2]) for l in p.stdout.splitlines() if l.startswith("Validation Loss")]
    print(p.stdout)
    assert len(losses) == 2 and losses[1] < losses[0]
    assert len(vals) == 2 and vals[1] != vals[0]                 # the validation pass sees the trained weights
    for f in ("model.pth", "model-always.pth", "optimizer.pth", "optimizer-always.pth"):
        assert (tmp_path / f).exists(), f
    sd = torch.load(tmp_path / "model-always.pth", weights_only=True)
    assert len(sd) == 288
    p2 = subprocess.run(cmd + ["--resume", "--epochs", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-2000:]
    resumed = [float(l.split()[3]) for l in p2.stdout.splitlines() if l.startswith("Epoch") and "Loss" in l]
    assert resumed and resumed[0] < losses[0]


def test_train_entry_point_with_graph_replay_prints_the_same_epochs(tmp_path):
    """train.py --graph (every iteration one hipGraph replay) against the same command line launch by launch: the epoch losses /
    SDRs and the validation figures (inference path on the weights the replays wrote) agree; --resume continues from the
    optimizer's saved step count."""
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, PYTHONPATH=REPO)
    runs = {}
    for tag, extra in (("eager", []), ("graph", ["--graph"])):
        out = tmp_path / tag
        out.mkdir()
        cmd = [sys.executable, os.path.join(REPO, "train.py"), "--synthetic", "4", "--seconds", "1", "--epochs", "2", "--batch_size", "1",
               "--synthetic-weights", "0", "--outdir", str(out)] + extra
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        ep = [(float(l.split()[3]), float(l.split()[5])) for l in p.stdout.splitlines() if l.startswith("Epoch") and "Loss" in l]
        va = [(float(l.split()[2]), float(l.split()[5])) for l in p.stdout.splitlines() if l.startswith("Validation Loss")]
        runs[tag] = (ep, va, cmd)
        print(tag, ep, va)
    (e_ep, e_va, _), (g_ep, g_va, g_cmd) = runs["eager"], runs["graph"]
    assert len(g_ep) == 2 and len(g_va) == 2
    for a, b in zip(g_ep + g_va, e_ep + e_va):
        assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-4 * max(1.0, abs(b[1])), (runs["graph"][:2], runs["eager"][:2])
    before = {float(st["step"]) for st in torch.load(tmp_path / "graph" / "optimizer.pth", weights_only=True)["state"].values()}
    assert before in ({4.0}, {8.0})                               # the best epoch's optimizer: 4 clips per epoch
    p2 = subprocess.run(g_cmd + ["--resume", "--epochs", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-2000:]
    resumed = [float(l.split()[3]) for l in p2.stdout.splitlines() if l.startswith("Epoch") and "Loss" in l]
    assert resumed and resumed[0] < g_ep[0][0]
    after = {float(st["step"]) for st in torch.load(tmp_path / "graph" / "optimizer-always.pth", weights_only=True)["state"].values()}
    assert after == {before.pop() + 4.0}                          # the device-side step count continued from the loaded one


def test_training_gradients_on_the_41_band_table():
    """K = 42 (BASELINE config 5's band table): 41 live bands = four grouped launches per Sequential slot (12 jobs per launch),
    band sequences of 42 steps; the gradients of all parameters against torch.autograd on the CPU restatement."""
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, train, weights
    from speechseparation_amd.bsrnn import BSRNN
    v = spec.variant_bandsplits("41")
    sd = weights.synth_state_dict(v, seed=3)
    x = torch.from_numpy(weights.synth_tensor((2, 2050, 5), seed=7, scale=1.0))
    target = torch.from_numpy(weights.synth_tensor((2, 2050, 5), seed=8, scale=1.0))
    ref = TorchCpuBSRNN(sd, v)
    params = ref.trainable()
    (ref.forward_differentiable(x) - target).abs().mean().backward()
    m = BSRNN(v).train()
    m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
    m = m.to("cuda:0")
    (train.forward_train(m, x.cuda()) - target.cuda()).abs().mean().backward()
    worst = ("", 0.0)
    n = 0
    for name, p in m.named_parameters():
        if p.numel() == 0:
            continue
        g_ref = params[name].grad
        e = _rel(p.grad, g_ref) if float(g_ref.abs().max()) > 0 else float(p.grad.abs().max())
        worst = max(worst, (name, e), key=lambda t: t[1])
        n += 1
    print("41-band table: %d parameter tensors, worst relative gradient error %.2e (%s)" % (n, worst[1], worst[0]))
    assert n > 800 and worst[1] < 1e-3, worst


def test_graphs_of_two_clip_lengths_survive_a_growing_scratch():
    """train.py --graph keeps one captured iteration per clip length.  The library's training scratch is grow-only: the eager warm-up of a
    LONGER clip outgrows it after the first graph has been captured with the old buffer's address in its kernel nodes.  The old buffer must stay
    valid (it is retired, not freed: csrc/api.hip::train_scratch): replaying the first graph after the second one's warm-up and capture
    gives the same losses as the Python-driven loop on the same sequence of clips."""
    from speechseparation_amd import train, weights
    from speechseparation_amd.bsrnn import BSRNN
    sd = weights.synth_state_dict(None, seed=0)
    n_a, n_b = 4 * 1024, 9 * 1024                                            # ascending: the second length more than 25 % larger
    mk = lambda n, s: (torch.from_numpy(weights.synth_waveform(2, n, seed=s)).cuda(), torch.from_numpy(weights.synth_waveform(2, n, seed=s + 50)).cuda())
    seq = [("a", mk(n_a, 1)), ("a", mk(n_a, 2)), ("a", mk(n_a, 3)), ("b", mk(n_b, 4)), ("b", mk(n_b, 5)), ("b", mk(n_b, 6)),
           ("a", mk(n_a, 7)), ("b", mk(n_b, 8)), ("a", mk(n_a, 9))]

    def fresh(capturable):
        m = BSRNN().train()
        m.load_state_dict({k: torch.from_numpy(np.array(a, copy=True)) for k, a in sd.items()})
        m = m.to("cuda:0")
        return m, train.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, capturable=capturable)

    m1, o1 = fresh(False)
    eager = [float(train.train_step(m1, o1, *clip)) for _, clip in seq]
    m2, o2 = fresh(True)
    steps = {"a": train.GraphedTrainStep(m2, o2, 2, n_a, warmup=1), "b": train.GraphedTrainStep(m2, o2, 2, n_b, warmup=1)}
    graphed = [float(steps[k](*clip)) for k, clip in seq]
    assert steps["a"].graph is not None and steps["b"].graph is not None
    print("eager  ", eager)
    print("graphed", graphed)
    for a, b in zip(graphed, eager):
        assert np.isfinite(a) and abs(a - b) <= 5e-6 * abs(b), (graphed, eager)
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    worst = max((float((p1[k] - p2[k]).abs().max()), k) for k in p1 if p1[k].numel())
    print("largest parameter difference eager vs two graphs after nine steps: %.2e (%s)" % worst)
    assert worst[0] < 5e-6
