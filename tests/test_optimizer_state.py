"""The optimizer checkpoint the reference writes (`torch.save(optimizer.state_dict(), "optimizer.pth")`, train.py:169-172) is
torch.optim.AdamW's own state_dict; speechseparation_amd.train.AdamW must read and write exactly that.
CPU part: layout round trips against torch.optim.AdamW (no kernel runs); the GPU part steps both and compares parameters."""
import copy

import numpy as np
import pytest
import torch

from speechseparation_amd import train as hip_train


def _params(device="cpu"):
    g = torch.Generator().manual_seed(3)
    shapes = [(6, 4), (6,), (0,), (5, 6), (5,), (0,), (3, 5)]           # zero-size parameters in the middle, like the model's
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(device)) for s in shapes]


def _torch_steps(params, n, seed=0, skip=()):
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-2)
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        for i, p in enumerate(params):
            p.grad = None if i in skip else torch.randn(p.shape, generator=g).to(p.device)
        opt.step()
    return opt


def test_torch_state_dict_loads_and_comes_back_unchanged():
    params = _params()
    ref = _torch_steps(params, 3, skip=(6,))                          # the last parameter never got a gradient: no state for it
    sd = ref.state_dict()
    ours = hip_train.AdamW(params, lr=5e-4, betas=(0.8, 0.9), eps=1e-6, weight_decay=0.5)     # everything differs until the load
    ours.load_state_dict(copy.deepcopy(sd))
    assert (ours.lr, ours.betas, ours.eps, ours.weight_decay) == (1e-3, (0.9, 0.999), 1e-8, 1e-2)
    assert ours.steps == [3, 3, 3, 3, 3, 3, 0] and ours.t == 3
    back = ours.state_dict()
    assert sorted(back["state"]) == sorted(sd["state"]) == [0, 1, 2, 3, 4, 5]
    for i in sd["state"]:
        for k in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(back["state"][i][k], sd["state"][i][k]), (i, k)
            assert back["state"][i][k].dtype == sd["state"][i][k].dtype and back["state"][i][k].shape == sd["state"][i][k].shape
    assert back["param_groups"] == sd["param_groups"]                 # every key of this torch version's AdamW group, same values
    # ... and torch accepts it: a fresh torch.optim.AdamW loaded from OUR file continues exactly like the original
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in params]
    cont = torch.optim.AdamW(p2, lr=0.5)
    cont.load_state_dict(back)
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    for p, q in zip(params, p2):
        p.grad = torch.randn(p.shape, generator=g1)
        q.grad = torch.randn(q.shape, generator=g2)
    ref.step(); cont.step()
    for p, q in zip(params, p2):
        assert torch.equal(p, q)


def test_load_refuses_a_state_that_does_not_fit():
    params = _params()
    ours = hip_train.AdamW(params)
    with pytest.raises(ValueError, match="torch.optim.AdamW state_dict"):
        ours.load_state_dict({"step": 3, "exp_avg": [], "exp_avg_sq": []})          # the round-2 private layout
    sd = _torch_steps(_params()[:-1], 1).state_dict()
    with pytest.raises(ValueError, match="parameters"):
        ours.load_state_dict(sd)
    sd = _torch_steps(_params(), 1).state_dict()
    sd["state"][0]["exp_avg"] = torch.zeros(4, 6)
    with pytest.raises(ValueError, match="shape"):
        ours.load_state_dict(sd)


@pytest.mark.gpu
def test_optimizer_state_interchanges_with_torch_on_the_gpu():
    """Two steps on the library's AdamW kernel, state saved, loaded into torch.optim.AdamW, one more step on both from the same
    gradients: the parameters agree (and the reverse direction: torch's state into ours)."""
    assert torch.cuda.is_available()
    dev = "cuda:0"
    params = _params(dev)
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    ours = hip_train.AdamW(params, lr=1e-3, weight_decay=1e-2)
    ref = torch.optim.AdamW(twin, lr=1e-3, weight_decay=1e-2)
    g = torch.Generator().manual_seed(5)
    def grads():
        return [torch.randn(p.shape, generator=g).to(dev) for p in params]
    for _ in range(2):
        gs = grads()
        for p, q, gr in zip(params, twin, gs):
            p.grad, q.grad = gr.clone(), gr.clone()
        ours.step(); ref.step()
    for p, q in zip(params, twin):
        assert torch.allclose(p, q, rtol=0, atol=1e-6)
    # ours -> torch
    fresh = [torch.nn.Parameter(p.detach().clone()) for p in params]
    t2 = torch.optim.AdamW(fresh, lr=123.0)
    t2.load_state_dict(ours.state_dict())
    # torch -> ours
    fresh2 = [torch.nn.Parameter(q.detach().clone()) for q in twin]
    o2 = hip_train.AdamW(fresh2, lr=123.0)
    o2.load_state_dict(ref.state_dict())
    gs = grads()
    for p, q, a, b, gr in zip(params, twin, fresh, fresh2, gs):
        p.grad, q.grad, a.grad, b.grad = gr.clone(), gr.clone(), gr.clone(), gr.clone()
    ours.step(); ref.step(); t2.step(); o2.step()
    for p, q, a, b in zip(params, twin, fresh, fresh2):
        assert torch.allclose(p, q, rtol=0, atol=1e-6) and torch.allclose(a, q, rtol=0, atol=1e-6) and torch.allclose(b, q, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_leaky_relu_backward_at_exactly_zero_follows_torch():
    """torch's leaky_relu_backward takes slope 0.01 where the pre-activation is exactly 0 (x > 0 ? 1 : 0.01); the library reads the sign
    off the OUTPUT, which is 0 there as well."""
    dev = "cuda:0"
    x = torch.tensor([[1.0, 0.0, -2.0, 0.0], [0.0, 3.0, 0.0, -1.0]], device=dev)
    w = torch.eye(4, device=dev).requires_grad_(True)                    # identity: the pre-activation equals x, zeros included
    b = torch.zeros(4, device=dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = hip_train.LinearFunction.apply(xr, w, b, True)
    y.backward(torch.ones_like(y))
    x2 = x.clone().requires_grad_(True)
    w2 = torch.eye(4, device=dev).requires_grad_(True)
    b2 = torch.zeros(4, device=dev).requires_grad_(True)
    torch.nn.functional.leaky_relu(torch.nn.functional.linear(x2, w2, b2), 0.01).backward(torch.ones_like(y))
    assert torch.equal(xr.grad, x2.grad) and torch.equal(w.grad, w2.grad) and torch.equal(b.grad, b2.grad)
