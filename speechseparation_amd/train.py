"""Training step, part 1: the recurrent layers' forward-with-saves and backward on the HIP library.

The reference trains through torch.autograd (train.py:97-115); `LstmLayerFunction` is the same `nn.LSTM` layer
(bsrnn.py:66-72) as a `torch.autograd.Function` whose forward and backward are the library's kernels
(`bsrnn_lstm_train_forward` / `bsrnn_lstm_train_backward`, exact fp32), so that it can stand where `nn.LSTM` stands while
the rest of the step is still built.  No CPU fallback: the extension has to be there and the tensors on the GPU.
"""
import ctypes

import torch

from . import _native
from .spec import BAND_FEATURES as band_features, generate_bandsplits      # bsrnn.py:60, :247

_lib = _native.lib
_ctx = {}


def _context(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctx:
        v = generate_bandsplits()[0]                        # (the training entry points use no model weights of the context)
        widths = (ctypes.c_int32 * len(v))(*v)
        ctx = ctypes.c_void_p()
        _native.check(_lib.bsrnn_create(idx, widths, len(v), ctypes.byref(ctx)))
        _ctx[idx] = ctx
    return _ctx[idx]


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _s(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def lstm_layer_forward(x, w_ih, w_hh, bias):
    """x [N, L, IN] (cuda), w_ih [ndir, 256, IN], w_hh [ndir, 256, 64], bias [ndir, 256] (= b_ih + b_hh)
    -> (h [N, L, ndir*64], gates [N, L, ndir, 256], cells [N, L, ndir, 64])."""
    if not x.is_cuda:
        raise ValueError("the training kernels run on the GPU: x must be a cuda tensor")
    N, L, IN = x.shape
    ndir = w_ih.shape[0]
    if tuple(w_ih.shape) != (ndir, 4 * band_features, IN) or tuple(w_hh.shape) != (ndir, 4 * band_features, band_features) \
            or tuple(bias.shape) != (ndir, 4 * band_features):
        raise ValueError("weight shapes do not match x [N, L, %d]" % IN)
    dev = x.device
    x, w_ih, w_hh, bias = _f32c(x), _f32c(w_ih), _f32c(w_hh), _f32c(bias)
    with torch.cuda.device(dev):
        h = torch.empty((N, L, ndir * band_features), device=dev)
        gates = torch.empty((N, L, ndir, 4 * band_features), device=dev)
        cells = torch.empty((N, L, ndir, band_features), device=dev)
        _native.check(_lib.bsrnn_lstm_train_forward(_context(dev), _p(x), _p(w_ih), _p(w_hh), _p(bias), _p(h), _p(gates), _p(cells),
                                                    N, L, IN, ndir, _s(dev)))
    return h, gates, cells


def lstm_layer_backward(x, h, gates, cells, dh, w_ih, w_hh, need_dx=True):
    """-> (dx or None, dw_ih, dw_hh, db) for the loss gradient dh [N, L, ndir*64] of the layer output."""
    N, L, IN = x.shape
    ndir = w_ih.shape[0]
    dev = x.device
    x, h, dh, w_ih, w_hh = _f32c(x), _f32c(h), _f32c(dh), _f32c(w_ih), _f32c(w_hh)
    with torch.cuda.device(dev):
        dx = torch.empty_like(x) if need_dx else None
        dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        db = torch.empty((ndir, 4 * band_features), device=dev)
        _native.check(_lib.bsrnn_lstm_train_backward(_context(dev), _p(x), _p(h), _p(gates), _p(cells), _p(dh), _p(w_ih), _p(w_hh),
                                                     _p(dx), _p(dw_ih), _p(dw_hh), _p(db), N, L, IN, ndir, _s(dev)))
    return dx, dw_ih, dw_hh, db


class LstmLayerFunction(torch.autograd.Function):
    """One nn.LSTM layer (all directions) with the library's forward and backward.
    apply(x, w_ih, w_hh, b_ih, b_hh) -> h; parameters stacked over directions as in `stack_direction_weights`."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        h, gates, cells = lstm_layer_forward(x, w_ih, w_hh, b_ih + b_hh)
        ctx.save_for_backward(x, h, gates, cells, w_ih, w_hh)
        ctx.need_dx = x.requires_grad
        return h

    @staticmethod
    def backward(ctx, dh):
        x, h, gates, cells, w_ih, w_hh = ctx.saved_tensors
        dx, dw_ih, dw_hh, db = lstm_layer_backward(x, h, gates, cells, dh, w_ih, w_hh, need_dx=ctx.need_dx)
        return dx, dw_ih, dw_hh, db, db


def stack_direction_weights(lstm, layer):
    """(w_ih, w_hh, b_ih, b_hh) of layer `layer` of a torch nn.LSTM, stacked over its directions (forward, then `_reverse`)."""
    sfx = ["", "_reverse"][: 2 if lstm.bidirectional else 1]
    get = lambda name: torch.stack([getattr(lstm, "%s_l%d%s" % (name, layer, s)) for s in sfx])   # noqa: E731
    return get("weight_ih"), get("weight_hh"), get("bias_ih"), get("bias_hh")
