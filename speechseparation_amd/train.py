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


# ------------------------------------------------------------------------------------------ nn.Linear (+ LeakyReLU)
def _rows(t):
    """2-D view whose rows are contiguous (a band is a column block of wider rows: row stride = the leading dimension)."""
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32) else t.to(torch.float32).contiguous()


class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b) on [M, K] rows with the library's forward and backward (bsrnn_linear_train_*);
    apply(x, weight, bias, leaky) where leaky selects LeakyReLU(0.01) (nn.LeakyReLU() of the reference's Sequentials)."""

    @staticmethod
    def forward(ctx, x, w, b, leaky):
        if not x.is_cuda:
            raise ValueError("the training kernels run on the GPU: x must be a cuda tensor")
        x, w, b = _rows(x.detach()), _f32c(w), _f32c(b)
        M, K = x.shape
        N = w.shape[0]
        if tuple(w.shape) != (N, K) or tuple(b.shape) != (N,):
            raise ValueError("Linear: weight %s / bias %s do not match x [M, %d]" % (tuple(w.shape), tuple(b.shape), K))
        dev = x.device
        with torch.cuda.device(dev):
            y = torch.empty((M, N), device=dev)
            _native.check(_lib.bsrnn_linear_train_forward(_context(dev), _p(x), x.stride(0), _p(w), _p(b), _p(y), N, M, K, N, int(leaky), _s(dev)))
        ctx.save_for_backward(x, w, y)
        ctx.leaky = bool(leaky)
        ctx.need_dx = True
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dy = _rows(dy)
        dev = x.device
        with torch.cuda.device(dev):
            dx = torch.empty((M, K), device=dev) if ctx.needs_input_grad[0] else None
            dw, db = torch.empty_like(w), torch.empty((N,), device=dev)
            _native.check(_lib.bsrnn_linear_train_backward(_context(dev), _p(x), x.stride(0), _p(w), _p(y), N, _p(dy), dy.stride(0), _p(dx), K,
                                                           _p(dw), _p(db), M, K, N, int(ctx.leaky), _s(dev)))
        return dx, dw, db, None


def _linear(x, lin, leaky):
    return LinearFunction.apply(x, lin.weight, lin.bias, leaky)


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else 0) for t in ts])


def _i32_array(vals):
    return (ctypes.c_int32 * len(vals))(*vals)


class GroupedLinearFunction(torch.autograd.Function):
    """n Linear (+ LeakyReLU) layers that share their row count - the same layer of all bands (the reference loops over the
    bands, bsrnn.py:406-411 / :423-425) - in grouped launches: apply(leaky, n, x_0..x_{n-1}, w_0.., b_0..) -> (y_0, ..., y_{n-1})."""

    @staticmethod
    def forward(ctx, leaky, n, *ts):
        xs = [_rows(t.detach()) for t in ts[:n]]
        ws = [_f32c(t) for t in ts[n:2 * n]]
        bs = [_f32c(t) for t in ts[2 * n:3 * n]]
        M = xs[0].shape[0]
        dev = xs[0].device
        Ks = [w.shape[1] for w in ws]
        Ns = [w.shape[0] for w in ws]
        for x, w, b in zip(xs, ws, bs):
            if not x.is_cuda or x.shape[0] != M or x.shape[1] != w.shape[1] or tuple(b.shape) != (w.shape[0],):
                raise ValueError("GroupedLinear: every job needs cuda x [%d, K_i], weight [N_i, K_i], bias [N_i]" % M)
        with torch.cuda.device(dev):
            ys = [torch.empty((M, N), device=dev) for N in Ns]
            _native.check(_lib.bsrnn_linear_group_train_forward(_context(dev), n, _ptr_array(xs), _i32_array([x.stride(0) for x in xs]), _ptr_array(ws),
                                                                _ptr_array(bs), _ptr_array(ys), _i32_array(Ns), _i32_array(Ks), _i32_array(Ns),
                                                                M, int(leaky), _s(dev)))
        ctx.save_for_backward(*xs, *ws, *ys)
        ctx.n, ctx.leaky = n, bool(leaky)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        xs, ws, ys = saved[:n], saved[n:2 * n], saved[2 * n:3 * n]
        M = xs[0].shape[0]
        dev = xs[0].device
        Ks = [w.shape[1] for w in ws]
        Ns = [w.shape[0] for w in ws]
        with torch.cuda.device(dev):
            dys = [_rows(d) if d is not None else torch.zeros((M, N), device=dev) for d, N in zip(dys, Ns)]
            dxs = [torch.empty((M, K), device=dev) if ctx.needs_input_grad[2 + i] else None for i, K in enumerate(Ks)]
            dws = [torch.empty_like(w) for w in ws]
            dbs = [torch.empty((N,), device=dev) for N in Ns]
            _native.check(_lib.bsrnn_linear_group_train_backward(
                _context(dev), n, _ptr_array(xs), _i32_array([x.stride(0) for x in xs]), _ptr_array(ws), _ptr_array(ys), _i32_array(Ns),
                _ptr_array(dys), _i32_array([d.stride(0) for d in dys]), _ptr_array(dxs), _i32_array(Ks), _ptr_array(dws), _ptr_array(dbs),
                _i32_array(Ks), _i32_array(Ns), M, int(ctx.leaky), _s(dev)))
        return (None, None, *dxs, *dws, *dbs)


def _glinear(xs, lins, leaky):
    """The same Sequential slot of every live band: lists of inputs and of nn.Linear modules -> list of outputs."""
    n = len(xs)
    return list(GroupedLinearFunction.apply(leaky, n, *xs, *[m.weight for m in lins], *[m.bias for m in lins]))


def _rnn_block(m, x):
    """NormRNNResidual (bsrnn.py:78-87) on [N, L, 64]: fc_in -> 2-layer LSTM -> fc -> + x."""
    N, L, H = x.shape
    u = _linear(x.reshape(N * L, H), m.fc_in, False).reshape(N, L, H)
    for layer in range(m.rnn.num_layers):
        u = LstmLayerFunction.apply(u, *stack_direction_weights(m.rnn, layer))
    return _linear(u.reshape(N * L, u.shape[2]), m.fc, False).reshape(N, L, H) + x


def forward_train(model, x):
    """BSRNN.forward (bsrnn.py:385-443) with the autograd graph kept: every parameterised layer (the 110 Linear layers of the
    band MLPs - one grouped launch per Sequential slot over all bands -, fc_in / fc and the LSTM layers of the four dual-path
    blocks) runs the library's training kernels forward and backward; the glue without parameters (slices, stack, permutes,
    residual adds, x * mask) is torch on the same device.
    `model` is a speechseparation_amd.BSRNN on a cuda device; x [C, 2050, T] -> y [C, 2050, T]."""
    C, F2, T = x.shape
    v = model.band_widths
    H = band_features
    K = len(v)
    live = [i for i, w in enumerate(v) if w > 0]
    xt = x.permute(0, 2, 1).reshape(C * T, F2)                    # frame rows (bsrnn.py:406)
    pos, cols = 0, {}
    for i in live:
        cols[i] = (pos, pos + 2 * v[i])
        pos += 2 * v[i]
    slot = lambda mods, k: [mods[i][k] for i in live]             # noqa: E731  the k-th entry of every live band's Sequential
    # BandSplit (bsrnn.py:404-415): bandFCs_pre (2 layers, its output is the mask's residual), bandFCs (3 layers)
    h = _glinear([xt[:, a:b] for a, b in (cols[i] for i in live)], slot(model.bandFCs_pre, 0), True)
    residual = _glinear(h, slot(model.bandFCs_pre, 2), True)
    f = _glinear(residual, slot(model.bandFCs, 0), True)
    f = _glinear(f, slot(model.bandFCs, 2), True)
    f = dict(zip(live, _glinear(f, slot(model.bandFCs, 4), False)))
    feats = [f[i] if i in f else model.bandFCs[i][0].trainable_constant.expand(C * T, H) for i in range(K)]   # bsrnn.py:12-24
    z = torch.stack(feats, 1).reshape(C, T, K, H)               # bsrnn.py:415
    for j, holder in enumerate(model.lstms):
        if j % 2 == 0:                                          # BandwiseLSTM (bsrnn.py:138-153)
            z = _rnn_block(holder.m, z.reshape(C * T, K, H)).reshape(C, T, K, H)
        else:                                                   # TimewiseLSTM (bsrnn.py:106-120)
            zt = z.permute(0, 2, 1, 3).reshape(C * K, T, H)
            z = _rnn_block(holder.m, zt).reshape(C, K, T, H).permute(0, 2, 1, 3)
    # MaskEstimation (bsrnn.py:420-430): bandFCs_back (3 layers, LeakyReLU after each), bandFCs_back_post (2 layers) + skip
    zb = z.permute(2, 0, 1, 3).reshape(K, C * T, H)             # one contiguous [M, 64] block per band
    b = _glinear([zb[i] for i in live], slot(model.bandFCs_back, 0), True)
    b = _glinear(b, slot(model.bandFCs_back, 2), True)
    b = _glinear(b, slot(model.bandFCs_back, 4), True)
    b = _glinear(b, slot(model.bandFCs_back_post, 0), True)
    b = _glinear(b, slot(model.bandFCs_back_post, 2), False)
    mask = torch.cat([r + p for r, p in zip(residual, b)], 1).reshape(C, T, F2).permute(0, 2, 1)   # bsrnn.py:425-432
    return x * mask


# ------------------------------------------------------------------------------------------ the DSP ends and the loss
def stft(wave):
    """infer.py:29-33 / m_dataset.py:187-190 on the library's kernel: wave [R, n] (cuda) -> [R, 2050, 1 + n // 1024]; no gradient
    (the mixture and the target are data)."""
    R, n = wave.shape
    dev = wave.device
    w = _f32c(wave)
    with torch.cuda.device(dev):
        x = torch.empty((R, 2050, 1 + n // 1024), device=dev)
        _native.check(_lib.bsrnn_stft(_context(dev), _p(w), _p(x), R, n, _s(dev)))
    return x


class IstftFunction(torch.autograd.Function):
    """torch.istft of m_dataset.py:192-195 ([R, 2050, T] interleaved -> [R, (T-1)*1024]) with the library's kernel forward and
    its transpose (bsrnn_istft_backward) backward."""

    @staticmethod
    def forward(ctx, y):
        R, F2, T = y.shape
        dev = y.device
        yc = _f32c(y)
        with torch.cuda.device(dev):
            out = torch.empty((R, (T - 1) * 1024), device=dev)
            _native.check(_lib.bsrnn_istft(_context(dev), _p(yc), _p(out), R, T, _s(dev)))
        ctx.shape = (R, F2, T)
        return out

    @staticmethod
    def backward(ctx, dwave):
        R, F2, T = ctx.shape
        dev = dwave.device
        g = _f32c(dwave)
        with torch.cuda.device(dev):
            dy = torch.empty((R, F2, T), device=dev)
            _native.check(_lib.bsrnn_istft_backward(_context(dev), _p(g), _p(dy), R, T, _s(dev)))
        return dy


def train_loss(model, mix, speech):
    """`train_infer` of the reference without the discriminator (m_dataset.py:182-216): STFT -> model -> iSTFT and the L1
    tri-loss  L1(x_time, speech_time) + L1(Re X, Re S) + L1(Im X, Im S)  (nn.L1Loss, mean).  mix, speech [R, n] on the GPU.
    Returns (loss, x_time); loss.backward() runs the library's backward kernels for every parameterised layer and the iSTFT."""
    x = stft(mix)
    y = forward_train(model, x)
    x_time = IstftFunction.apply(y)
    s = stft(speech)
    s_time = speech[:, :x_time.shape[1]]
    l1 = lambda a, b: (a - b).abs().mean()                      # noqa: E731
    loss = l1(x_time, s_time) + l1(y[:, 0::2, :], s[:, 0::2, :]) + l1(y[:, 1::2, :], s[:, 1::2, :])
    return loss, x_time


def sdr(x_time, s_time):
    """`sdr` of train_infer (m_dataset.py:217-220): mean over the rows of 10 log10((sum s^2 + 1e-9) / (sum (x - s)^2 + 1e-9));
    `train.py --loss_sdr` back-propagates its negative (train.py:100-101)."""
    n2s = torch.sum(torch.square(s_time), dim=1) + 1e-9
    n2d = torch.sum(torch.square(x_time - s_time), dim=1) + 1e-9
    return (10 * torch.log10(n2s / n2d)).mean()


# ------------------------------------------------------------------------------------------ optimizer and the step
class AdamW:
    """torch.optim.AdamW(params, lr=0.001, weight_decay=0.01) of train.py:50 on the library's kernel (all tensors of a step in one
    launch).  `state_dict()` / `load_state_dict()` speak torch.optim.AdamW's own format - {'state': {index: {'step', 'exp_avg',
    'exp_avg_sq'}}, 'param_groups': [...]}, indices in `model.parameters()` order with the zero-size parameters counted - so the
    reference's `optimizer.pth` / `optimizer-always.pth` (train.py:169-172) and the files written here interchange."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        """capturable (as torch.optim.AdamW's flag): the step count and the learning rate live in device memory
        (bsrnn_adamw_step_multi_dev), so that step() can be captured into a hipGraph with the rest of an iteration (GraphedTrainStep);
        every parameter must then have a gradient in every step (one step count for all, as in the reference's loop)."""
        self.capturable = bool(capturable)
        self._state_dev = None                              # capturable: {lr, bc1, bc2s, step} on the device
        self.all_params = list(params)                      # model.parameters() order = the indices of torch's state_dict
        self.slot = [i for i, p in enumerate(self.all_params) if p.numel() > 0]      # the ones the kernel updates
        self.params = [self.all_params[i] for i in self.slot]
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.steps = [0] * len(self.all_params)             # per parameter, like torch (a parameter without a gradient is not stepped)
        self._cache = {}

    @property
    def t(self):
        return max(self.steps) if self.steps else 0

    def zero_grad(self):
        for p in self.all_params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        """Every tensor that has a gradient, in one launch per distinct step count (bsrnn_adamw_step_multi; one launch normally)."""
        for i, p in enumerate(self.all_params):
            if p.grad is not None:
                self.steps[i] += 1
        idx_all = [j for j, p in enumerate(self.params) if p.grad is not None]
        if not idx_all:
            return
        dev = self.params[idx_all[0]].device
        if self.capturable:
            self._step_capturable(idx_all, dev)
            return
        for t in sorted({self.steps[self.slot[j]] for j in idx_all}):
            idx = [j for j in idx_all if self.steps[self.slot[j]] == t]
            grads = [_f32c(self.params[j].grad) for j in idx]
            n = len(idx)
            key = tuple(idx)
            if key not in self._cache:                       # the static arrays (parameters, moments, sizes) are built once
                arr = lambda ts: (ctypes.c_void_p * n)(*[ts[j].data_ptr() for j in idx])      # noqa: E731
                self._cache = {key: (arr(self.params), arr(self.m), arr(self.v), (ctypes.c_int64 * n)(*[self.params[j].numel() for j in idx]))}
            ap, am, av, sizes = self._cache[key]
            ag = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
            with torch.cuda.device(dev):
                _native.check(_lib.bsrnn_adamw_step_multi(_context(dev), ap, ag, am, av, sizes, n, self.lr, self.betas[0], self.betas[1],
                                                          self.eps, self.weight_decay, t, _s(dev)))
        for j in idx_all:  # the kernel wrote the parameters behind torch's back: bump their version counters (BSRNN re-uploads
            torch.autograd.graph.increment_version(self.params[j])      # its inference weights when a version changes)

    def _device_state(self, dev, steps_so_far):
        """{lr, bc1, bc2s, step} for bsrnn_adamw_step_multi_dev; (re)written whenever the host-side step count or lr was changed
        behind it (load_state_dict, set_lr)."""
        if self._state_dev is None or self._state_dev.device != dev:
            self._state_dev = torch.zeros(4, device=dev)
            self._state_host = None
        want = (float(self.lr), int(steps_so_far))
        if self._state_host != want:
            self._state_dev[0] = want[0]
            self._state_dev.view(torch.int32)[3] = want[1]
            self._state_host = want
        return self._state_dev

    def set_lr(self, lr):
        """Learning rate of the next steps (a schedule); in capturable mode a 4-byte write the captured step reads from the device."""
        self.lr = float(lr)
        if self._state_dev is not None:
            self._state_dev[0] = self.lr
            self._state_host = (self.lr, self._state_host[1]) if self._state_host else None

    def _step_capturable(self, idx_all, dev):
        if len(idx_all) != len(self.params) or len({self.steps[i] for i in self.slot}) != 1:
            raise RuntimeError("capturable AdamW: every parameter needs a gradient in every step (one step count for all)")
        t = self.steps[self.slot[0]]                         # already counted for this step
        state = self._device_state(dev, t - 1)
        n = len(idx_all)
        key = ("dev",) + tuple(idx_all)
        if key not in self._cache:
            arr = lambda ts: (ctypes.c_void_p * n)(*[ts[j].data_ptr() for j in idx_all])      # noqa: E731
            self._cache = {key: (arr(self.params), arr(self.m), arr(self.v), (ctypes.c_int64 * n)(*[self.params[j].numel() for j in idx_all]))}
        ap, am, av, sizes = self._cache[key]
        grads = [_f32c(self.params[j].grad) for j in idx_all]
        ag = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
        with torch.cuda.device(dev):
            _native.check(_lib.bsrnn_adamw_step_multi_dev(_context(dev), ap, ag, am, av, sizes, n, _p(state), self.betas[0], self.betas[1],
                                                          self.eps, self.weight_decay, _s(dev)))
        self._state_host = (self._state_host[0], t)          # the device counted this step
        for j in idx_all:
            torch.autograd.graph.increment_version(self.params[j])

    def note_replayed_step(self):
        """Book-keeping of one replay of a captured step(): the device advanced its own step count and wrote the parameters."""
        for i in self.slot:
            self.steps[i] += 1
        if self._state_host:
            self._state_host = (self._state_host[0], self._state_host[1] + 1)
        for p in self.params:
            torch.autograd.graph.increment_version(p)

    def _group_template(self):
        """param_groups[0] with every key this torch version's AdamW carries (its load_state_dict adopts the saved group as is)."""
        g = dict(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=self.lr, betas=self.betas, eps=self.eps,
                                   weight_decay=self.weight_decay).state_dict()["param_groups"][0])
        g["params"] = list(range(len(self.all_params)))
        return g

    def state_dict(self):
        """torch.optim.AdamW.state_dict() of the same optimizer: what train.py:169-172 saves."""
        state = {}
        pos = {i: j for j, i in enumerate(self.slot)}
        for i, p in enumerate(self.all_params):
            if self.steps[i] == 0:
                continue                                     # torch creates a parameter's state at its first step
            if i in pos:
                m, v = self.m[pos[i]].detach().cpu().clone(), self.v[pos[i]].detach().cpu().clone()
            else:
                m, v = torch.zeros_like(p.detach().cpu()), torch.zeros_like(p.detach().cpu())
            state[i] = {"step": torch.tensor(float(self.steps[i])), "exp_avg": m, "exp_avg_sq": v}
        return {"state": state, "param_groups": [self._group_template()]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW.state_dict() (the reference's optimizer.pth) or what state_dict() above returns."""
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("expected a torch.optim.AdamW state_dict ({'state': ..., 'param_groups': [...]})")
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.all_params):
            raise ValueError("optimizer state has %d parameter group(s) / %d parameters, the model has 1 / %d" % (
                len(groups), len(groups[0]["params"]) if groups else 0, len(self.all_params)))
        g = groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            raise ValueError("amsgrad / maximize optimizer states are not supported")
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"])
        order = list(g["params"])                            # saved index of the parameter at each position
        pos = {i: j for j, i in enumerate(self.slot)}
        self.steps = [0] * len(self.all_params)
        for m in self.m:
            m.zero_()
        for v in self.v:
            v.zero_()
        for i, saved in enumerate(order):
            st = sd["state"].get(saved, sd["state"].get(str(saved)))
            if st is None:
                continue
            self.steps[i] = int(round(float(st["step"])))
            if i in pos:
                if tuple(st["exp_avg"].shape) != tuple(self.params[pos[i]].shape):
                    raise ValueError("optimizer state of parameter %d has shape %s, the model's is %s" % (
                        i, tuple(st["exp_avg"].shape), tuple(self.params[pos[i]].shape)))
                self.m[pos[i]].copy_(st["exp_avg"])
                self.v[pos[i]].copy_(st["exp_avg_sq"])
        self._cache = {}
        if self._state_dev is not None:                      # capturable: the device's step count follows the loaded one
            self._device_state(self._state_dev.device, self.t)


def train_step(model, optimizer, mix, speech, group=None, loss_sdr=False):
    """One iteration of the reference's loop (train.py:97-115 with batch_size 1): loss, backward (of the L1 tri-loss, or of -SDR with
    loss_sdr as `train.py --loss_sdr`), optimizer step, zero_grad.  Under torch.distributed (one process per GPU, backend nccl = RCCL) every rank passes its own clip and the gradients are
    averaged over the ranks in a few large buckets before the step (dist.all_reduce_gradients): data-parallel training."""
    loss, x_time = train_loss(model, mix, speech)
    (-sdr(x_time, speech[:, :x_time.shape[1]]) if loss_sdr else loss).backward()
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size(group) > 1:
        from .dist import all_reduce_gradients
        all_reduce_gradients(model.parameters(), group=group)
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach()


class GraphedTrainStep:
    """`train_step` with the whole iteration - STFTs, forward, loss, backward, AdamW, zero_grad - captured ONCE into a hipGraph and
    replayed: one graph launch per iteration instead of ~ 400 kernel launches driven by Python and autograd (the reference's own
    configuration, batch_size 1 = two rows, is bound by exactly that).  Fixed clip shape [rows, samples]; single process (the
    data-parallel all-reduce is not captured: use train_step under torch.distributed); the optimizer must be
    AdamW(..., capturable=True).  The first `warmup` calls run eagerly on the capture stream (they are ordinary training steps
    and size the library's scratch buffers, which must not grow during capture); the next call captures, every call from then on
    replays.  Results are those of train_step (same kernels, same order).

        step = GraphedTrainStep(model, optimizer, rows=2, samples=128000)
        for mix, speech in loader: loss = step(mix, speech)
    """

    def __init__(self, model, optimizer, rows, samples, loss_sdr=False, warmup=2):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs AdamW(..., capturable=True): the step count must live on the device")
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep is single-process; use train_step under torch.distributed")
        self.model, self.optimizer, self.loss_sdr = model, optimizer, bool(loss_sdr)
        self.device = next(p for p in model.parameters() if p.numel() > 0).device
        if self.device.type != "cuda":
            raise ValueError("the training kernels run on the GPU: move the model to a cuda device first")
        self.mix = torch.zeros((rows, samples), device=self.device)
        self.speech = torch.zeros((rows, samples), device=self.device)
        self.stream = torch.cuda.Stream(self.device)
        self.warmup, self.calls = max(1, int(warmup)), 0     # at least one eager step: scratch and workspaces must exist before capture
        self.graph, self.loss, self.sdr = None, None, None
        self.last_sdr = None                                  # `sdr` of train_infer for the last clip (train.py prints its epoch mean)

    def _iteration(self):
        loss, x_time = train_loss(self.model, self.mix, self.speech)
        s = sdr(x_time, self.speech[:, :x_time.shape[1]])
        (-s if self.loss_sdr else loss).backward()
        self.optimizer.step()
        self.optimizer.zero_grad()
        return loss.detach(), s.detach()

    def __call__(self, mix, speech):
        if tuple(mix.shape) != tuple(self.mix.shape) or tuple(speech.shape) != tuple(self.speech.shape):
            raise ValueError("GraphedTrainStep was built for clips of shape %s" % (tuple(self.mix.shape),))
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.mix.copy_(mix, non_blocking=True)
            self.speech.copy_(speech, non_blocking=True)
            if self.graph is None and self.calls < self.warmup:
                out, self.last_sdr = (t.clone() for t in self._iteration())
            else:
                if self.graph is None:
                    self.optimizer._device_state(self.device, self.optimizer.t)    # exists and is current BEFORE capture (no captured writes)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.stream):
                        self.loss, self.sdr = self._iteration()
                    self.graph = g
                    # capture ran the Python side of optimizer.step() once without executing anything: take its count back,
                    # the replay below is the step
                    for i in self.optimizer.slot:
                        self.optimizer.steps[i] -= 1
                    self.optimizer._state_host = (self.optimizer._state_host[0], self.optimizer._state_host[1] - 1)
                self.graph.replay()
                self.optimizer.note_replayed_step()
                out, self.last_sdr = self.loss.clone(), self.sdr.clone()
        self.calls += 1
        cur.wait_stream(self.stream)
        return out
