"""Host-side mirror of the reference's `bsrnn.py` operator interface, running on libbsrnn_hip.

Same surface as the reference model class (bsrnn.py:328-510):
    BSRNN()                                   no constructor arguments (band table optional here)
    .to(device) / .eval() / .state_dict() / .load_state_dict(sd)   -- identical key names and
                                              shapes (288 tensors)
    .forward(x [C,2050,T]) -> [C,2050,T]      bsrnn.py:385
    .forward_recurrent(x [C,2050], state [4,2,C*K,64]) -> (y, new_state)   bsrnn.py:445
    generate_bandsplits(), band_features, merge_channels module globals    bsrnn.py:247, :60-61
plus the fused device-side sandwich of the callers (`separate`, `stft`, `istft`) and the chunked
streaming form.  The torch modules below are *parameter containers only* (they give the
state_dict its names, shapes and default initialisation); their own forward() is never
called -- all arithmetic happens in hand-written HIP kernels behind the C ABI, and without
the built library this module cannot be imported (no CPU fallback).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _native
from . import spec as _spec
from .spec import generate_bandsplits  # noqa: F401  (re-export, bsrnn.py:247)

band_features = _spec.BAND_FEATURES      # bsrnn.py:60
merge_channels = _spec.MERGE_CHANNELS    # bsrnn.py:61
_lib = _native.lib
_check = _native.check


class TrainableConstantModule(nn.Module):
    """Parameter container for the zero-width band's learned constant (bsrnn.py:12-24)."""

    def __init__(self, shape):
        super().__init__()
        self.trainable_constant = nn.Parameter(torch.zeros(shape, dtype=torch.float32))


def _mlp(dims, act_after_last):
    """Sequential whose even slots hold Linear(dims[i] -> dims[i+1]) so that the parameter
    names come out as `<i*2>.weight/.bias` like the reference's Sequentials."""
    mods = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or act_after_last:
            mods.append(nn.LeakyReLU())
    return nn.Sequential(*mods)


class _RNNBlockParams(nn.Module):
    """fc_in / rnn / fc of NormRNNResidual (bsrnn.py:63-76), as containers."""

    def __init__(self, bidirectional):
        super().__init__()
        H = band_features
        self.fc_in = nn.Linear(H, H)
        self.rnn = nn.LSTM(H, H, batch_first=True, num_layers=2, bidirectional=bidirectional)
        self.fc = nn.Linear(2 * H if bidirectional else H, H)


class _Holder(nn.Module):
    def __init__(self, bidirectional):
        super().__init__()
        self.m = _RNNBlockParams(bidirectional)


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class BSRNN(nn.Module):
    def __init__(self, band_widths=None):
        super().__init__()
        v = list(band_widths) if band_widths is not None else generate_bandsplits()[0]
        self.band_widths = v
        H, MH = band_features, _spec.MASK_HIDDEN
        self.bandFCs_pre = nn.ModuleList(
            [_mlp([2 * w, 2 * w, 2 * w], True) if w > 0 else nn.Sequential(TrainableConstantModule([0])) for w in v])
        self.bandFCs = nn.ModuleList(
            [_mlp([2 * w, max(2 * w, H), H, H], False) if w > 0 else nn.Sequential(TrainableConstantModule([H])) for w in v])
        self.lstms = nn.Sequential(_Holder(True), _Holder(False), _Holder(True), _Holder(False))
        self.bandFCs_back = nn.ModuleList(
            [_mlp([H, MH, max(2 * w, MH), 2 * w], True) if w > 0 else nn.Sequential(TrainableConstantModule(0)) for w in v])
        self.bandFCs_back_post = nn.ModuleList(
            [_mlp([2 * w, 2 * w, 2 * w], False) if w > 0 else nn.Sequential(TrainableConstantModule([0])) for w in v])
        self._ctx = None
        self._ctx_device = None
        self._pushed_fingerprint = None
        self._plist = None                  # cached parameter tensors (the module tree is fixed after construction)
        self._pident = ()                   # their (id, data_ptr) pairs when the list was built
        self._epoch = 0                     # bumped by everything that may rebind parameter storage (_apply, load_state_dict)
        self._range_policy = _native.RANGE_EXACT

    # ------------------------------------------------------------------ native context / weights
    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.bsrnn_destroy(self._ctx)
        except Exception:
            pass

    def _fingerprint(self):
        """Cheap identity of the current weights: every in-place edit of a parameter bumps its tensor's `_version`; `.to()` /
        `.cuda()` / `.float()` (all through `_apply`) and `load_state_dict` bump `_epoch`; rebinding a parameter's storage
        (`p.data = t`, `module.weight = Parameter(...)`, `swap_tensors`) changes the (id, data_ptr) pairs taken when the
        parameter list is (re)built.  One attribute read per parameter per check."""
        if self._plist is None:
            self._plist = list(self.parameters())
            self._pident = tuple((id(p), p.data_ptr()) for p in self._plist)
        return (self._epoch, self._pident) + tuple(p._version for p in self._plist)

    def _weights_touched(self, k, n=8):
        """Slice k of n of the cached parameter list: has any of them been edited in place or had its storage rebound since
        the last upload?  (The streaming wrapper calls this every step with a rotating k: an optimizer step or any other
        edit of ALL parameters is seen at the very next chunk, an edit of ONE tensor within n chunks, for ~3 us per step.)"""
        fp = self._pushed_fingerprint
        if fp is None or self._plist is None:
            return True
        pl, ident, vers = self._plist, fp[1], fp[2:]
        for i in range(k % n, len(pl), n):
            p = pl[i]
            if p._version != vers[i] or p.data_ptr() != ident[i][1]:
                return True
        return False

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._plist = None
        self._epoch += 1
        self._pushed_fingerprint = None     # (also tells a running StreamingSeparator to look at the weights at its next step)
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._plist = None
        self._epoch += 1
        self._pushed_fingerprint = None
        return out

    def set_range_policy(self, policy):
        """'exact' (default): every call waits for its kernels and, if an activation left the fp16x2 range (|a| > 65504),
        is run again on the exact-fp32 kernels before it returns - results always match the reference's forward.
        'deferred': calls return without waiting (benchmark loops, launch pipelines); a violation surfaces as a NativeError
        at the next call on this model (include/bsrnn_hip.h, bsrnn_set_range_policy)."""
        self._range_policy = {"exact": _native.RANGE_EXACT, "deferred": _native.RANGE_DEFERRED}[policy]
        if self._ctx is not None:
            _check(_lib.bsrnn_set_range_policy(self._ctx, self._range_policy))

    def _context(self, device):
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        if self._ctx is None or self._ctx_device != dev_index:
            if self._ctx is not None:
                _lib.bsrnn_destroy(self._ctx)
                self._ctx = None
            widths = (ctypes.c_int32 * len(self.band_widths))(*self.band_widths)
            ctx = ctypes.c_void_p()
            _check(_lib.bsrnn_create(dev_index, widths, len(self.band_widths), ctypes.byref(ctx)))
            self._ctx, self._ctx_device, self._pushed_fingerprint = ctx, dev_index, None
            _check(_lib.bsrnn_set_range_policy(ctx, self._range_policy))
        fp = self._fingerprint()
        if fp != self._pushed_fingerprint:
            for key, val in self.state_dict().items():
                a = np.ascontiguousarray(val.detach().to("cpu", torch.float32).numpy())
                _check(_lib.bsrnn_set_param(self._ctx, key.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size))
            _check(_lib.bsrnn_commit_params(self._ctx))
            self._pushed_fingerprint = fp
        return self._ctx

    def mlp_flow(self, device=None):
        """'fused' (one launch per MLP chain, the default) or 'layers' (one grouped launch per Linear layer) on `device`."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return "fused" if _lib.bsrnn_mlp_fused(self._context(dev)) == 1 else "layers"

    def overlap_state(self, device=None):
        """How this model's context runs the dual path of large calls: 1 overlapped (default), 0 launch after launch (BSRNN_OVERLAP=0),
        2 switched off after a consumer's wait expired (include/bsrnn_hip.h, bsrnn_overlap_state)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return _lib.bsrnn_overlap_state(self._context(dev))

    def sync(self, device=None):
        """Wait for this model's work on the current stream of `device`; raises if a call made under the 'deferred' range
        policy left the fp16x2 range (bsrnn_sync)."""
        dev = torch.device("cuda", self._ctx_device if self._ctx_device is not None else torch.cuda.current_device()) if device is None else torch.device(device)
        if self._ctx is not None:
            with torch.cuda.device(dev):
                _check(_lib.bsrnn_sync(self._ctx, _stream_ptr(dev)))

    def refresh_weights(self):
        """Force re-upload (only needed after edits that bypass tensor versioning, e.g. `p.data = other`)."""
        self._pushed_fingerprint = None
        self._plist = None

    @staticmethod
    def _device_for(x):
        if x.is_cuda:
            return x.device
        if not torch.cuda.is_available():
            raise _native.NativeError("BSRNN needs a HIP device: no GPU visible and there is no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())

    @staticmethod
    def _prep(x, dev):
        return x.detach().to(device=dev, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ reference interface
    def forward(self, x):
        """bsrnn.py:385: x [C, 2050, T] (re/im interleaved STFT) -> x * mask, same shape."""
        return self._forward(x, want_mask=False)[0]

    def forward_with_mask(self, x):
        """forward plus the mask itself (bsrnn.py:425-432), for parity checks."""
        return self._forward(x, want_mask=True)

    def _forward(self, x, want_mask):
        if x.dim() != 3 or x.shape[1] != _spec.N_BINS * 2:
            raise ValueError("expected x [C, 2050, T], got %s" % (tuple(x.shape),))
        dev = self._device_for(x)
        xd = self._prep(x, dev)
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            y = torch.empty_like(xd)
            mask = torch.empty_like(xd) if want_mask else None
            _check(_lib.bsrnn_forward(ctx, _ptr(xd), _ptr(y), _ptr(mask) if want_mask else None,
                                      xd.shape[0], xd.shape[2], _stream_ptr(dev)))
        if not x.is_cuda:
            y = y.cpu()
            mask = mask.cpu() if want_mask else None
        return y, mask

    def forward_recurrent(self, x, state):
        """bsrnn.py:445: one frame.  x [C, 2050], state [4, 2, C*K, 64] -> (y, new_state)."""
        if x.dim() != 2 or x.shape[1] != _spec.N_BINS * 2:
            raise ValueError("expected x [C, 2050], got %s" % (tuple(x.shape),))
        y, ns = self.forward_chunk(x.unsqueeze(2), state)
        return y.squeeze(2), ns

    def forward_chunk(self, x, state):
        """L consecutive frames with causal state carry: x [C, 2050, L]."""
        C, _, L = x.shape
        K = len(self.band_widths)
        if tuple(state.shape) != (4, 2, C * K, band_features):
            raise ValueError("expected state [4, 2, %d, 64], got %s" % (C * K, tuple(state.shape)))
        dev = self._device_for(x)
        xd, sd = self._prep(x, dev), self._prep(state, dev)
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            y, ns = torch.empty_like(xd), torch.empty_like(sd)
            _check(_lib.bsrnn_forward_chunk(ctx, _ptr(xd), _ptr(sd), _ptr(y), _ptr(ns), C, L, _stream_ptr(dev)))
        if not x.is_cuda:
            y, ns = y.cpu(), ns.cpu()
        return y, ns

    def dual_path(self, z, state=None):
        """self.lstms(z) (bsrnn.py:417): z [C, T, K, 64] -> (z_out, new_state)."""
        C, T, K, Hh = z.shape
        if K != len(self.band_widths) or Hh != band_features:
            raise ValueError("expected z [C, T, %d, 64]" % len(self.band_widths))
        dev = self._device_for(z)
        zd = self._prep(z, dev)
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            out = torch.empty_like(zd)
            sd = self._prep(state, dev) if state is not None else torch.zeros((4, 2, C * K, Hh), device=dev)
            ns = torch.empty_like(sd)
            _check(_lib.bsrnn_dual_path(ctx, _ptr(zd), _ptr(out), _ptr(sd), _ptr(ns), C, T, _stream_ptr(dev)))
        if not z.is_cuda:
            out, ns = out.cpu(), ns.cpu()
        return out, ns

    # ------------------------------------------------------------------ the callers' sandwich, on device
    def stft(self, waveform):
        """infer.py:29-33: [R, n] -> [R, 2050, 1 + n//1024]."""
        dev = self._device_for(waveform)
        w = self._prep(waveform, dev)
        R, n = w.shape
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            x = torch.empty((R, 2 * _spec.N_BINS, 1 + n // _spec.HOP), device=dev, dtype=torch.float32)
            _check(_lib.bsrnn_stft(ctx, _ptr(w), _ptr(x), R, n, _stream_ptr(dev)))
        return x if waveform.is_cuda else x.cpu()

    def istft(self, y):
        """infer.py:35-37: [R, 2050, T] -> [R, (T-1)*1024]."""
        dev = self._device_for(y)
        yd = self._prep(y, dev)
        R, _, T = yd.shape
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            out = torch.empty((R, (T - 1) * _spec.HOP), device=dev, dtype=torch.float32)
            _check(_lib.bsrnn_istft(ctx, _ptr(yd), _ptr(out), R, T, _stream_ptr(dev)))
        return out if y.is_cuda else out.cpu()

    def separate(self, waveform, out=None):
        """STFT -> forward -> iSTFT fused on the device: [R, n] -> [R, (n//1024)*1024]."""
        dev = self._device_for(waveform)
        w = self._prep(waveform, dev)
        R, n = w.shape
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            if out is None:
                out = torch.empty((R, (n // _spec.HOP) * _spec.HOP), device=dev, dtype=torch.float32)
            _check(_lib.bsrnn_separate(ctx, _ptr(w), _ptr(out), R, n, _stream_ptr(dev)))
        return out if waveform.is_cuda else out.cpu()

    def evaluate(self, mix, speech, return_estimate=False):
        """The reference's validation arithmetic on the device (m_dataset.py:182-226 `infer` + `train_infer` without
        the discriminator, and the "Separation dB" of infer.py:44-47): mix, speech [R, n] -> dict of
        loss / sdr / input_sdr / sisdr / l1_time / l1_re / l1_im / separation_db (see include/bsrnn_hip.h)."""
        if mix.dim() != 2 or tuple(mix.shape) != tuple(speech.shape):
            raise ValueError("expected mix and speech [R, n] of the same shape, got %s and %s" % (tuple(mix.shape), tuple(speech.shape)))
        dev = self._device_for(mix)
        m, s = self._prep(mix, dev), self._prep(speech, dev)
        R, n = m.shape
        vals = (ctypes.c_double * len(_native.METRIC_NAMES))()
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            est = torch.empty((R, (n // _spec.HOP) * _spec.HOP), device=dev, dtype=torch.float32) if return_estimate else None
            _check(_lib.bsrnn_evaluate(ctx, _ptr(m), _ptr(s), R, n, _ptr(est) if return_estimate else None, vals, _stream_ptr(dev)))
        out = {k: vals[i] for i, k in enumerate(_native.METRIC_NAMES)}
        if return_estimate:
            out["x_time"] = est if mix.is_cuda else est.cpu()
        return out

    # ------------------------------------------------------------------ measurement
    def set_profiling(self, on, device=None):
        """on: False/True (all stages) or an iterable of stage names to bracket with events."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if on is True:
            mask = -1
        elif not on:
            mask = 0
        else:
            names = _native.stage_names()
            mask = 0
            for s in on:
                mask |= 1 << names.index(s)
        _check(_lib.bsrnn_set_profiling(self._context(dev), mask))

    def stage_times(self, reset=True):
        """-> {stage: (total_ms, launches)} accumulated while profiling was on."""
        n = _lib.bsrnn_stage_count()
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        _check(_lib.bsrnn_stage_times(self._ctx, ms, cnt, 1 if reset else 0))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_native.stage_names())}

    def save_flat(self, path):
        from . import weights
        weights.save_flat(path, {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}, self.band_widths)


class StreamingSeparator:
    """Device-resident form of the infer-streaming.py loop (lines 84-147): feed [C, 1024] chunks,
    get [C, 1024] chunks delayed by one hop.  State, the sliding buffer and the previous
    synthesis frame stay on the GPU between steps."""

    def __init__(self, model, channels=2, device=None):
        self.model = model
        self.C = channels
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ctx = model._context(self.device)
        h = ctypes.c_void_p()
        _check(_lib.bsrnn_stream_create(ctx, channels, ctypes.byref(h)))
        self._h = h
        self._steps = 0

    def __del__(self):
        try:
            _lib.bsrnn_stream_destroy(self._h)
        except Exception:
            pass

    def reset(self):
        _check(_lib.bsrnn_stream_reset(self._h, _stream_ptr(self.device)))

    def step(self, chunk, mix=1.0):
        """chunk [C, 1024] float32 (cuda or cpu) -> same-shaped output on the same device."""
        if tuple(chunk.shape) != (self.C, _spec.HOP):
            raise ValueError("expected chunk [%d, 1024], got %s" % (self.C, tuple(chunk.shape)))
        # The reference's forward_recurrent always sees the current parameters (bsrnn.py:445).  Here the weights live packed on the
        # device, so a running stream looks at the model every step, cheaply: a rotating eighth of the parameters' version counters
        # and storage pointers per step (an optimizer step or load_state_dict is seen at the next chunk, an in-place edit or
        # rebinding of a single tensor within 8 chunks), and the whole module tree again every 32nd step (a parameter OBJECT
        # swapped into a submodule).  model.refresh_weights() forces it at once.
        if self._steps % 32 == 0 or self.model._weights_touched(self._steps):
            self.model._plist = None            # (re-read the parameter objects and their storage pointers)
            with torch.cuda.device(self.device):
                self.model._context(self.device)
        self._steps += 1
        if chunk.is_cuda:
            c = chunk.detach().to(torch.float32).contiguous()
            out = torch.empty_like(c)
            with torch.cuda.device(self.device):
                _check(_lib.bsrnn_stream_step(self._h, _ptr(c), _ptr(out), float(mix), _stream_ptr(self.device)))
            return out
        c = np.ascontiguousarray(chunk.detach().numpy(), dtype=np.float32)
        o = np.empty_like(c)
        _check(_lib.bsrnn_stream_step_host(self._h, c.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), float(mix)))
        return torch.from_numpy(o)

    def state(self):
        K = len(self.model.band_widths)
        s = np.empty((4, 2, self.C * K, band_features), np.float32)
        _check(_lib.bsrnn_stream_get_state(self._h, s.ctypes.data_as(ctypes.c_void_p)))
        return torch.from_numpy(s)
