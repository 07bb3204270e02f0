"""ORACLE -- test infrastructure, never the product path.

A plain-numpy CPU restatement of the reference's BSRNN separation path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this file;
the product (speechseparation_amd/) never does and fails loudly without its HIP library.

Pinning: the reference has no tests or golden vectors of its own, so this
restatement is pinned against outputs of the reference itself, run in the build
container by `tests/golden/make_golden.py` (imports /root/reference/bsrnn.py read-only
and restates infer.py:29-37 / infer-streaming.py:116-145 with stock torch ops) and
committed under `tests/golden/*.npz`.  `tests/test_oracle_golden.py` checks every function
here against those fixtures.

Each function cites the reference lines it follows.  All arithmetic is done in `dtype`
(float32 by default; float64 gives the tolerance-accounting copy).
"""
import numpy as np

H = 64                     # bsrnn.py:60 band_features
N_FFT = 2048               # infer.py:31
HOP = 1024
LEAK = 0.01                # nn.LeakyReLU() default slope


# --------------------------------------------------------------------------- band table
def generate_bandsplits():
    """bsrnn.py:247-326.  Integer bookkeeping, bit-exact."""
    pos, mul = 3, 2
    v = [(1, 0), (2, 1)]
    fft_size = 1025
    while pos < fft_size:
        n = int(pos * mul)
        if n == pos:
            n = n + 1
        d = n - pos
        v.append((d, pos))
        pos += d
    v.pop()
    pos = 0
    for x, y in v:
        assert y == pos
        pos += x
    v = [x[0] for x in v]
    if sum(v) != fft_size:
        v = v + [fft_size - sum(v)]
    v = v + [0]
    w = [v[i] + v[i + 1] for i in range(len(v) - 1)]
    w.append(v[0] + v[-1])
    return (v, w)


# --------------------------------------------------------------------------- primitives
def leaky(x):
    return np.where(x >= 0, x, x * x.dtype.type(LEAK))


def linear(x, w, b):
    """nn.Linear: y = x W^T + b, W stored [out, in]."""
    return x @ w.T + b


def sigmoid(x):
    one = x.dtype.type(1)
    return one / (one + np.exp(-x))


def _p(sd, key, dtype):
    return np.asarray(sd[key], dtype=dtype)


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None, reverse=False):
    """One direction of one nn.LSTM layer, batch_first.  x [N, L, in] -> out [N, L, 64].
    Gate order i, f, g, o (torch); c' = s(f) c + s(i) tanh(g); h' = s(o) tanh(c')
    (called from bsrnn.py:83 / :94)."""
    N, L, _ = x.shape
    dt = x.dtype
    h = np.zeros((N, H), dt) if h0 is None else h0.astype(dt).copy()
    c = np.zeros((N, H), dt) if c0 is None else c0.astype(dt).copy()
    out = np.empty((N, L, H), dt)
    gx = x @ w_ih.T + (b_ih + b_hh)                 # [N, L, 4H]
    order = range(L - 1, -1, -1) if reverse else range(L)
    for t in order:
        g = gx[:, t, :] + h @ w_hh.T
        i = sigmoid(g[:, 0:H])
        f = sigmoid(g[:, H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = sigmoid(g[:, 3 * H:4 * H])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t, :] = h
    return out, h, c


def norm_rnn_residual(sd, prefix, x, bidir, state=None, dtype=np.float32):
    """NormRNNResidual.forward / forward_recurrent, bsrnn.py:78-98 (groupnorm is None).
    x [N, L, 64]; state = (h0, c0) each [2 layers, N, 64] (unidirectional only)."""
    p = prefix + "m."
    u = linear(x, _p(sd, p + "fc_in.weight", dtype), _p(sd, p + "fc_in.bias", dtype))
    hs, cs = [], []
    inp = u
    for layer in range(2):
        outs = []
        for suffix in (("", "_reverse") if bidir else ("",)):
            h0 = c0 = None
            if state is not None:
                h0, c0 = state[0][layer], state[1][layer]
            o, hT, cT = lstm_layer(
                inp,
                _p(sd, p + "rnn.weight_ih_l%d%s" % (layer, suffix), dtype),
                _p(sd, p + "rnn.weight_hh_l%d%s" % (layer, suffix), dtype),
                _p(sd, p + "rnn.bias_ih_l%d%s" % (layer, suffix), dtype),
                _p(sd, p + "rnn.bias_hh_l%d%s" % (layer, suffix), dtype),
                h0, c0, reverse=(suffix != ""))
            outs.append(o)
            hs.append(hT)
            cs.append(cT)
        inp = np.concatenate(outs, axis=2) if bidir else outs[0]
    out = linear(inp, _p(sd, p + "fc.weight", dtype), _p(sd, p + "fc.bias", dtype)) + x
    return out, (np.stack(hs, 0), np.stack(cs, 0))


def bandwise(sd, j, z, dtype=np.float32):
    """BandwiseLSTM.forward, bsrnn.py:138-153: [C,T,K,64] -> N=C*T sequences of length K."""
    C, T, K, _ = z.shape
    out, _ = norm_rnn_residual(sd, "lstms.%d." % j, z.reshape(C * T, K, H), True, None, dtype)
    return out.reshape(C, T, K, H)


def timewise(sd, j, z, state=None, dtype=np.float32):
    """TimewiseLSTM.forward / forward_recurrent, bsrnn.py:106-128: N=C*K sequences of length T
    (row order c*K+k), causal; state slab [2(h,c), 2 layers, C*K, 64]."""
    C, T, K, _ = z.shape
    x = np.ascontiguousarray(z.transpose(0, 2, 1, 3)).reshape(C * K, T, H)
    st = None if state is None else (state[0], state[1])
    out, (hT, cT) = norm_rnn_residual(sd, "lstms.%d." % j, x, False, st, dtype)
    out = out.reshape(C, K, T, H).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(out), np.stack((hT, cT), 0)


def dual_path(sd, z, state=None, dtype=np.float32, taps=None):
    """self.lstms = Band, Time, Band, Time (bsrnn.py:352-356, :417 / :472-481).
    state [4, 2, C*K, 64] or None -> (z_out, new_state [4,2,C*K,64])."""
    new_state = []
    si = 0
    for j in range(4):
        if j % 2 == 0:
            z = bandwise(sd, j, z, dtype)
        else:
            st = None if state is None else state[2 * si:2 * si + 2]
            z, s = timewise(sd, j, z, st, dtype)
            new_state.append(s)
            si += 1
        if taps is not None:
            taps["z_after_%d" % j] = z.copy()
    return z, np.concatenate(new_state, 0)


# --------------------------------------------------------------------------- band MLPs
def _band_cols(v):
    off, pos = [], 0
    for x in v:
        off.append(2 * pos)
        pos += x
    return off


def band_split(sd, xt, v, dtype=np.float32):
    """bandFCs_pre + bandFCs, bsrnn.py:404-415.  xt [C,T,2050] -> (residual list, Z [C,T,K,64])."""
    C, T, _ = xt.shape
    cols = _band_cols(v)
    residual, feats = [], []
    for i, w in enumerate(v):
        a = 2 * w
        if w == 0:                                   # TrainableConstantModule, bsrnn.py:12-24
            residual.append(np.zeros((C, T, 0), dtype))
            const = _p(sd, "bandFCs.%d.0.trainable_constant" % i, dtype)
            feats.append(np.broadcast_to(const, (C, T, H)).copy())
            continue
        b = xt[:, :, cols[i]:cols[i] + a]
        y = leaky(linear(b, _p(sd, "bandFCs_pre.%d.0.weight" % i, dtype), _p(sd, "bandFCs_pre.%d.0.bias" % i, dtype)))
        y = leaky(linear(y, _p(sd, "bandFCs_pre.%d.2.weight" % i, dtype), _p(sd, "bandFCs_pre.%d.2.bias" % i, dtype)))
        residual.append(y)
        y = leaky(linear(y, _p(sd, "bandFCs.%d.0.weight" % i, dtype), _p(sd, "bandFCs.%d.0.bias" % i, dtype)))
        y = leaky(linear(y, _p(sd, "bandFCs.%d.2.weight" % i, dtype), _p(sd, "bandFCs.%d.2.bias" % i, dtype)))
        y = linear(y, _p(sd, "bandFCs.%d.4.weight" % i, dtype), _p(sd, "bandFCs.%d.4.bias" % i, dtype))
        feats.append(y)
    return residual, np.stack(feats, 2)


def mask_estimation(sd, z, residual, v, dtype=np.float32):
    """bandFCs_back + bandFCs_back_post + skip, bsrnn.py:420-430.  -> mask [C,T,2050]."""
    parts = []
    for i, w in enumerate(v):
        if w == 0:
            continue
        b = z[:, :, i, :]
        b = leaky(linear(b, _p(sd, "bandFCs_back.%d.0.weight" % i, dtype), _p(sd, "bandFCs_back.%d.0.bias" % i, dtype)))
        b = leaky(linear(b, _p(sd, "bandFCs_back.%d.2.weight" % i, dtype), _p(sd, "bandFCs_back.%d.2.bias" % i, dtype)))
        b = leaky(linear(b, _p(sd, "bandFCs_back.%d.4.weight" % i, dtype), _p(sd, "bandFCs_back.%d.4.bias" % i, dtype)))
        b = leaky(linear(b, _p(sd, "bandFCs_back_post.%d.0.weight" % i, dtype), _p(sd, "bandFCs_back_post.%d.0.bias" % i, dtype)))
        b = linear(b, _p(sd, "bandFCs_back_post.%d.2.weight" % i, dtype), _p(sd, "bandFCs_back_post.%d.2.bias" % i, dtype))
        parts.append(residual[i] + b)
    return np.concatenate(parts, 2)


# --------------------------------------------------------------------------- model entry points
def forward(sd, x, v=None, dtype=np.float32, taps=None):
    """BSRNN.forward, bsrnn.py:385-443.  x [C, 2050, T] (re/im interleaved) -> y same shape.
    `taps` (dict) receives mask, Z0 and Z after each dual-path block."""
    if v is None:
        v = generate_bandsplits()[0]
    x = np.asarray(x, dtype)
    xt = np.ascontiguousarray(x.transpose(0, 2, 1))             # [C,T,2050]  bsrnn.py:406
    residual, z = band_split(sd, xt, v, dtype)
    if taps is not None:
        taps["z0"] = z.copy()
    z, _ = dual_path(sd, z, None, dtype, taps)
    mask = mask_estimation(sd, z, residual, v, dtype)
    if taps is not None:
        taps["mask"] = np.ascontiguousarray(mask.transpose(0, 2, 1))
    return x * mask.transpose(0, 2, 1)                          # bsrnn.py:441


def forward_recurrent(sd, x, state, v=None, dtype=np.float32):
    """BSRNN.forward_recurrent, bsrnn.py:445-510.  x [C,2050], state [4,2,C*K,64]."""
    if v is None:
        v = generate_bandsplits()[0]
    x = np.asarray(x, dtype)
    state = np.asarray(state, dtype)
    xt = x[:, None, :]
    residual, z = band_split(sd, xt, v, dtype)
    z, new_state = dual_path(sd, z, state, dtype)
    mask = mask_estimation(sd, z, residual, v, dtype)
    return x * mask[:, 0, :], new_state


def forward_chunked(sd, x, state, v=None, dtype=np.float32):
    """L frames with causal state carry (BASELINE.json config 3): equals running
    forward_recurrent frame by frame, computed block-wise.  x [C,2050,L]."""
    if v is None:
        v = generate_bandsplits()[0]
    x = np.asarray(x, dtype)
    xt = np.ascontiguousarray(x.transpose(0, 2, 1))
    residual, z = band_split(sd, xt, v, dtype)
    z, new_state = dual_path(sd, z, np.asarray(state, dtype), dtype)
    mask = mask_estimation(sd, z, residual, v, dtype)
    return x * mask.transpose(0, 2, 1), new_state


# --------------------------------------------------------------------------- offline STFT sandwich
def hann_periodic(n=N_FFT, dtype=np.float32):
    """torch.hann_window(n) (periodic): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def stft_interleaved(wave, dtype=np.float32):
    """infer.py:29-33: torch.stft(n_fft=2048, hop=1024, hann, center=True, reflect, onesided)
    then stack(re, im) on a new dim 2 and reshape -> [C, 2050, T], column 2f=re, 2f+1=im."""
    wave = np.asarray(wave, dtype)
    C, n = wave.shape
    pad = N_FFT // 2
    xp = np.pad(wave, ((0, 0), (pad, pad)), mode="reflect")
    T = 1 + n // HOP
    win = hann_periodic(N_FFT, dtype)
    idx = np.arange(T)[:, None] * HOP + np.arange(N_FFT)[None, :]
    frames = xp[:, idx] * win                                  # [C,T,2048]
    X = np.fft.rfft(frames, axis=2)                            # [C,T,1025]
    out = np.empty((C, T, 2 * (N_FFT // 2 + 1)), dtype)
    out[:, :, 0::2] = X.real
    out[:, :, 1::2] = X.imag
    return np.ascontiguousarray(out.transpose(0, 2, 1))


def istft_interleaved(y, dtype=np.float32):
    """infer.py:35-37: un-interleave, torch.istft(same window): irfft, * window, overlap-add,
    / sum(window^2), drop n_fft/2 at both ends -> [C, (T-1)*1024]."""
    y = np.asarray(y, dtype)
    C, F2, T = y.shape
    yt = y.transpose(0, 2, 1)
    Y = yt[:, :, 0::2] + 1j * yt[:, :, 1::2]
    frames = np.fft.irfft(Y, n=N_FFT, axis=2).astype(dtype)     # [C,T,2048]
    win = hann_periodic(N_FFT, dtype)
    frames = frames * win
    total = N_FFT + HOP * (T - 1)
    out = np.zeros((C, total), dtype)
    env = np.zeros(total, dtype)
    w2 = win * win
    for t in range(T):
        out[:, t * HOP:t * HOP + N_FFT] += frames[:, t]
        env[t * HOP:t * HOP + N_FFT] += w2
    pad = N_FFT // 2
    out = out[:, pad:total - pad]
    env = env[pad:total - pad]
    return out / env


def separate(sd, wave, v=None, dtype=np.float32):
    """The infer.py:29-37 sandwich: waveform [C,n] -> waveform [C,(T-1)*1024]."""
    return istft_interleaved(forward(sd, stft_interleaved(wave, dtype), v, dtype), dtype)


def infer_outputs(sd, wave, v=None):
    """The files infer.py writes for a [C, n] input (mono already duplicated, infer.py:26-27): the separated signal
    (:29-42), the printed "Separation dB" (:44-47, natural log) and the five re-mixes mix_{100,90,50,20,-100}.wav
    (:49-79): dialog + g * inst with inst = waveform - dialog, each scaled by orig_peak / mix.max() where orig_peak is
    the input's largest (signed) sample (:28) and mix.max() the largest signed sample of the re-mix."""
    wave = np.asarray(wave, np.float32)
    orig_peak = wave.max()                                                   # :28
    x = separate(sd, wave, v)                                                # :29-37
    w = wave[:, :x.shape[1]]                                                 # :41
    db = 10 * np.log(np.sum(np.square(w)) / np.sum(np.square(w - x)))        # :44-47
    inst = w - x                                                             # :50
    mixes = {}
    for tag, g in (("100", 0.0), ("90", 0.3), ("50", 0.5), ("20", 0.8)):     # :51-73
        mix = x + np.float32(g) * inst if g else x
        mixes[tag] = mix * orig_peak / mix.max()
    mixes["-100"] = inst * orig_peak / inst.max()                            # :75-79
    return x, db, mixes


# --------------------------------------------------------------------------- streaming (infer-streaming.py)
class StreamingOracle:
    """infer-streaming.py:84-147.  Sliding 2048 buffer (zeros initially), one model step per
    1024-sample chunk, irfft, 2-slot overlap-add with NO synthesis window, / sum of window."""

    def __init__(self, sd, C=2, v=None, dtype=np.float32):
        self.sd, self.dtype = sd, dtype
        self.v = generate_bandsplits()[0] if v is None else v
        self.C = C
        self.win = hann_periodic(N_FFT, dtype)
        self.buf = np.zeros((C, N_FFT), dtype)                           # :85
        self.state = np.zeros((4, 2, C * len(self.v), H), dtype)         # :88
        self.prev = np.zeros((C, N_FFT), dtype)                          # :93 previous_speech

    def step(self, chunk):
        """chunk [C,1024] -> out [C,1024] (delayed by one chunk)."""
        dt = self.dtype
        self.buf = np.concatenate((self.buf[:, HOP:], np.asarray(chunk, dt)), 1)   # :116
        X = np.fft.rfft(self.buf * self.win, axis=1)                               # :119
        x = np.empty((self.C, 2 * X.shape[1]), dt)
        x[:, 0::2], x[:, 1::2] = X.real, X.imag                                    # :120-121
        y, self.state = forward_recurrent(self.sd, x, self.state, self.v, dt)      # :123
        Y = y[:, 0::2] + 1j * y[:, 1::2]
        wf = np.fft.irfft(Y, n=N_FFT, axis=1).astype(dt)                           # :127
        sow = self.win[:HOP] + self.win[HOP:]                                      # :138-143
        out = (wf[:, :HOP] + self.prev[:, HOP:]) / sow                             # :145
        self.prev = wf
        return out


# --------------------------------------------------------------------------- LADSPA chunker (speech-ladspa-onnx.cpp)
class LadspaOracle:
    """speech-ladspa-onnx.cpp:152-267: arbitrary-size run() re-blocked to 1024-sample chunks;
    channel 0 feeds both model rows; double-precision FFT; wet/dry `mix` control; mono result
    to both outputs; 1024-sample output delay; state carried."""

    def __init__(self, sd, v=None):
        self.sd = sd
        self.v = generate_bandsplits()[0] if v is None else v
        k = np.arange(N_FFT, dtype=np.float64)
        self.window = (0.5 * (1 - np.cos(2 * np.pi * k / N_FFT))).astype(np.float32)   # :57-59
        self.current = np.zeros((2, N_FFT), np.float32)
        self.overlap = np.zeros((2, N_FFT), np.float32)
        self.state = np.zeros((4, 2, 2 * len(self.v), H), np.float32)
        self.pos = 0
        self.buf_in = np.zeros((2, HOP), np.float32)
        self.buf_out = np.zeros((2, HOP), np.float32)

    def run(self, in1, in2, mix):
        n = len(in1)
        out1 = np.empty(n, np.float32)
        out2 = np.empty(n, np.float32)
        p = 0
        while p < n:                                                            # :154-168
            k = min(HOP - self.pos, n - p)
            self.buf_in[0, self.pos:self.pos + k] = in1[p:p + k]
            self.buf_in[1, self.pos:self.pos + k] = in2[p:p + k]
            out1[p:p + k] = self.buf_out[0, self.pos:self.pos + k]
            out2[p:p + k] = self.buf_out[1, self.pos:self.pos + k]
            self.pos += k
            p += k
            if self.pos == HOP:
                self._new_chunk(np.float32(mix))
        return out1, out2

    def _new_chunk(self, mix):
        self.current[:, :N_FFT - HOP] = self.current[:, HOP:]                   # :178-181
        self.current[:, N_FFT - HOP:] = self.buf_in[0]                          # :183-188 (channel 0 to both)
        t = (self.current[0] * self.window).astype(np.float64)                  # :192 float product stored to double
        F = np.fft.fft(t)                                                       # :196
        x = np.empty((2, 2050), np.float32)
        x[:, 0::2] = F[:1025].real.astype(np.float32)                           # :199-205
        x[:, 1::2] = F[:1025].imag.astype(np.float32)
        y, self.state = forward_recurrent(self.sd, x, self.state, self.v)       # :208-211, :264
        yr = y[0, 0::2].astype(np.float64)
        yi = y[0, 1::2].astype(np.float64)
        m = float(mix)
        if mix >= 0:                                                            # :216-226
            re = m * yr + (1.0 - m) * F[:1025].real
            im = m * yi + (1.0 - m) * F[:1025].imag
        else:
            re = m * yr + F[:1025].real
            im = m * yi + F[:1025].imag
        G = np.empty(N_FFT, np.complex128)
        G[:1025] = re + 1j * im
        G[1025:] = np.conj(G[1023:0:-1])                                        # :229-233
        s = np.fft.ifft(G).real                                                 # :236, :245 (/fft_size)
        self.overlap[0] = self.overlap[1]                                       # :239-241
        self.overlap[1] = s.astype(np.float32)                                  # :244-246
        tot = self.overlap[0, HOP:] + self.overlap[1, :HOP]                     # :249-256  j=0 -> pos=i+1024, j=1 -> pos=i
        wsum = self.window[HOP:] + self.window[:HOP]
        res = tot / wsum
        self.buf_out[0] = res                                                   # :258-260
        self.buf_out[1] = res
        self.pos = 0
