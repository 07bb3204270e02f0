"""Band table and parameter inventory of the BSRNN separation path.

This is *data*, not model code: the HIP library, the Python host class, the
oracle and the synthetic-weight generator all take the band table and the
(key -> shape) inventory from here, so a different band split (e.g. the
41-band / 48 kHz variant of BASELINE.json config 5) is a different table, not
different code.

Reference behaviour restated here:
  * band table            /root/reference/bsrnn.py:247-326 (generate_bandsplits)
  * parameter names/shapes /root/reference/bsrnn.py:329-376 (BSRNN.__init__),
                          :63-76 (NormRNNResidual), :12-16 (TrainableConstantModule)
"""
from collections import OrderedDict

BAND_FEATURES = 64          # bsrnn.py:60
MERGE_CHANNELS = False      # bsrnn.py:61
N_FFT = 2048                # infer.py:31
HOP = 1024                  # infer.py:31
N_BINS = N_FFT // 2 + 1     # 1025, bsrnn.py:299
LEAKY_SLOPE = 0.01          # nn.LeakyReLU() default, used everywhere in bsrnn.py
MASK_HIDDEN = 2 * BAND_FEATURES   # bsrnn.py:360


def generate_bandsplits(n_bins=N_BINS, mul=2):
    """Octave band split; returns (v, w) exactly as bsrnn.py:247-326 does.

    v: band widths in bins: [1, 2] then octaves from pos=3 (each band spans
       pos .. int(pos*mul)), the band that overshoots n_bins is dropped and
       replaced by the remainder, and one zero-width "virtual" band is
       appended.  w: neighbour sums (computed by the reference, never used).
    Integer bookkeeping: must be bit-exact.
    """
    v = [1, 2]
    pos = 3
    while pos < n_bins:
        n = int(pos * mul)
        if n == pos:
            n += 1
        d = n - pos
        v.append(d)
        pos += d
    v.pop()
    if sum(v) != n_bins:
        v.append(n_bins - sum(v))
    v.append(0)
    w = [v[i] + v[i + 1] for i in range(len(v) - 1)]
    w.append(v[0] + v[-1])
    return (v, w)


# Paper-style 41-band table for the 48 kHz variant;
# the reference's zero-width band is appended by `variant_bandsplits`.
BANDS_41 = [4, 5, 4, 4, 4, 5, 4, 4, 4, 5, 10, 11, 11, 10, 11, 11, 10, 11, 11, 10, 11, 11,
            21, 21, 22, 21, 21, 22, 21, 21, 43, 43, 42, 43, 43, 42, 43, 43, 85, 85, 172]


def variant_bandsplits(name):
    if name in (None, "default", "octave12"):
        return generate_bandsplits()[0]
    if name in ("41", "bands41", "48k41"):
        assert sum(BANDS_41) == N_BINS
        return list(BANDS_41) + [0]
    raise ValueError("unknown band table %r" % (name,))


def band_offsets(v):
    """Start bin of every band (prefix sums); interleaved re/im column = 2*bin."""
    off, pos = [], 0
    for x in v:
        off.append(pos)
        pos += x
    return off


def param_spec(v=None):
    """OrderedDict key -> shape, in the reference's state_dict order (288 tensors for
    the default table).  Linear weights are [out, in] (torch layout)."""
    if v is None:
        v = generate_bandsplits()[0]
    H = BAND_FEATURES
    spec = OrderedDict()

    def lin(prefix, n_out, n_in):
        spec[prefix + ".weight"] = (n_out, n_in)
        spec[prefix + ".bias"] = (n_out,)

    for i, x in enumerate(v):                       # bandFCs_pre  bsrnn.py:333-340
        a = 2 * x
        if x > 0:
            lin("bandFCs_pre.%d.0" % i, a, a)
            lin("bandFCs_pre.%d.2" % i, a, a)
        else:
            spec["bandFCs_pre.%d.0.trainable_constant" % i] = (0,)
    for i, x in enumerate(v):                       # bandFCs      bsrnn.py:341-349
        a = 2 * x
        m = max(a, H)
        if x > 0:
            lin("bandFCs.%d.0" % i, m, a)
            lin("bandFCs.%d.2" % i, H, m)
            lin("bandFCs.%d.4" % i, H, H)
        else:
            spec["bandFCs.%d.0.trainable_constant" % i] = (H,)
    for j in range(4):                              # lstms        bsrnn.py:352-356
        bidir = (j % 2 == 0)                        # Band, Time, Band, Time
        p = "lstms.%d.m." % j
        lin(p + "fc_in", H, H)
        for layer in range(2):
            n_in = H if layer == 0 else (2 * H if bidir else H)
            for suffix in (("", "_reverse") if bidir else ("",)):
                spec[p + "rnn.weight_ih_l%d%s" % (layer, suffix)] = (4 * H, n_in)
                spec[p + "rnn.weight_hh_l%d%s" % (layer, suffix)] = (4 * H, H)
                spec[p + "rnn.bias_ih_l%d%s" % (layer, suffix)] = (4 * H,)
                spec[p + "rnn.bias_hh_l%d%s" % (layer, suffix)] = (4 * H,)
        lin(p + "fc", H, 2 * H if bidir else H)
    for i, x in enumerate(v):                       # bandFCs_back bsrnn.py:361-369
        a = 2 * x
        pz = max(a, MASK_HIDDEN)
        if x > 0:
            lin("bandFCs_back.%d.0" % i, MASK_HIDDEN, H)
            lin("bandFCs_back.%d.2" % i, pz, MASK_HIDDEN)
            lin("bandFCs_back.%d.4" % i, a, pz)
        else:
            spec["bandFCs_back.%d.0.trainable_constant" % i] = (0,)
    for i, x in enumerate(v):                       # bandFCs_back_post :370-376
        a = 2 * x
        if x > 0:
            lin("bandFCs_back_post.%d.0" % i, a, a)
            lin("bandFCs_back_post.%d.2" % i, a, a)
        else:
            spec["bandFCs_back_post.%d.0.trainable_constant" % i] = (0,)
    return spec


def n_frames(n_samples, hop=HOP):
    """torch.stft(center=True) frame count: 1 + n // hop (infer.py:31)."""
    return 1 + n_samples // hop
