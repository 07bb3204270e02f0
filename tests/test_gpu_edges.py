"""GPU edge cases against the oracle: minimum-length input (T = 2), a single row, odd row counts that
do not fill any tile, the 41-band table through the recurrent / streaming entry points, and a long
sequence checked through causality (chunked == offline)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def make_model(sd, v=None):
    from speechseparation_amd.bsrnn import BSRNN
    m = BSRNN(v).eval()
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()}, strict=True)
    return m.to("cuda")


def test_minimum_length_and_single_row(sd_default):
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    m = make_model(sd_default)
    for rows, n in ((1, 1025), (1, 2047), (3, 2048), (5, 3000)):
        wave = weights.synth_waveform(rows, n, seed=100 + n)
        out = m.separate(torch.from_numpy(wave).cuda())
        ref = onp.separate(sd_default, wave)
        assert tuple(out.shape) == ref.shape == (rows, (n // 1024) * 1024)
        assert maxabs(out.cpu().numpy(), ref) < TOL, (rows, n)


def test_odd_shapes_forward_and_mask(sd_default):
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    m = make_model(sd_default)
    x = onp.stft_interleaved(weights.synth_waveform(5, 2 * 1024 + 17, seed=77))       # [5, 2050, 3]
    taps = {}
    ref = onp.forward(sd_default, x, taps=taps)
    y, mask = m.forward_with_mask(torch.from_numpy(x).cuda())
    assert maxabs(y.cpu().numpy(), ref) < TOL and maxabs(mask.cpu().numpy(), taps["mask"]) < TOL
    z = weights.synth_tensor((3, 2, 12, 64), seed=5, scale=0.4)
    st = weights.synth_tensor((4, 2, 36, 64), seed=6, scale=0.3)
    zr, sr = onp.dual_path(sd_default, z, st)
    zo, so = m.dual_path(torch.from_numpy(z).cuda(), torch.from_numpy(st).cuda())
    assert maxabs(zo.cpu().numpy(), zr) < 2e-5 and maxabs(so.cpu().numpy(), sr) < 2e-5


def test_bands41_recurrent_and_streaming():
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import spec, weights
    from speechseparation_amd.bsrnn import StreamingSeparator
    v = spec.variant_bandsplits("41")
    sd = weights.synth_state_dict(v, seed=3)
    m = make_model(sd, v)
    K = len(v)
    x = onp.stft_interleaved(weights.synth_waveform(2, 3 * 1024, seed=9))              # [2, 2050, 4]
    state = weights.synth_tensor((4, 2, 2 * K, 64), seed=11, scale=0.2)
    yr, sr = onp.forward_recurrent(sd, x[:, :, 1], state, v)
    y, s = m.forward_recurrent(torch.from_numpy(np.ascontiguousarray(x[:, :, 1])).cuda(), torch.from_numpy(state).cuda())
    assert maxabs(y.cpu().numpy(), yr) < TOL and maxabs(s.cpu().numpy(), sr) < 2e-5
    st = StreamingSeparator(m, channels=2)
    so = onp.StreamingOracle(sd, C=2, v=v)
    wave = weights.synth_waveform(2, 3 * 1024, seed=12)
    for i in range(3):
        c = wave[:, i * 1024:(i + 1) * 1024]
        assert maxabs(st.step(torch.from_numpy(c.copy()).cuda()).cpu().numpy(), so.step(c)) < TOL


def test_long_sequence_causality(sd_default):
    """T = 400 frames (about 9.5 s at 44.1 kHz): arbitrary chunkings with state carry equal the
    offline forward; the state after the whole sequence is independent of the chunking."""
    from speechseparation_amd import weights
    m = make_model(sd_default)
    x = m.stft(torch.from_numpy(weights.synth_waveform(2, 399 * 1024 + 5, seed=21)).cuda())   # [2, 2050, 400]
    y_off = m(x)
    finals = []
    for cuts in ((0, 400), (0, 1, 257, 400), (0, 100, 200, 300, 400)):
        state = torch.zeros((4, 2, 24, 64), device="cuda")
        ys = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            y, state = m.forward_chunk(x[:, :, a:b].contiguous(), state)
            ys.append(y)
        assert maxabs(torch.cat(ys, 2).cpu().numpy(), y_off.cpu().numpy()) < 3e-5
        finals.append(state.cpu().numpy())
    assert maxabs(finals[0], finals[1]) < 2e-5 and maxabs(finals[0], finals[2]) < 2e-5


@pytest.mark.parametrize("scale", [1e-4, 1e-6])
def test_small_inputs_stay_accurate(sd_default, scale):
    """The reference's fp32 forward has no magnitude floor (bsrnn.py:385-443): quiet audio keeps its relative accuracy.  fp16 pieces go
    subnormal below 6.1e-5 and vanish below 6e-8, so the split-precision kernels are held, on inputs scaled by 1e-4 and 1e-6 (spectrum
    values down to ~1e-7 and below), to the float32 oracle's own distance from the float64 evaluation - relative to the output range -
    for forward, the dual path on its own and waveform -> waveform."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    m = make_model(sd_default)
    x = weights.synth_tensor((2, 2050, 8), seed=21, scale=scale)
    y64 = onp.forward(sd_default, x, dtype=np.float64)
    y32 = onp.forward(sd_default, x, dtype=np.float32)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    rng = np.abs(y64).max()
    e_hip, e_f32 = maxabs(y, y64) / rng, maxabs(y32, y64) / rng
    print("forward, inputs x %g: max|y| %.3g  |hip - f64| / range %.2e   |f32 oracle - f64| / range %.2e" % (scale, rng, e_hip, e_f32))
    assert e_hip <= 3 * e_f32 + 2e-7
    z = weights.synth_tensor((2, 8, 12, 64), seed=22, scale=scale)
    z64, _ = onp.dual_path(sd_default, z.astype(np.float64), None, np.float64)
    z32, _ = onp.dual_path(sd_default, z, None, np.float32)
    zo, _ = m.dual_path(torch.from_numpy(z).cuda())
    rng = np.abs(z64).max()
    e_hip, e_f32 = maxabs(zo.cpu().numpy(), z64) / rng, maxabs(z32, z64) / rng
    print("dual path, inputs x %g: max|z| %.3g  |hip - f64| / range %.2e   |f32 oracle - f64| / range %.2e" % (scale, rng, e_hip, e_f32))
    assert e_hip <= 3 * e_f32 + 2e-7
    wave = weights.synth_waveform(2, 5 * 1024 + 3, seed=23) * scale
    w64 = onp.separate(sd_default, wave.astype(np.float64), dtype=np.float64)
    w32 = onp.separate(sd_default, wave, dtype=np.float32)
    wo = m.separate(torch.from_numpy(wave.astype(np.float32)).cuda()).cpu().numpy()
    rng = np.abs(w64).max()
    e_hip, e_f32 = maxabs(wo, w64) / rng, maxabs(w32, w64) / rng
    print("separate, inputs x %g: max|out| %.3g  |hip - f64| / range %.2e   |f32 oracle - f64| / range %.2e" % (scale, rng, e_hip, e_f32))
    assert e_hip <= 3 * e_f32 + 2e-7


def test_small_weight_rows_stay_accurate(sd_default):
    """... and neither do the weights: a band whose Linear layers have rows at the 1e-5 scale (a band the training all but switched off)
    and an LSTM with 1e-5-scale input weights must come out as accurately as the float32 oracle does, relative to the output range."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    sd = {k: v.copy() for k, v in sd_default.items()}
    for k in sd:
        if k.startswith(("bandFCs_pre.7.", "bandFCs.7.", "bandFCs_back.9.", "bandFCs_back_post.9.")) and k.endswith("weight"):
            sd[k][::2] *= 1e-5                                   # every other output row tiny
        if k in ("lstms.1.m.rnn.weight_ih_l0", "lstms.2.m.rnn.weight_ih_l1"):
            sd[k] *= 1e-5
    m = make_model(sd)
    x = weights.synth_tensor((2, 2050, 8), seed=24, scale=1.0)
    y64 = onp.forward(sd, x, dtype=np.float64)
    y32 = onp.forward(sd, x, dtype=np.float32)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    rng = np.abs(y64).max()
    e_hip, e_f32 = maxabs(y, y64) / rng, maxabs(y32, y64) / rng
    print("tiny weight rows: max|y| %.3g  |hip - f64| / range %.2e   |f32 oracle - f64| / range %.2e" % (rng, e_hip, e_f32))
    assert e_hip <= 3 * e_f32 + 2e-7
    # per band: the bands fed by the tiny rows against their own range (the global range would hide them)
    off = np.cumsum([0] + [2 * w for w in m.band_widths])
    for b in (7, 9):
        sl = slice(off[b], off[b + 1])
        r_b = np.abs(y64[:, sl]).max()
        print("  band %d: range %.3g  |hip - f64| / band range %.2e  |f32 - f64| / band range %.2e" % (
            b, r_b, maxabs(y[:, sl], y64[:, sl]) / r_b, maxabs(y32[:, sl], y64[:, sl]) / r_b))
        assert maxabs(y[:, sl], y64[:, sl]) / r_b <= 3 * maxabs(y32[:, sl], y64[:, sl]) / r_b + 1e-6


def test_large_inputs_stay_accurate_and_out_of_range_is_exact(sd_default):
    """The fp16x2 matrix path represents operands up to 65504.  Inside that range accuracy must not depend on the magnitude
    (a spectrum scaled to |x| ~ 1e3, the ceiling for audio in [-1, 1]).  Beyond it the reference's forward still returns
    the right numbers, so the drop-in must too: under the default range policy `m(big)` ITSELF equals the float64 oracle
    (the call is run again on the exact-fp32 kernels before it returns) and no later call raises; the same through
    forward_recurrent (state carried from the same starting point), separate and the device-side streaming step.  Under
    the 'deferred' policy (benchmark loops) nothing waits and the violation is reported, once, by the next call."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import _native, weights
    from speechseparation_amd._native import NativeError
    from speechseparation_amd.bsrnn import StreamingSeparator
    m = make_model(sd_default)
    x = weights.synth_tensor((2, 2050, 6), seed=5, scale=250.0)          # |x| up to ~1e3
    ref = onp.forward(sd_default, x, dtype=np.float64)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    rel = maxabs(y, ref) / np.abs(ref).max()
    print("scaled input: max|y| %.3g, relative error %.2e" % (np.abs(ref).max(), rel))
    assert rel < 2e-6
    if _native.compute_mode()["gemm"] != "fp16x2":
        return
    xb = x * 100.0                                                      # |x| up to 1.2e5: outside the fp16x2 range
    refb = onp.forward(sd_default, xb, dtype=np.float64)
    yb = m(torch.from_numpy(xb).cuda()).cpu().numpy()
    relb = maxabs(yb, refb) / np.abs(refb).max()
    print("out-of-range input: max|y| %.3g, relative error %.2e" % (np.abs(refb).max(), relb))
    assert np.isfinite(yb).all() and relb < 2e-6
    y2 = m(torch.from_numpy(x).cuda()).cpu().numpy()                    # nothing is left pending: the next call neither raises nor differs
    assert np.array_equal(y2, y)
    # one recurrent frame from a non-trivial state
    K = len(m.band_widths)
    state = weights.synth_tensor((4, 2, 2 * K, 64), seed=11, scale=0.2)
    yr, sr = onp.forward_recurrent(sd_default, xb[:, :, 1], state, dtype=np.float64)
    yg, sg = m.forward_recurrent(torch.from_numpy(np.ascontiguousarray(xb[:, :, 1])).cuda(), torch.from_numpy(state).cuda())
    # (gates this far into saturation amplify rounding: the float32 numpy oracle itself is 3e-5 from float64 on this state)
    assert maxabs(yg.cpu().numpy(), yr) / np.abs(yr).max() < 2e-6 and maxabs(sg.cpu().numpy(), sr) < 2e-4
    # waveform -> waveform and the device-side streaming step on a waveform far outside [-1, 1]
    wave = weights.synth_waveform(2, 5 * 1024 + 3, seed=31) * 3e4
    refw = onp.separate(sd_default, wave.astype(np.float64), dtype=np.float64)
    outw = m.separate(torch.from_numpy(wave).cuda()).cpu().numpy()
    assert maxabs(outw, refw) / np.abs(refw).max() < 2e-6
    st = StreamingSeparator(m, channels=2)
    so = onp.StreamingOracle(sd_default, C=2)
    for i in range(3):
        c = wave[:, i * 1024:(i + 1) * 1024]
        r = so.step(c)
        g = st.step(torch.from_numpy(c.copy()).cuda()).cpu().numpy()
        assert maxabs(g, r) <= 2e-5 * max(1.0, np.abs(r).max()), i
    del st
    # the opt-out: nothing waits, the next call reports it once, the context keeps working
    m.set_range_policy("deferred")
    try:
        m(torch.from_numpy(xb).cuda())
        torch.cuda.synchronize()
        with pytest.raises(NativeError, match="65504"):
            m(torch.from_numpy(x).cuda())
        y3 = m(torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.array_equal(y3, y)
    finally:
        m.set_range_policy("exact")


@pytest.mark.parametrize("T", [4, 5, 6, 7, 8, 9, 12, 13, 16, 17, 26])
def test_frame_counts_around_the_kernels_chunk_sizes(sd_default, T):
    """Sequence lengths on both sides of every internal chunk size: 6 frames per STFT / iSTFT workgroup, 8-step
    staging chunks and 4-step batched input groups of the time-axis LSTM (layer 1 runs 4 steps behind)."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    m = make_model(sd_default)
    n = (T - 1) * 1024 + 37
    wave = weights.synth_waveform(3, n, seed=900 + T)
    out = m.separate(torch.from_numpy(wave).cuda())
    ref = onp.separate(sd_default, wave)
    assert tuple(out.shape) == ref.shape == (3, (T - 1) * 1024)
    e = maxabs(out.cpu().numpy(), ref)
    print("T=%d separate err %.3e" % (T, e))
    assert e < TOL
    # the chunked / recurrent entry point with a state carry across an odd split of the same frames
    x = onp.stft_interleaved(wave)
    xt = torch.from_numpy(x).cuda()
    y_off = m(xt)
    K = len(m.band_widths)
    state = torch.zeros((4, 2, 3 * K, 64), device="cuda")
    cut = max(1, T // 3)
    y1, state = m.forward_chunk(xt[:, :, :cut].contiguous(), state)
    y2, state = m.forward_chunk(xt[:, :, cut:].contiguous(), state)
    assert maxabs(torch.cat([y1, y2], dim=2).cpu().numpy(), y_off.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("rows", [64, 130, 200])
def test_large_batches_equal_their_row_blocks(sd_default, rows):
    """Rows are independent (bsrnn.py:394-395).  Once a batch's time-axis launch no longer fits one round of workgroups (from 171 rows on at
    12 bands; from 128 on until round 4) bsrnn_separate runs two row blocks concurrently on two streams (the second a stage behind): a
    200-row call - and a 130-row one (one block, time-axis launches with eight sequences per workgroup) and a 64-row one, the benchmark's
    shape - must equal the same rows separated block by block, bit for bit, run to run, and the oracle on a few rows.  (This check exposed
    the co-residency hazard of the vectorised FFT kernels, csrc/fft.hip.)"""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    m = make_model(sd_default)
    wave = weights.synth_waveform(rows, 16 * 1024 + 9, seed=31)
    w = torch.from_numpy(wave).cuda()
    whole = m.separate(w).cpu().numpy()
    h = rows // 2
    halves = np.concatenate([m.separate(w[:h].contiguous()).cpu().numpy(), m.separate(w[h:].contiguous()).cpu().numpy()], 0)
    assert np.array_equal(whole, halves)
    for _ in range(5):
        assert np.array_equal(whole, m.separate(w).cpu().numpy())
    pick = [0, h - 1, h, rows - 1]
    ref = onp.separate(sd_default, wave[pick])
    assert maxabs(whole[pick], ref) < TOL


def test_stream_survives_regrow_and_recommit(sd_default, sd_hot):
    """A streaming step replays a captured hipGraph that holds the context's workspace and weight-arena pointers.  A larger
    call on the same model (workspace regrown and freed) and a load_state_dict (arena rebuilt) between two steps must
    lead to a re-capture, not to a replay into freed memory: every step equals the oracle's."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    from speechseparation_amd.bsrnn import StreamingSeparator
    m = make_model(sd_default)
    st = StreamingSeparator(m, channels=2)
    so = onp.StreamingOracle(sd_default, C=2)
    chunks = weights.synth_waveform(2, 6 * 1024, seed=77)
    def step(i, oracle):
        c = chunks[:, i * 1024:(i + 1) * 1024]
        out = st.step(torch.from_numpy(c.copy()).cuda()).cpu().numpy()
        e = maxabs(out, oracle.step(c))
        assert e < TOL, (i, e)
    step(0, so)
    step(1, so)
    big = weights.synth_waveform(40, 20 * 1024 + 5, seed=78)          # 840 frame rows: far beyond the stream's 2-row workspace
    ref_big = onp.separate(sd_default, big[:2])
    assert maxabs(m.separate(torch.from_numpy(big).cuda()).cpu().numpy()[:2], ref_big) < TOL
    step(2, so)                                                         # workspace was regrown: graph re-captured
    # new weights into the same model: the stream keeps its DSP state and LSTM state, the oracle gets the same switch
    m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd_hot.items()})
    so.sd = sd_hot
    step(3, so)
    bigger = weights.synth_waveform(48, 20 * 1024 + 5, seed=79)
    m.separate(torch.from_numpy(bigger).cuda())
    step(4, so)
    step(5, so)


def test_context_outlives_destroy_while_a_stream_is_alive(sd_default):
    """bsrnn_destroy with a live stream retires the context instead of freeing what the stream points at."""
    import ctypes
    from speechseparation_amd import _native
    lib = _native.lib
    m = make_model(sd_default)
    ctx = m._context(torch.device("cuda", 0))
    h = ctypes.c_void_p()
    _native.check(lib.bsrnn_stream_create(ctx, 2, ctypes.byref(h)))
    lib.bsrnn_destroy(ctx)                                # deferred: the stream still holds the context
    m._ctx = None                                         # the model must not destroy it a second time
    x = torch.zeros((2, 1024), device="cuda")
    rc = lib.bsrnn_stream_step(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_float(1.0), None)
    assert rc == 2 and b"destroyed" in lib.bsrnn_last_error()        # BSRNN_ESTATE, not a crash
    lib.bsrnn_stream_destroy(h)                           # the last stream takes the context with it
    torch.cuda.synchronize()


def test_synchronous_entry_points_rerun_out_of_range_calls_in_fp32(sd_default):
    """|x| ~ 1e7 saturates the fp16x2 operand pieces.  The synchronous entry points (host-buffer streaming step, evaluate)
    must notice before they return and hand back the exact-fp32 result of the same call (rc 0), not saturated numbers one
    call early and an error one call late."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import _native, weights
    from speechseparation_amd.bsrnn import StreamingSeparator
    if _native.compute_mode()["gemm"] != "fp16x2":
        pytest.skip("range guard belongs to the fp16x2 mode")
    m = make_model(sd_default)
    st = StreamingSeparator(m, channels=2)
    so = onp.StreamingOracle(sd_default, C=2, dtype=np.float64)
    chunks = weights.synth_waveform(2, 5 * 1024, seed=91).astype(np.float64)
    scale = [1.0, 1.0, 3e5, 1.0, 1.0]                     # chunk 2 drives the spectrum to |x| ~ 1e7
    for i in range(5):
        c = (chunks[:, i * 1024:(i + 1) * 1024] * scale[i]).astype(np.float32)
        out = st.step(torch.from_numpy(c.copy())).numpy()          # host tensors -> bsrnn_stream_step_host
        ref = so.step(c)
        rel = maxabs(out, ref) / max(np.abs(ref).max(), 1e-30)
        print("step %d: |ref|max %.3g relative error %.2e" % (i, np.abs(ref).max(), rel))
        assert rel < 2e-6, (i, rel)
    # evaluate(): same guarantee
    mix = weights.synth_waveform(2, 6 * 1024, seed=92)
    big = (mix * 3e5).astype(np.float32)
    r = m.evaluate(torch.from_numpy(big).cuda(), torch.from_numpy(big * 0.5).cuda(), return_estimate=True)
    ref = onp.separate(sd_default, big, dtype=np.float64)
    rel = maxabs(r["x_time"].cpu().numpy(), ref) / np.abs(ref).max()
    print("evaluate: relative error of the estimate %.2e" % rel)
    assert rel < 2e-6
    # and nothing is left pending: the next call on the context succeeds
    m.separate(torch.from_numpy(mix).cuda())
    torch.cuda.synchronize()
