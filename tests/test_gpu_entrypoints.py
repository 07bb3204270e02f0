"""GPU tests of the drop-in entry points: the LADSPA plugin driven through its C ABI exactly as
a LADSPA host would (ctypes mirror of the public LADSPA_Descriptor), and the two command-line
scripts.  Checked against the oracle's restatement of speech-ladspa-onnx.cpp:152-267 and of
infer.py / infer-streaming.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
PLUGIN = os.path.join(REPO, "speechseparation_amd", "lib", "speech_separator_ladspa.so")


class PortRangeHint(C.Structure):
    _fields_ = [("HintDescriptor", C.c_int), ("LowerBound", C.c_float), ("UpperBound", C.c_float)]


class Descriptor(C.Structure):
    pass


INSTANTIATE = C.CFUNCTYPE(C.c_void_p, C.POINTER(Descriptor), C.c_ulong)
CONNECT = C.CFUNCTYPE(None, C.c_void_p, C.c_ulong, C.POINTER(C.c_float))
VOIDH = C.CFUNCTYPE(None, C.c_void_p)
RUN = C.CFUNCTYPE(None, C.c_void_p, C.c_ulong)
GAIN = C.CFUNCTYPE(None, C.c_void_p, C.c_float)
Descriptor._fields_ = [
    ("UniqueID", C.c_ulong), ("Label", C.c_char_p), ("Properties", C.c_int), ("Name", C.c_char_p), ("Maker", C.c_char_p),
    ("Copyright", C.c_char_p), ("PortCount", C.c_ulong), ("PortDescriptors", C.POINTER(C.c_int)),
    ("PortNames", C.POINTER(C.c_char_p)), ("PortRangeHints", C.POINTER(PortRangeHint)), ("ImplementationData", C.c_void_p),
    ("instantiate", INSTANTIATE), ("connect_port", CONNECT), ("activate", VOIDH), ("run", RUN), ("run_adding", RUN),
    ("set_run_adding_gain", GAIN), ("deactivate", VOIDH), ("cleanup", VOIDH)]


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.fixture(scope="module")
def weight_file(tmp_path_factory, sd_default):
    from speechseparation_amd import weights
    p = str(tmp_path_factory.mktemp("w") / "model-always.bsrnnw")
    weights.save_flat(p, sd_default)
    return p


def test_ladspa_descriptor_matches_reference():
    lib = C.CDLL(PLUGIN)
    lib.ladspa_descriptor.restype = C.POINTER(Descriptor)
    lib.ladspa_descriptor.argtypes = [C.c_ulong]
    assert not lib.ladspa_descriptor(1)
    d = lib.ladspa_descriptor(0).contents
    # speech-ladspa-onnx.cpp:293-337
    assert d.UniqueID == 0xF433B044 and d.Label == b"speech_separator" and d.Name == b"Speech Separator"
    assert d.Maker == b"Pierre-Hugues Husson @ Freebox" and d.Copyright == b"None" and d.Properties == 0 and d.PortCount == 5
    assert [d.PortDescriptors[i] for i in range(5)] == [0x1 | 0x4, 0x1 | 0x8, 0x1 | 0x8, 0x2 | 0x8, 0x2 | 0x8]
    assert [d.PortNames[i] for i in range(5)] == [b"Control", b"Input (Left)", b"Input (Right)", b"Output (Left)", b"Output (Right)"]
    assert d.PortRangeHints[0].HintDescriptor == (0x240 | 0x1) and d.PortRangeHints[0].LowerBound == 0.0
    assert not d.activate and not d.run_adding and not d.deactivate and d.run and d.cleanup


def test_ladspa_instantiate_fails_cleanly_without_weights(tmp_path):
    env = dict(os.environ, BSRNN_WEIGHTS=str(tmp_path / "missing.bsrnnw"))
    code = ("import ctypes as C; l=C.CDLL(%r); l.ladspa_descriptor.restype=C.c_void_p; "
            "import sys; sys.path.insert(0,%r); sys.path.insert(0,%r+'/tests'); from test_gpu_entrypoints import Descriptor; "
            "d=C.cast(l.ladspa_descriptor(0), C.POINTER(Descriptor)).contents; h=d.instantiate(None, 44100); "
            "print('HANDLE', h)") % (PLUGIN, REPO, REPO)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert "HANDLE None" in out.stdout, out.stdout + out.stderr     # NULL, no crash, nothing thrown across the ABI


@pytest.mark.parametrize("mix", [1.0, 0.35, -0.5])
def test_ladspa_run_matches_oracle(weight_file, sd_default, mix):
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import weights
    os.environ["BSRNN_WEIGHTS"] = weight_file
    lib = C.CDLL(PLUGIN)
    lib.ladspa_descriptor.restype = C.POINTER(Descriptor)
    lib.ladspa_descriptor.argtypes = [C.c_ulong]
    d = lib.ladspa_descriptor(0).contents
    h = d.instantiate(None, 44100)
    assert h
    wave = weights.synth_waveform(2, 6 * 1024, seed=77)
    control = np.array([mix], np.float32)
    oracle = onp.LadspaOracle(sd_default)
    p = 0
    outs, refs = [], []
    for n in (333, 1024, 1, 2048, 690, 1024, 1024):          # arbitrary host block sizes, 6144 samples
        i1 = np.ascontiguousarray(wave[0, p:p + n]); i2 = np.ascontiguousarray(wave[1, p:p + n])
        o1 = np.empty(n, np.float32); o2 = np.empty(n, np.float32)
        d.connect_port(h, 0, fptr(control)); d.connect_port(h, 1, fptr(i1)); d.connect_port(h, 2, fptr(i2))
        d.connect_port(h, 3, fptr(o1)); d.connect_port(h, 4, fptr(o2))
        d.run(h, n)
        r1, r2 = oracle.run(i1, i2, mix)
        assert np.array_equal(o1, o2)                        # mono result on both outputs
        outs.append(o1); refs.append(r1)
        p += n
    d.cleanup(h)
    got, ref = np.concatenate(outs), np.concatenate(refs)
    assert np.all(got[:1024] == 0)                           # one-chunk output delay
    err = float(np.abs(got - ref).max())
    print("ladspa mix=%g err %.3e" % (mix, err))
    assert err < 1e-4


def test_ladspa_run_does_no_first_use_work(weight_file):
    """The reference builds its session, FFT plans and state in the plugin's constructor and run() only uses them
    (speech-ladspa-onnx.cpp:55-120, :152-169).  Here instantiate() must have loaded every kernel, captured and instantiated both
    parity graphs of the streaming step and sized every buffer: the first chunks run() processes allocate, capture and instantiate
    nothing (process-wide counters of the library) and take no longer than twice a steady-state chunk."""
    import time
    from speechseparation_amd import _native, weights
    os.environ["BSRNN_WEIGHTS"] = weight_file
    lib = C.CDLL(PLUGIN)
    lib.ladspa_descriptor.restype = C.POINTER(Descriptor)
    lib.ladspa_descriptor.argtypes = [C.c_ulong]
    d = lib.ladspa_descriptor(0).contents
    counters = lambda: tuple(_native.lib.bsrnn_debug_counter(i) for i in range(3))    # allocations, captures, instantiations
    before = counters()
    h = d.instantiate(None, 44100)
    assert h
    ready = counters()
    graph = os.environ.get("BSRNN_STREAM_GRAPH") is not None
    if graph:
        assert ready[1] - before[1] == 2 and ready[2] - before[2] == 2   # both parity graphs were made in instantiate()
    launches0 = _native.lib.bsrnn_debug_counter(3)
    wave = weights.synth_waveform(2, 64 * 1024, seed=9)
    control = np.array([1.0], np.float32)
    o1 = np.empty(1024, np.float32); o2 = np.empty(1024, np.float32)
    times = []
    for k in range(64):                                                    # one model step per run() call
        i1 = np.ascontiguousarray(wave[0, 1024 * k:1024 * (k + 1)]); i2 = np.ascontiguousarray(wave[1, 1024 * k:1024 * (k + 1)])
        d.connect_port(h, 0, fptr(control)); d.connect_port(h, 1, fptr(i1)); d.connect_port(h, 2, fptr(i2))
        d.connect_port(h, 3, fptr(o1)); d.connect_port(h, 4, fptr(o2))
        t0 = time.perf_counter()
        d.run(h, 1024)
        times.append(time.perf_counter() - t0)
        if k == 1:
            assert counters() == ready, "run() allocated, captured or instantiated on its first chunks"
    assert counters() == ready
    assert _native.lib.bsrnn_debug_counter(3) - launches0 == (64 if graph else 0)      # (BSRNN_STREAM_GRAPH=1: every chunk a graph replay)
    d.cleanup(h)
    steady = float(np.median(times[8:]))
    print("ladspa run(): first chunks %.0f / %.0f us, steady state %.0f us" % (1e6 * times[0], 1e6 * times[1], 1e6 * steady))
    assert times[0] <= 2.0 * steady + 50e-6 and times[1] <= 2.0 * steady + 50e-6, (times[:4], steady)


def test_ladspa_run_with_the_step_as_a_graph(weight_file):
    """BSRNN_STREAM_GRAPH=1 (A/B: the model part of a step as one hipGraph per carry parity, captured and instantiated in instantiate()):
    the same no-first-use-work guarantees, in a child process (the plugin reads the switch when it creates its stream)."""
    env = dict(os.environ, BSRNN_STREAM_GRAPH="1", BSRNN_WEIGHTS=weight_file, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-s", "-k", "test_ladspa_run_does_no_first_use_work",
                        "-p", "no:cacheprovider"], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-1200:])
    assert r.returncode == 0, r.stdout[-3000:]


def test_ladspa_in_place_buffers(weight_file):
    """Hosts may run in place (output buffer == input buffer); the reference reads before it writes."""
    from speechseparation_amd import weights
    os.environ["BSRNN_WEIGHTS"] = weight_file
    lib = C.CDLL(PLUGIN)
    lib.ladspa_descriptor.restype = C.POINTER(Descriptor)
    lib.ladspa_descriptor.argtypes = [C.c_ulong]
    d = lib.ladspa_descriptor(0).contents
    wave = weights.synth_waveform(2, 3 * 1024, seed=5)
    control = np.array([1.0], np.float32)
    res = []
    for inplace in (False, True):
        h = d.instantiate(None, 48000)
        a = wave[0].copy(); b = wave[1].copy()
        o1 = a if inplace else np.empty_like(a); o2 = b if inplace else np.empty_like(b)
        d.connect_port(h, 0, fptr(control)); d.connect_port(h, 1, fptr(a)); d.connect_port(h, 2, fptr(b))
        d.connect_port(h, 3, fptr(o1)); d.connect_port(h, 4, fptr(o2))
        d.run(h, a.size)
        res.append(o1.copy())
        d.cleanup(h)
    assert np.array_equal(res[0], res[1])


def _write_wav(path, wave, sr):
    from speechseparation_amd import audio
    audio.save_wav(path, torch.from_numpy(wave), sr)


def test_infer_cli(tmp_path, sd_default):
    """infer.py: same flags and output files as the reference; values against the oracle sandwich."""
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import audio, weights
    wave = weights.synth_waveform(1, 16000 * 2 + 123, seed=9)            # mono 2 s @ 16 kHz, ragged tail
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    _write_wav(src, wave, 16000)
    out = subprocess.run([sys.executable, os.path.join(REPO, "infer.py"), "--input", src, "--output", dst,
                          "--synthetic-weights", "0", "--outdir", str(tmp_path)], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr
    assert "Separation dB" in out.stdout
    got, sr = audio.load_wav(dst)
    assert sr == 16000 and got.shape[0] == 2
    # mono duplicated to two rows (infer.py:26-27); every output file against the oracle restatement of infer.py:28-79
    ref, expect_db, mixes = onp.infer_outputs(sd_default, np.concatenate((wave, wave), 0))
    assert got.shape[1] == ref.shape[1]
    assert float(np.abs(got.numpy() - ref).max()) < 1e-4
    for tag in ("100", "90", "50", "20", "-100"):
        m, sr_m = audio.load_wav(str(tmp_path / ("mix_%s.wav" % tag)))
        assert sr_m == 16000 and m.shape == got.shape
        # the re-mix is normalised to the input's peak: the values are O(input), the separated signal's 1e-4 carries over
        e = float(np.abs(m.numpy() - mixes[tag]).max())
        print("mix_%s.wav: max |file - oracle| %.2e (peak %.3f)" % (tag, e, float(np.abs(mixes[tag]).max())))
        assert e < 1e-4 * max(1.0, float(np.abs(mixes[tag]).max()) / float(np.abs(ref).max())), (tag, e)
        assert abs(float(m.max()) - float(wave.max())) < 1e-5          # peak normalisation: mix.max() == orig_peak
    # the printed figure uses the natural log, as the reference does (infer.py:47)
    printed = float(out.stdout.split("Separation dB")[1].split()[0])
    assert abs(printed - expect_db) < 1e-2


def test_infer_streaming_cli(tmp_path, sd_default):
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import audio, weights
    wave = weights.synth_waveform(2, 44100 // 4 + 50, seed=10)           # already 44.1 kHz: no resampling in the way
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    _write_wav(src, wave, 44100)
    out = subprocess.run([sys.executable, os.path.join(REPO, "infer-streaming.py"), "--input", src, "--output", dst, "--name", "t",
                          "--synthetic-weights", "0", "--export", str(tmp_path / "hello.bsrnnw")],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr
    assert "Elapsed" in out.stdout and os.path.exists(str(tmp_path / "hello.bsrnnw"))
    got, sr = audio.load_wav(dst)
    n_chunks = wave.shape[1] // 1024
    assert sr == 44100 and tuple(got.shape) == (2, n_chunks * 1024)
    so = onp.StreamingOracle(sd_default, C=2)
    ref = np.concatenate([so.step(wave[:, i * 1024:(i + 1) * 1024]) for i in range(n_chunks)], 1)
    assert float(np.abs(got.numpy() - ref).max()) < 1e-4
    v, sd2 = weights.load_flat(str(tmp_path / "hello.bsrnnw"))
    assert all(np.array_equal(sd2[k], sd_default[k]) for k in sd_default)


def test_infer_streaming_cli_resamples_to_44k1(tmp_path, sd_default):
    """infer-streaming.py reads its input at 44.1 kHz whatever the file's rate (torchaudio StreamReader with
    sample_rate=44100, infer-streaming.py:77-78).  Here: a 48 kHz file -> audio.resample (scipy resample_poly, 147 / 160)
    -> the streaming loop; expected = the streaming oracle on the same resampled signal.  Parity of resample_poly with the
    reference's ffmpeg resampler is NOT pinned (no ffmpeg / torchaudio in this image): this test pins the plumbing, the
    rate, the length and the model, not the resampling filter."""
    from scipy.signal import resample_poly
    from oracle import bsrnn_numpy as onp
    from speechseparation_amd import audio, weights
    wave = weights.synth_waveform(2, 48000 // 8 + 33, seed=12)           # 0.125 s stereo @ 48 kHz
    src, dst = str(tmp_path / "in48.wav"), str(tmp_path / "out.wav")
    _write_wav(src, wave, 48000)
    out = subprocess.run([sys.executable, os.path.join(REPO, "infer-streaming.py"), "--input", src, "--output", dst, "--name", "t",
                          "--synthetic-weights", "0", "--export", ""], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr
    got, sr = audio.load_wav(dst)
    res = resample_poly(wave, 147, 160, axis=1).astype(np.float32)
    n_chunks = res.shape[1] // 1024
    assert sr == 44100 and n_chunks >= 5 and tuple(got.shape) == (2, n_chunks * 1024)
    so = onp.StreamingOracle(sd_default, C=2)
    ref = np.concatenate([so.step(res[:, i * 1024:(i + 1) * 1024]) for i in range(n_chunks)], 1)
    assert float(np.abs(got.numpy() - ref).max()) < 1e-4


def test_validate_cli(tmp_path, sd_default):
    """validate.py: the reference's validation block (train.py:132-150) over file pairs; averages against the oracle."""
    from oracle import metrics_torch as mt
    from oracle.bsrnn_torch_cpu import TorchCpuBSRNN
    from speechseparation_amd import spec, weights
    oracle = TorchCpuBSRNN(sd_default, spec.generate_bandsplits()[0])
    files, refs = [], []
    for i, n in enumerate((16000 + 7, 3 * 4096)):
        mix = weights.synth_waveform(2, n, seed=60 + i)
        speech = weights.synth_waveform(2, n, seed=70 + i, scale=0.05)
        pm, ps = str(tmp_path / ("mix%d.wav" % i)), str(tmp_path / ("speech%d.wav" % i))
        _write_wav(pm, mix, 16000)
        _write_wav(ps, speech, 16000)
        files += [pm, ps]
        refs.append(mt.train_infer(oracle.forward, torch.from_numpy(mix), torch.from_numpy(speech)))
    out = subprocess.run([sys.executable, os.path.join(REPO, "validate.py"), "--pairs"] + files + ["--synthetic-weights", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr
    words = out.stdout.split()
    loss = float(words[words.index("Loss") + 1])
    sdr = float(words[words.index("SDR") + 1])
    sisdr = float(words[words.index("SI-SDR") + 1])
    assert abs(loss - np.mean([r["loss"] for r in refs])) < 1e-4
    assert abs(sdr - np.mean([r["sdr"] for r in refs])) < 2e-3
    assert abs(sisdr - np.mean([r["sisdr"] for r in refs])) < 2e-3
