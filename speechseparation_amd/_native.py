"""ctypes binding of libbsrnn_hip.so (include/bsrnn_hip.h).

There is deliberately no fallback: if the library is missing or cannot be loaded, importing
this module raises, and so does every attempt to run the model.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C speechseparation_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSRNN_HIP_LIB: another build of the same library (A/B measurements of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get("BSRNN_HIP_LIB") or os.path.join(_HERE, "lib", "libbsrnn_hip.so")

# Every symbol include/bsrnn_hip.h declares (tests/test_abi.py checks header == this list == the .so)
SYMBOLS = [
    "bsrnn_abi_version", "bsrnn_last_error", "bsrnn_compute_mode", "bsrnn_create", "bsrnn_destroy", "bsrnn_n_bands", "bsrnn_device",
    "bsrnn_param_count", "bsrnn_param_info", "bsrnn_set_param", "bsrnn_get_param", "bsrnn_commit_params",
    "bsrnn_load_weights_file", "bsrnn_forward", "bsrnn_forward_recurrent", "bsrnn_forward_chunk", "bsrnn_dual_path",
    "bsrnn_stft", "bsrnn_istft", "bsrnn_separate", "bsrnn_stream_create", "bsrnn_stream_destroy", "bsrnn_stream_reset",
    "bsrnn_stream_step", "bsrnn_stream_step_host", "bsrnn_stream_get_state", "bsrnn_set_profiling", "bsrnn_stage_count",
    "bsrnn_stage_name", "bsrnn_stage_times", "bsrnn_dev_alloc", "bsrnn_dev_free", "bsrnn_copy_h2d", "bsrnn_copy_d2h",
    "bsrnn_sync", "bsrnn_evaluate", "bsrnn_io_count", "bsrnn_io_info", "bsrnn_mlp_fused",
    "bsrnn_lstm_train_forward", "bsrnn_lstm_train_backward", "bsrnn_linear_train_forward", "bsrnn_linear_train_backward",
    "bsrnn_istft_backward", "bsrnn_adamw_step", "bsrnn_adamw_step_multi", "bsrnn_adamw_step_multi_dev",
    "bsrnn_linear_group_train_forward", "bsrnn_linear_group_train_backward",
    "bsrnn_set_range_policy", "bsrnn_get_range_policy", "bsrnn_overlap_state", "bsrnn_debug_peek", "bsrnn_debug_counter",
]
RANGE_DEFERRED, RANGE_EXACT = 0, 1          # BSRNN_RANGE_* of include/bsrnn_hip.h
METRIC_NAMES = ("loss", "sdr", "input_sdr", "sisdr", "l1_time", "l1_re", "l1_im", "separation_db")   # BSRNN_M_* order


class NativeError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "HIP library not built: %s is missing (run __graft_entry__.build()). "
            "The BSRNN product path has no CPU fallback." % LIB_PATH)
    # torch (if used by the host) must load its HIP runtime first so both share one runtime
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, fp = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
    sig = {
        "bsrnn_abi_version": (C.c_int, []),
        "bsrnn_last_error": (C.c_char_p, []),
        "bsrnn_compute_mode": (C.c_char_p, []),
        "bsrnn_create": (C.c_int, [C.c_int, C.POINTER(i32), i32, C.POINTER(vp)]),
        "bsrnn_destroy": (None, [vp]),
        "bsrnn_n_bands": (C.c_int, [vp]),
        "bsrnn_device": (C.c_int, [vp]),
        "bsrnn_set_range_policy": (C.c_int, [vp, i32]),
        "bsrnn_get_range_policy": (C.c_int, [vp]),
        "bsrnn_mlp_fused": (C.c_int, [vp]),
        "bsrnn_overlap_state": (C.c_int, [vp]),
        "bsrnn_debug_peek": (C.c_int, [vp, i32, vp, i64]),
        "bsrnn_debug_counter": (C.c_longlong, [i32]),
        "bsrnn_param_count": (C.c_int, [vp]),
        "bsrnn_param_info": (C.c_int, [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]),
        "bsrnn_set_param": (C.c_int, [vp, C.c_char_p, vp, i64]),
        "bsrnn_get_param": (C.c_int, [vp, C.c_char_p, vp, i64]),
        "bsrnn_commit_params": (C.c_int, [vp]),
        "bsrnn_load_weights_file": (C.c_int, [vp, C.c_char_p]),
        "bsrnn_forward": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
        "bsrnn_forward_recurrent": (C.c_int, [vp, vp, vp, vp, vp, i32, vp]),
        "bsrnn_forward_chunk": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp]),
        "bsrnn_dual_path": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp]),
        "bsrnn_lstm_train_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "bsrnn_lstm_train_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "bsrnn_linear_train_forward": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "bsrnn_linear_train_backward": (C.c_int, [vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
        "bsrnn_stft": (C.c_int, [vp, vp, vp, i32, i64, vp]),
        "bsrnn_istft": (C.c_int, [vp, vp, vp, i32, i32, vp]),
        "bsrnn_istft_backward": (C.c_int, [vp, vp, vp, i32, i32, vp]),
        "bsrnn_linear_group_train_forward": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
        "bsrnn_linear_group_train_backward": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
        "bsrnn_adamw_step_multi": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]),
        "bsrnn_adamw_step_multi_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
        "bsrnn_adamw_step": (C.c_int, [vp, vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]),
        "bsrnn_separate": (C.c_int, [vp, vp, vp, i32, i64, vp]),
        "bsrnn_stream_create": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "bsrnn_stream_destroy": (None, [vp]),
        "bsrnn_stream_reset": (C.c_int, [vp, vp]),
        "bsrnn_stream_step": (C.c_int, [vp, vp, vp, C.c_float, vp]),
        "bsrnn_stream_step_host": (C.c_int, [vp, vp, vp, C.c_float]),
        "bsrnn_stream_get_state": (C.c_int, [vp, vp]),
        "bsrnn_set_profiling": (C.c_int, [vp, i32]),
        "bsrnn_stage_count": (C.c_int, []),
        "bsrnn_stage_name": (C.c_char_p, [i32]),
        "bsrnn_stage_times": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i64), i32]),
        "bsrnn_dev_alloc": (C.c_int, [vp, i64, C.POINTER(vp)]),
        "bsrnn_dev_free": (C.c_int, [vp, vp]),
        "bsrnn_copy_h2d": (C.c_int, [vp, vp, vp, i64]),
        "bsrnn_copy_d2h": (C.c_int, [vp, vp, vp, i64]),
        "bsrnn_sync": (C.c_int, [vp, vp]),
        "bsrnn_evaluate": (C.c_int, [vp, vp, vp, i32, i64, vp, C.POINTER(C.c_double), vp]),
        "bsrnn_io_count": (C.c_int, []),
        "bsrnn_io_info": (C.c_int, [vp, i32, i32, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = sig[name]
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise NativeError("libbsrnn_hip: error %d: %s" % (rc, lib.bsrnn_last_error().decode("utf-8", "replace")))


def stage_names():
    return [lib.bsrnn_stage_name(i).decode() for i in range(lib.bsrnn_stage_count())]


def compute_mode():
    """{'gemm': 'f32'|'fp16x2'|'fp16', 'lstm': 'f32'|'fp16x2'} as reported by the library."""
    return dict(kv.split("=") for kv in lib.bsrnn_compute_mode().decode().split())
