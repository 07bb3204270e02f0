"""MI355X-native BSRNN separation inference path (drop-in for phhusson/SpeechSeparation's
bsrnn.py forward / forward_recurrent behind its own entry points).

`BSRNN` and the native binding are imported lazily so that `spec` / `weights` (pure
Python) stay importable on machines without the HIP library; any attempt to *run* the
model without it raises (there is no CPU fallback in the product path).
"""
from . import spec, weights  # noqa: F401
from .spec import generate_bandsplits, BAND_FEATURES as band_features, MERGE_CHANNELS as merge_channels  # noqa: F401


def __getattr__(name):
    if name in ("BSRNN", "StreamingSeparator"):
        from . import bsrnn as _b
        return getattr(_b, name)
    raise AttributeError(name)
