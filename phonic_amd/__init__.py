"""phonic_amd — MI355X-native implementation of the per-block DSP hot path of emuell/phonic.

The product is the HIP library `phonic_amd/csrc/libphonic_gpu.so` behind the C ABI declared in
`include/phonic_gpu.h`; this package is the Python-side mirror of that ABI used by the tests, the
smoke test and bench.py. There is no CPU fallback: importing the handles without the built
library raises.
"""
from . import _capi  # noqa: F401
from ._capi import (  # noqa: F401
    FX_CHORUS,
    FX_COMPRESSOR,
    FX_DELAY,
    FX_DISTORTION,
    FX_EQ5,
    FX_FILTER,
    FX_GAIN,
    FX_GATE,
    FX_PANNING,
    FX_REVERB,
    fourcc,
)
from ._wrap import PhonicError  # noqa: F401


def Effect(kind, params=None, reverb_seeds=None, device=0, lfo_seed=None):
    """A standalone effect instance on the GPU (reference `impl Effect`)."""
    from ._wrap import EffectHandle

    return EffectHandle(_capi.load(), "pg_", kind, params, reverb_seeds, device, lfo_seed)


def Graph(sample_rate=48000, channels=2, max_frames=4096, device=0):
    """The GPU-resident main mixer graph (reference `MixedSource` behind `Player`)."""
    from .graph import Graph as _G

    return _G(sample_rate, channels, max_frames, device)
