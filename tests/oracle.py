"""Test-side binding of the CPU oracle (oracle/_build/libphonic_oracle.so, prefix po_).

The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py may load it. It is built on demand with oracle/Makefile (g++)."""
import ctypes as C
import os
import subprocess

import numpy as np

from phonic_amd import _capi
from phonic_amd._wrap import EffectHandle, GraphHandle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libphonic_oracle.so")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
        path = os.environ.get("PHONIC_ORACLE_LIB") or LIB_PATH     # (tests/test_sanitizers.py: the ASan / UBSan build of the same sources)
        if path == LIB_PATH and (not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)):
            build()
        l = C.CDLL(path)
        _capi.declare(l, "po_")
        P, f32, sz = C.POINTER, C.c_float, C.c_size_t
        l.po_clear_buffer.argtypes = [P(f32), sz]
        l.po_scale_buffer.argtypes = [P(f32), sz, f32]
        l.po_add_buffers.argtypes = [P(f32), P(f32), sz]
        l.po_copy_buffers.argtypes = [P(f32), P(f32), sz]
        l.po_max_abs_sample.argtypes = [P(f32), sz]
        l.po_max_abs_sample.restype = f32
        l.po_remap_buffer_channels.argtypes = [P(f32), sz, P(f32), sz, sz]
        l.po_db_to_linear.argtypes = [f32]
        l.po_db_to_linear.restype = f32
        l.po_linear_to_db.argtypes = [f32]
        l.po_linear_to_db.restype = f32
        l.po_panning_factors.argtypes = [f32, P(f32), P(f32)]
        l.po_sine_approx.argtypes = [f32]
        l.po_sine_approx.restype = f32
        l.po_smoother_run.argtypes = [C.c_int, f32, C.c_uint32, f32, C.c_int, f32, C.c_int, C.c_uint32, C.c_uint32, P(f32), P(f32)]
        l.po_biquad_run.argtypes = [C.c_int, C.c_uint32, f32, f32, f32, P(f32), sz]
        l.po_svf_run.argtypes = [C.c_int, C.c_uint32, f32, f32, P(f32), sz]
        l.po_dc_run.argtypes = [C.c_int, C.c_uint32, P(f32), sz]
        l.po_cubic_resample.argtypes = [P(f32), sz, C.c_uint32, C.c_uint32, sz, P(f32), sz, sz, P(sz)]
        l.po_cubic_resample.restype = sz
        l.po_allpass_run.argtypes = [sz, sz, P(C.c_double), sz]
        l.po_interp_delay_run.argtypes = [sz, f32, f32, P(f32), sz]
        l.po_effect_reverb_state.argtypes = [C.c_void_p, P(C.c_double), P(C.c_uint64)]
        l.po_graphs_render_parallel.argtypes = [P(C.c_void_p), C.c_int, C.c_int, P(f32), sz, sz, C.c_uint64]
        l.po_small_rng_run.argtypes = [P(C.c_uint64), sz, P(C.c_uint64), P(f32)]
        l.po_small_rng_run.restype = None
        l.po_index_log_begin.argtypes = []
        l.po_index_log_begin.restype = None
        l.po_index_log_end.argtypes = [P(C.c_int32), sz]
        l.po_index_log_end.restype = sz
        l.po_knee_log_begin.argtypes = []
        l.po_knee_log_begin.restype = None
        l.po_knee_log_end.argtypes = [P(C.c_uint64), sz]
        l.po_knee_log_end.restype = sz
        _LIB = l
    return _LIB


_NATIVE = None


def lib_native():
    """(library, flags) for the cpu_baseline leg of bench.py: oracle/_build/libphonic_oracle_native.so (-O3 -march=native, `make native`,
    built on the host it is timed on — it must not travel between hosts), else the portable build."""
    global _NATIVE
    if _NATIVE is None:
        portable_flags = "-O2 -ffp-contract=off -fno-fast-math (portable build)"
        try:
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "native"], check=True, capture_output=True, timeout=300)
            l = C.CDLL(os.path.join(ORACLE_DIR, "_build", "libphonic_oracle_native.so"))
            _capi.declare(l, "po_")
            l.po_graphs_render_parallel.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_uint64]
            l.po_build_flags.restype = C.c_char_p
            _NATIVE = (l, l.po_build_flags().decode())
        except Exception:
            _NATIVE = (lib(), portable_flags)
    return _NATIVE


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleEffect(EffectHandle):
    def __init__(self, kind, params=None, reverb_seeds=None, lfo_seed=None):
        super().__init__(lib(), "po_", kind, params, reverb_seeds, lfo_seed=lfo_seed)

    def reverb_state(self):
        ph = (C.c_double * 16)()
        cnt = (C.c_uint64 * 8)()
        assert lib().po_effect_reverb_state(self._h, ph, cnt) == 0
        return np.array(ph[:]), np.array(cnt[:], dtype=np.uint64)


class OracleGraph(GraphHandle):
    def __init__(self, sample_rate=48000, channels=2, max_frames=4096, library=None):
        super().__init__(library or lib(), "po_", sample_rate, channels, max_frames, 0)


def knee_edge_frames(render):
    """Sample times at which a Compressor's envelope came within 64 ulps of its knee's upper edge while `render()` — something that pulls oracle
    graphs on this thread — ran: where the reference's gain computer is discontinuous (compressor.rs:258-270, po_utils.hpp log_knee_edge)."""
    l = lib()
    l.po_knee_log_begin()
    try:
        render()
    finally:
        cap = 1 << 16
        buf = (C.c_uint64 * cap)()
        n = int(l.po_knee_log_end(buf, cap))
    return sorted(set(int(buf[i]) for i in range(min(n, cap))))
