"""The parity tests drive the product (pg_) and the oracle (po_) through the SAME Python wrapper (phonic_amd/_wrap.py + _capi.py): a
marshalling slip there — a struct field out of place, an argument type narrowed — would cancel on both sides. This test drives the oracle
a second time through a binding of its own, written here from include/phonic_gpu.h by hand (own ctypes structs, own argtypes, raw
calls), on a graph that gives every field of pg_voice_options and pg_effect_init a distinctive value, and requires the two renders to be
bit-identical. (tests/test_c_client.py does the same for the product with a C99 program.)"""
import ctypes as C
import os

import numpy as np

import oracle
import workloads
from phonic_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class RawInit(C.Structure):  # pg_effect_init, include/phonic_gpu.h
    _fields_ = [("n_params", C.c_uint32), ("fourcc", C.c_uint32 * 16), ("value", C.c_float * 16), ("has_reverb_seeds", C.c_uint32),
                ("reverb_fpd_l", C.c_uint32), ("reverb_fpd_r", C.c_uint32), ("reverb_vib_phase", C.c_double * 16)]


class RawVoice(C.Structure):  # pg_voice_options
    _fields_ = [("volume", C.c_float), ("panning", C.c_float), ("speed", C.c_double), ("repeat", C.c_uint64), ("has_repeat", C.c_uint32),
                ("has_loop_range", C.c_uint32), ("loop_start", C.c_uint64), ("loop_end", C.c_uint64), ("start_time", C.c_uint64),
                ("fade_in_seconds", C.c_float), ("fade_out_seconds", C.c_float), ("source_rate", C.c_uint32), ("non_transient", C.c_uint32)]


def cc(s):
    b = s.encode()
    return (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]


def test_oracle_through_an_independent_binding_equals_the_shared_wrapper():
    oracle.lib()  # built
    raw = C.CDLL(oracle.LIB_PATH)
    vp, f32p = C.c_void_p, C.POINTER(C.c_float)
    raw.po_graph_create.restype = vp
    raw.po_graph_create.argtypes = [C.c_uint32, C.c_uint32, C.c_size_t, C.c_int]
    raw.po_graph_add_mixer.argtypes = [vp]
    raw.po_graph_add_effect.argtypes = [vp, C.c_int, C.c_int, C.POINTER(RawInit)]
    raw.po_graph_add_voice.argtypes = [vp, C.c_int, f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(RawVoice)]
    raw.po_graph_schedule_param.argtypes = [vp, C.c_int, C.c_uint32, C.c_float, C.c_int, C.c_uint64]
    raw.po_graph_set_voice_volume.argtypes = [vp, C.c_int, C.c_float, C.c_uint64]
    raw.po_graph_set_voice_speed.argtypes = [vp, C.c_int, C.c_double, C.c_float, C.c_uint64]
    raw.po_graph_seek_voice.argtypes = [vp, C.c_int, C.c_double, C.c_uint64]
    raw.po_graph_stop_voice.argtypes = [vp, C.c_int, C.c_uint64]
    raw.po_graph_write.restype = C.c_size_t
    raw.po_graph_write.argtypes = [vp, f32p, C.c_size_t, C.c_uint64]
    raw.po_graph_destroy.argtypes = [vp]

    pcm_a = workloads.tone_buffer(3, 44100, 0.2)
    pcm_b = workloads.tone_buffer(5, 32000, 0.25, channels=1)
    seeds = workloads.reverb_seeds(17)
    blocks, N = 10, 1000

    # ---- through the shared wrapper
    g = oracle.OracleGraph(48000, 2, N)
    m = g.add_mixer()
    rv = g.add_effect(m, _capi.FX_REVERB, params={"room": 0.45, "wet ": 0.6}, reverb_seeds=seeds)
    dl = g.add_effect(m, _capi.FX_DELAY, params={"dlay": 23.0, "fdbk": 0.4, "mode": 1})
    va = g.add_voice(m, pcm_a, 2, 44100, volume=0.7, panning=-0.35, speed=1.1, has_repeat=1, repeat=3, has_loop_range=1, loop_start=200, loop_end=6000,
                     start_time=333, fade_in_seconds=0.02, fade_out_seconds=0.03)
    vb = g.add_voice(0, pcm_b, 1, 32000, volume=0.4, panning=0.8, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER, source_rate=24000)
    want = np.zeros((blocks, 2 * N), np.float32)
    for b in range(blocks):
        if b == 2:
            g.schedule_param(rv, "wet ", 0.2, b * N + 123)
            g.schedule_param(dl, "fdbk", 0.9, b * N + 500, normalized=True)
            g.set_voice_volume(vb, 0.1, b * N + 77)
            g.set_voice_speed(va, 0.8, b * N + 640, glide=36.0)
        if b == 5:
            g.seek_voice(va, 0.05, b * N + 10)
            g.stop_voice(vb, b * N + 400)
        g.write(want[b], b * N)

    # ---- through raw calls
    h = raw.po_graph_create(48000, 2, N, 0)
    m2 = raw.po_graph_add_mixer(h)
    ini = RawInit()
    ini.n_params = 2
    ini.fourcc[0], ini.value[0] = cc("room"), 0.45
    ini.fourcc[1], ini.value[1] = cc("wet "), 0.6
    ini.has_reverb_seeds, ini.reverb_fpd_l, ini.reverb_fpd_r = 1, seeds[0], seeds[1]
    for i in range(16):
        ini.reverb_vib_phase[i] = seeds[2][i]
    rv2 = raw.po_graph_add_effect(h, m2, 5, C.byref(ini))
    ini = RawInit()
    ini.n_params = 3
    for i, (k, v) in enumerate((("dlay", 23.0), ("fdbk", 0.4), ("mode", 1.0))):
        ini.fourcc[i], ini.value[i] = cc(k), v
    dl2 = raw.po_graph_add_effect(h, m2, 4, C.byref(ini))
    o = RawVoice(volume=0.7, panning=-0.35, speed=1.1, repeat=3, has_repeat=1, has_loop_range=1, loop_start=200, loop_end=6000, start_time=333,
                 fade_in_seconds=0.02, fade_out_seconds=0.03, source_rate=0, non_transient=0)
    va2 = raw.po_graph_add_voice(h, m2, pcm_a.ctypes.data_as(f32p), pcm_a.size // 2, 2, 44100, C.byref(o))
    o = RawVoice(volume=0.4, panning=0.8, speed=1.0, repeat=2**64 - 1, has_repeat=1, has_loop_range=0, loop_start=0, loop_end=0, start_time=0,
                 fade_in_seconds=0.0, fade_out_seconds=0.05, source_rate=24000, non_transient=0)
    vb2 = raw.po_graph_add_voice(h, 0, pcm_b.ctypes.data_as(f32p), pcm_b.size, 1, 32000, C.byref(o))
    assert (m2, rv2, dl2, va2, vb2) == (m, rv, dl, va, vb)
    got = np.zeros((blocks, 2 * N), np.float32)
    for b in range(blocks):
        if b == 2:
            assert raw.po_graph_schedule_param(h, rv2, cc("wet "), 0.2, 0, b * N + 123) == 0
            assert raw.po_graph_schedule_param(h, dl2, cc("fdbk"), 0.9, 1, b * N + 500) == 0
            assert raw.po_graph_set_voice_volume(h, vb2, 0.1, b * N + 77) == 0
            assert raw.po_graph_set_voice_speed(h, va2, 0.8, 36.0, b * N + 640) == 0
        if b == 5:
            assert raw.po_graph_seek_voice(h, va2, 0.05, b * N + 10) == 0
            assert raw.po_graph_stop_voice(h, vb2, b * N + 400) == 0
        raw.po_graph_write(h, got[b].ctypes.data_as(f32p), 2 * N, b * N)
    raw.po_graph_destroy(h)
    assert np.array_equal(got, want)
    assert np.abs(want).max() > 1e-2 and not np.array_equal(want[1], want[6])
