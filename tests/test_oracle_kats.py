"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §8c). Each test names the reference test it restates.

  src/utils/buffer.rs:621-797       clear/scale/add/copy_buffers, max_abs_sample (exact)
  src/utils.rs:94-104               lin_db_conversion
  src/utils/smoothing.rs:556-728    the three smoothers (behavioural + num_pending_steps == 20)
  src/source/file/preloaded.rs:486-533   resampling (cubic branch; rubato is out of scope)
"""
import ctypes as C
import math

import numpy as np

import oracle
from oracle import fp


def f32(a):
    return np.array(a, dtype=np.float32)


# ---- src/utils/buffer.rs tests -----------------------------------------------------------------
def test_clear_buffer_simd(oracle_lib):  # buffer.rs:662-670
    b = f32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
    oracle_lib.po_clear_buffer(fp(b), b.size)
    assert np.array_equal(b, np.zeros(11, np.float32))


def test_scale_buffer_simd(oracle_lib):  # buffer.rs:672-686
    b = f32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
    oracle_lib.po_scale_buffer(fp(b), b.size, 2.0)
    assert np.array_equal(b, f32([2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22]))
    oracle_lib.po_scale_buffer(fp(b), b.size, 0.5)
    assert np.array_equal(b, f32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]))


def test_add_buffers_simd(oracle_lib):  # buffer.rs:688-697
    d = f32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
    s = f32([0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5])
    oracle_lib.po_add_buffers(fp(d), fp(s), d.size)
    assert np.array_equal(d, f32([1.5, 3.0, 4.5, 6.0, 7.5, 9.0, 10.5, 12.0, 13.5, 15.0, 16.5]))


def test_copy_buffers_simd(oracle_lib):  # buffer.rs:699-708
    d = np.zeros(11, np.float32)
    s = f32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
    oracle_lib.po_copy_buffers(fp(d), fp(s), d.size)
    assert np.array_equal(d, s)


def test_max_abs_sample_simd(oracle_lib):  # buffer.rs:710-722
    b = f32([0.1, -0.5, 0.3, -0.2, 0.15, -0.25, 0.35, -0.45, 0.05, -0.15, 0.25])
    assert oracle_lib.po_max_abs_sample(fp(b), b.size) == 0.5
    e = np.zeros(0, np.float32)
    assert oracle_lib.po_max_abs_sample(fp(e), 0) == 0.0


def test_remap_mono_stereo(oracle_lib):  # buffer.rs:209-229 (documented behaviour; no reference test)
    mono = f32([1, 2, 3])
    st = np.zeros(6, np.float32)
    oracle_lib.po_remap_buffer_channels(fp(mono), 1, fp(st), 2, 3)
    assert np.array_equal(st, f32([1, 1, 2, 2, 3, 3]))
    back = np.zeros(3, np.float32)
    st2 = f32([1, 3, 2, 4, 0.5, 0.25])
    oracle_lib.po_remap_buffer_channels(fp(st2), 2, fp(back), 1, 3)
    assert np.array_equal(back, f32([2, 3, 0.375]))


# ---- src/utils.rs:94-104 -----------------------------------------------------------------------
def test_lin_db_conversion(oracle_lib):
    L = oracle_lib
    assert L.po_linear_to_db(1.0) == 0.0
    assert L.po_linear_to_db(0.0) == -200.0
    assert L.po_db_to_linear(-200.0) == 0.0
    assert L.po_db_to_linear(0.0) == 1.0
    assert abs(L.po_linear_to_db(L.po_db_to_linear(20.0)) - 20.0) < 1e-4
    assert abs(L.po_linear_to_db(L.po_db_to_linear(-20.0)) + 20.0) < 1e-4
    assert math.isnan(L.po_db_to_linear(float("nan")))
    assert math.isnan(L.po_linear_to_db(-1.0))


def test_panning_factors_constant_power(oracle_lib):  # utils.rs:56-62: l^2 + r^2 == 2
    for p in np.linspace(-1, 1, 41):
        l, r = C.c_float(), C.c_float()
        oracle_lib.po_panning_factors(float(p), C.byref(l), C.byref(r))
        assert abs(l.value**2 + r.value**2 - 2.0) < 1e-5
    l, r = C.c_float(), C.c_float()
    oracle_lib.po_panning_factors(0.0, C.byref(l), C.byref(r))
    assert abs(l.value - 1.0) < 1e-6 and abs(r.value - 1.0) < 1e-6


# ---- src/utils/smoothing.rs tests --------------------------------------------------------------
def smoother(L, kind, init, sr, target, arg=None, duration=None, n_ramps=0):
    out4 = (C.c_float * 4)()
    trace = np.zeros(max(n_ramps, 1), np.float32)
    L.po_smoother_run(kind, init, sr, 0.0 if arg is None else arg, 0 if arg is None else 1, target, 0 if duration is None else 1,
                      0 if duration is None else duration, n_ramps, fp(trace), out4)
    return dict(current=out4[0], target=out4[1], need_ramp=out4[2] != 0.0, extra=out4[3], trace=trace[:n_ramps])


def test_exp_smoothed_value(oracle_lib):  # smoothing.rs:556-611
    L = oracle_lib
    r = smoother(L, 0, 0.0, 44100, 0.0)
    assert r["current"] == 0.0 and r["target"] == 0.0 and not r["need_ramp"]
    r = smoother(L, 0, 0.0, 44100, 1.0)
    assert r["target"] == 1.0 and r["need_ramp"]
    r = smoother(L, 0, 0.0, 44100, 1.0, n_ramps=1)
    assert r["current"] > 0.0
    assert r["current"] == np.float32(1.0) * np.float32(1.0 / 256.0)  # (1-0)*inertia*comp(=1)
    r = smoother(L, 0, 0.0, 44100, 1.0, n_ramps=10)
    assert 0.0 < r["current"] < 1.0 and r["need_ramp"]
    hi = smoother(L, 0, 0.0, 44100, 1.0, arg=0.1, n_ramps=1)["current"]
    lo = smoother(L, 0, 0.0, 44100, 1.0, arg=0.01, n_ramps=1)["current"]
    assert hi > lo


def test_linear_smoothed_value(oracle_lib):  # smoothing.rs:613-659
    L = oracle_lib
    r = smoother(L, 1, 0.0, 44100, 0.0)
    assert r["current"] == 0.0 and not r["need_ramp"]
    r = smoother(L, 1, 0.0, 44100, 1.0, duration=10, n_ramps=1)
    assert r["target"] == 1.0 and r["current"] > 0.0
    r = smoother(L, 1, 0.0, 44100, 1.0, duration=5, n_ramps=5)
    assert not r["need_ramp"] and r["current"] == 1.0 and r["target"] == 1.0
    # set_step(0.05); set_target_with_duration(1.0, None) -> num_pending_steps == ceil(1/0.05) == 20
    r = smoother(L, 1, 0.0, 44100, 1.0, arg=0.05)
    assert r["need_ramp"] and r["extra"] == 20.0


def test_spring_smoothed_value(oracle_lib):  # smoothing.rs:661-727
    L = oracle_lib
    r = smoother(L, 2, 0.0, 44100, 0.0)
    assert r["current"] == 0.0 and not r["need_ramp"]
    r = smoother(L, 2, 0.0, 44100, 1.0)
    assert r["need_ramp"]
    r = smoother(L, 2, 0.0, 44100, 1.0, n_ramps=1)
    assert r["current"] > 0.0
    duration = 4410
    r = smoother(L, 2, 0.0, 44100, 1.0, arg=duration, n_ramps=duration)
    assert abs(r["current"] - 1.0) < 0.05
    r = smoother(L, 2, 0.0, 44100, 1.0, arg=duration, n_ramps=duration * 4)
    assert r["trace"].max() <= 1.0 + 1e-4
    r = smoother(L, 2, 0.0, 44100, 1.0, arg=duration, n_ramps=200)
    assert r["extra"] > 0.0  # velocity


# ---- rand ^0.9 SmallRng (third-party, un-vendored): the generator behind the LFO's Random / Smooth Random shapes --------------------
def test_small_rng_is_xoshiro256plusplus(oracle_lib):
    """The reference's `SmallRng` on 64-bit targets is Xoshiro256++ (rand 0.9: rngs/small.rs -> xoshiro256plusplus.rs). Known answers of the
    published reference implementation (xoshiro256plusplus.c, Blackman / Vigna) for the state {1, 2, 3, 4} — the vector rand's own unit
    test holds — and `random::<f32>()` = the upper 24 bits of next_u32 (= bits 63..40 of next_u64) times 2^-24 (StandardUniform)."""
    state = (C.c_uint64 * 4)(1, 2, 3, 4)
    u = (C.c_uint64 * 10)()
    f = np.zeros(10, np.float32)
    oracle_lib.po_small_rng_run(state, 10, u, fp(f))
    expected = [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162, 14011001112246962877,
                12406186145184390807, 15849039046786891736, 10450023813501588000]
    assert list(u) == expected
    assert np.array_equal(f, np.array([(x >> 40) * 2.0**-24 for x in expected], np.float32))
    # the first step by hand: rotl(s0 + s3, 23) + s0 = (5 << 23) + 1
    assert expected[0] == (5 << 23) + 1


# ---- src/source/file/preloaded.rs:486-533 resampling (Default = cubic) ------------------------
def test_preloaded_resampling_cubic():
    g = oracle.OracleGraph(sample_rate=48000, channels=1)
    file_buffer = f32([0.2, 1.0, 0.5, 0.0])  # "NB add extra tailing 0.0 sample for the cubic resampler"
    g.add_voice(0, file_buffer, 1, 44100)
    out = np.zeros(1024, np.float32)
    assert g.write(out, 0) == 1024  # MixedSource clears and returns the whole block
    written = int(np.nonzero(out)[0].max()) + 1
    expected_output = file_buffer.size * 44100 // 48000
    assert written >= expected_output
    assert abs(float(out.sum(dtype=np.float32)) - float(file_buffer.sum(dtype=np.float32))) < 0.1
    # the cubic case re-derived by hand from cubic.rs
    np.testing.assert_allclose(out[:3], [0.2, 0.97775948, 0.59562492], rtol=0, atol=2e-7)
    assert np.all(out[3:] == 0.0)


def test_cubic_hermite_reproduces_cubics(oracle_lib):
    """4-point 3rd-order Hermite (cubic.rs:125-142) is exact for polynomials of degree <= 2 and reproduces a
    cubic up to the Catmull-Rom tangent error; linear ramps must come out exactly linear."""
    n = 400
    x = (np.arange(n, dtype=np.float64) * 0.01).astype(np.float32)
    out = np.zeros(420, np.float32)
    consumed = C.c_size_t()
    produced = oracle_lib.po_cubic_resample(fp(x), n, 44100, 48000, 1, fp(out), out.size, 64, C.byref(consumed))
    assert produced > 300
    ratio = np.float32(44100 / 48000)
    # after the 3-sample preload y0 = x[0]: output k sits at input position k*ratio; the first interval sees the
    # zero-initialised ym1 (cubic.rs:19), so the ramp is exact only from position >= 1 on
    k = np.arange(2, 200)
    expect = k * float(ratio) * 0.01
    np.testing.assert_allclose(out[2:200], expect, atol=3e-6)
    assert out[0] == 0.0


def test_cubic_chunking_invariance(oracle_lib):
    """The f32 sub_pos schedule must not depend on how the output is chunked (state carries across calls)."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal(2000).astype(np.float32)
    outs = []
    for chunk in (1, 7, 64, 512, 4096):
        out = np.zeros(2100, np.float32)
        consumed = C.c_size_t()
        p = oracle_lib.po_cubic_resample(fp(x), x.size, 44100, 48000, 1, fp(out), out.size, chunk, C.byref(consumed))
        outs.append((p, consumed.value, out.copy()))
    for p, c, o in outs[1:]:
        assert p == outs[0][0] and c == outs[0][1]
        assert np.array_equal(o, outs[0][2])
