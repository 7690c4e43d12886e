"""The reference's OWN known-answer vectors for the hot path, run through the HIP library (tests/test_oracle_kats.py holds the same vectors
against the CPU oracle). They are the only data the reference itself holds for this path:

  src/source/file/preloaded.rs:486-533   `resampling`: [0.2, 1.0, 0.5, 0.0] at 44.1 kHz played at 48 kHz (cubic)
  src/utils/buffer.rs:660-722            clear / scale / add / copy_buffers and max_abs_sample vectors
  src/utils.rs:94-104                    lin_db_conversion
  src/utils/smoothing.rs:613-659         LinearSmoothedValue: a ramp of 5 steps ends exactly on its target; step 0.05 -> 20 pending steps

The buffer operations have no entry point of their own on the device — they are what the mixer does with its sources (add_buffers per
source, scale_buffer for a settled volume, copy for a lone source, clear for a silent block, max_abs for the silence gates) — so their
vectors travel as constant PCM sources through pg_graph_write and must come out bit for bit."""
import numpy as np
import pytest

import oracle
from phonic_amd import _capi

pytestmark = pytest.mark.gpu
SR = 48000


def f32(a):
    return np.array(a, dtype=np.float32)


def gpu_graph(max_frames=1024, sr=SR):
    from phonic_amd.graph import Graph

    return Graph(sr, 2, max_frames, 0)


def mono_source(g, values, **opts):
    """A preloaded mono file at the mixer's rate (resampler bypass, cubic.rs:53-58): `values` + the decoder's extra zero frame."""
    pcm = np.concatenate([f32(values), np.zeros(1, np.float32)])
    return g.add_voice(0, pcm, 1, g.sample_rate, fade_out_seconds=-1.0, **opts)


def test_preloaded_resampling_vector_on_the_device():
    """preloaded.rs:486-533: the reference asserts `written >= 3` and |sum(out) - sum(in)| < 0.1 for the cubic resampler; SURVEY §4 re-derives
    the three samples by hand from cubic.rs. The graph is stereo (the reference's test renders mono): the mono file is mapped to both channels."""
    for g in (gpu_graph(), oracle.OracleGraph(SR, 2, 1024)):
        file_buffer = f32([0.2, 1.0, 0.5, 0.0])      # "NB add extra tailing 0.0 sample for the cubic resampler"
        g.add_voice(0, file_buffer, 1, 44100, fade_out_seconds=-1.0)
        out = np.full(2048, 9.0, np.float32)
        assert g.write(out, 0) == 2048
        l, r = out[0::2], out[1::2]
        assert np.array_equal(l, r)
        written = int(np.nonzero(l)[0].max()) + 1
        assert written >= file_buffer.size * 44100 // 48000
        assert abs(float(l.sum(dtype=np.float32)) - float(file_buffer.sum(dtype=np.float32))) < 0.1
        np.testing.assert_allclose(l[:3], [0.2, 0.97775948, 0.59562492], rtol=0, atol=2e-7)
        assert np.all(l[3:] == 0.0)


def test_buffer_vectors_through_the_mixer():
    """buffer.rs:662-708. add_buffers: two sources -> their sum; scale_buffer: a source at volume 2.0, then its result at volume 0.5;
    copy_buffers: a lone source at unit volume; clear_buffer: a block in which nothing plays is silent whatever the caller's buffer held."""
    ramp = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    half = [0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5]

    def left(g, n=16):
        out = np.full(2 * n, 7.0, np.float32)
        assert g.write(out, 0) == 2 * n
        assert np.array_equal(out[0::2], out[1::2])
        return out[0::2]

    g = gpu_graph()                                                  # test_add_buffers_simd
    mono_source(g, ramp); mono_source(g, half)
    o = left(g)
    assert np.array_equal(o[:11], f32([1.5, 3.0, 4.5, 6.0, 7.5, 9.0, 10.5, 12.0, 13.5, 15.0, 16.5])) and np.all(o[11:] == 0.0)
    g = gpu_graph()                                                  # test_scale_buffer_simd
    mono_source(g, ramp, volume=2.0)
    o = left(g)
    assert np.array_equal(o[:11], f32([2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22]))
    g = gpu_graph()
    mono_source(g, list(o[:11]), volume=0.5)
    assert np.array_equal(left(g)[:11], f32(ramp))
    g = gpu_graph()                                                  # test_copy_buffers_simd
    mono_source(g, ramp)
    assert np.array_equal(left(g)[:11], f32(ramp))
    g = gpu_graph()                                                  # test_clear_buffer_simd
    mono_source(g, ramp, start_time=4096)
    assert np.all(left(g) == 0.0)


@pytest.mark.parametrize("peak,gate_closes", [(0.00095, True), (0.00105, False)])
def test_max_abs_sample_vector_drives_the_silence_gate(peak, gate_closes):
    """buffer.rs:710-722: max|x| of [0.1, -0.5, 0.3, -0.2, 0.15, -0.25, 0.35, -0.45, 0.05, -0.15, 0.25] is 0.5 (the NEGATIVE sample). On the
    device max_abs_sample is what SubMixerProcessor::process compares with SILENCE_THRESHOLD = 0.001 (src/source/mixed/submixer.rs:47-77): the
    vector, scaled so that its peak sits just below / just above the threshold and looped in a sub-mixer, must be dropped from the sum after 2 s
    of "silence" — or never. Against the oracle, bit for bit (no effect in the chain: pure buffer arithmetic)."""
    v = f32([0.1, -0.5, 0.3, -0.2, 0.15, -0.25, 0.35, -0.45, 0.05, -0.15, 0.25]) * np.float32(peak / 0.5)
    assert float(np.abs(v).max()) == float(np.abs(v[1]))
    outs = []
    for g in (gpu_graph(), oracle.OracleGraph(SR, 2, 1024)):
        m = g.add_mixer()
        g.add_voice(m, np.concatenate([v, np.zeros(1, np.float32)]), 1, SR, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
        outs.append(g.render(100, 1024))
    assert np.array_equal(outs[0], outs[1])
    per_block = np.abs(outs[0].reshape(100, -1)).max(axis=1)
    assert per_block[0] > 0 and per_block[90] > 0                   # audible for the first 2 s either way (96 000 frames = 93.75 blocks)
    assert (per_block[-1] == 0.0) == gate_closes


def test_lin_db_conversion_on_the_device():
    """utils.rs:94-104: db_to_linear(0) == 1 and db_to_linear(-200) == 0 exactly, round trips within 1e-4 dB. On the device db_to_linear sits in
    the Compressor's gain stage (compressor.rs:284-286: make-up 0 dB and no reduction -> the delayed input comes out bit for bit) and in the
    Gate (gate.rs:186-190: a gain at or below -60 dB is zero); the Gain effect's parameter is scaled in dB (gain.rs:67-74, scaling.rs:31)."""
    from phonic_amd import Effect as StandaloneEffect

    n = 256
    x = (0.3 * np.sin(np.arange(2 * n) * 0.05)).astype(np.float32)
    fx = StandaloneEffect(_capi.FX_COMPRESSOR, params={"thrs": 0.0, "rato": 20.0, "knee": 0.0, "gain": 0.0, "look": 0.001})   # a limiter that never limits: total gain = db_to_linear(0 - 0)
    fx.initialize(SR, 2, n)
    y = np.concatenate([fx.process(x.copy()), fx.process(np.zeros(2 * n, np.float32))])
    assert np.array_equal(y[2 * 49:2 * 49 + 2 * n], x)              # look-ahead ceil(0.001f * 48000) = 49 frames: the input, delayed, times exactly 1.0
    gate = StandaloneEffect(_capi.FX_GATE, params={"thrs": 0.0, "rnge": -60.0})                     # never opens; range -60 dB -> linear 0
    gate.initialize(SR, 2, n)
    assert np.all(gate.process(x.copy()) == 0.0)
    for db in (20.0, -20.0):                                         # the round trips of the reference test, through the Gain parameter's dB scaling
        outs = []
        for make in (lambda: StandaloneEffect(_capi.FX_GAIN), lambda: oracle.OracleEffect(_capi.FX_GAIN)):
            gain = make()
            gain.initialize(SR, 2, n)
            gain.set_parameter("gain", (db + 60.0) / 84.0, normalized=True)
            for _ in range(40):                                      # the exponential smoother settles (it stops within its own end condition of the target)
                y = gain.process(np.full(2 * n, 0.01, np.float32))
            outs.append(y)
        assert np.array_equal(outs[0], outs[1])                      # f32 smoother + one multiply per sample: bit for bit
        assert abs(20.0 * np.log10(float(outs[0][-1]) / 0.01) - db) < 0.1
        assert abs(float(oracle.lib().po_linear_to_db(oracle.lib().po_db_to_linear(db))) - db) < 1e-4      # (the reference's own assertion, on the oracle)


def test_linear_ramp_vectors_on_the_device():
    """smoothing.rs:640-659: a LinearSmoothedValue reaches its target EXACTLY when its pending steps are used up (5 steps in the reference's
    test; step 0.05 towards 1.0 -> 20 pending steps). On the device the linear smoothers drive e.g. the Distortion's drive (step 0.01 per frame
    at 44.1 kHz, distortion.rs:209-219): a drive change of 5 (20) steps must leave the memoryless shaper in its new steady state from the 5th (20th) frame
    on — bit for bit the output of an effect built with that drive — and not a frame earlier. The step counts come from the oracle's smoother."""
    import ctypes as C

    from phonic_amd import Effect as StandaloneEffect

    n = 64
    x = np.full(2 * n, 0.4, np.float32)
    for target, steps in ((0.05, 5), (0.2, 20)):
        out4 = (C.c_float * 4)()
        trace = np.zeros(1, np.float32)
        oracle.lib().po_smoother_run(1, 0.0, 44100, 0.01, 1, target, 0, 0, 0, oracle.fp(trace), out4)
        assert out4[3] == float(steps)                               # num_pending_steps (the reference asserts 20 for step 0.05 -> 1.0)
        ramped = StandaloneEffect(_capi.FX_DISTORTION, params={"type": 1, "driv": 0.0})
        steady = StandaloneEffect(_capi.FX_DISTORTION, params={"type": 1, "driv": target})
        ref = oracle.OracleEffect(_capi.FX_DISTORTION, params={"type": 1, "driv": 0.0})
        for e in (ramped, steady, ref):
            e.initialize(44100, 2, n)
        ramped.set_parameter("driv", target)
        ref.set_parameter("driv", target)
        a, s, b = ramped.process(x.copy()), steady.process(x.copy()), ref.process(x.copy())
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
        assert np.array_equal(a[2 * (steps - 1):], s[2 * (steps - 1):])          # the last of the `steps` steps lands exactly on the target: that frame already
        assert not np.array_equal(a[2 * (steps - 2):2 * (steps - 1)], s[2 * (steps - 2):2 * (steps - 1)])   # runs the target's shaper — and not a frame earlier
