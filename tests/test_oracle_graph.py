"""Mixer-graph semantics of the CPU oracle (MixedSource, EffectProcessor, SubMixerProcessor; reference
src/source/mixed.rs, mixed/effect.rs, mixed/submixer.rs): integer scheduling arithmetic, bypass/tail logic, edge cases."""
import numpy as np

import oracle
import workloads
from phonic_amd import _capi

SR = 48000


def test_empty_mixer_returns_zero_and_leaves_buffer():
    g = oracle.OracleGraph(SR, 2)
    out = np.full(256, 3.0, np.float32)
    assert g.write(out, 0) == 0  # mixed.rs:664-670
    assert np.all(out == 3.0)


def test_source_start_time_is_sample_accurate():
    g = oracle.OracleGraph(SR, 2)
    buf = np.ones(2 * 101, np.float32)
    buf[-2:] = 0  # extra zero frame
    g.add_voice(0, buf, 2, SR, start_time=300)
    out = np.zeros(2 * 1024, np.float32)
    assert g.write(out, 0) == 2048
    assert np.all(out[:600] == 0) and np.all(out[600:800] == 1.0) and out[802] == 0.0


def test_event_splits_block_at_sample_time():
    """A Gain parameter event at frame 517 takes effect exactly there (event.rs:41-50, mixed.rs:679-712)."""
    g = oracle.OracleGraph(SR, 2)
    buf = np.concatenate([np.full(2 * 4000, 0.5, np.float32), np.zeros(2, np.float32)])
    g.add_voice(0, buf, 2, SR)
    fx = g.add_effect(0, _capi.FX_GAIN)
    g.schedule_param(fx, "gain", 0.25, 517)
    out = np.zeros(2 * 1024, np.float32)
    g.write(out, 0)
    assert np.all(out[: 2 * 517] == 0.5)
    assert out[2 * 517] < 0.5 and out[2 * 517] > 0.49  # exponential ramp starts at the event frame
    assert out[-1] < out[2 * 600]


def test_stop_with_fade_out_then_source_is_dropped():
    g = oracle.OracleGraph(SR, 2)
    buf = np.concatenate([np.full(2 * 48000, 0.5, np.float32), np.zeros(2, np.float32)])
    v = g.add_voice(0, buf, 2, SR)
    g.stop_voice(v, 100)
    out = np.zeros(2 * 4096, np.float32)
    assert g.write(out, 0) == 8192
    assert np.all(out[:200] == 0.5)
    assert 0 < out[2 * 2400] < 0.01  # 50 ms fade-out = 2400 frames to ~1%
    rets = [g.write(out, 4096 * (i + 1)) for i in range(3)]
    assert rets[-1] == 0  # fader finished -> exhausted -> transient source removed -> mixer empty


def test_effect_auto_bypass_after_tail():
    """Gain has tail 0: with a silent input the processor bypasses it on the block after the source ended
    (effect.rs:88-145); the reverb on the bus keeps running for its tail."""
    g = oracle.OracleGraph(SR, 2)
    buf = np.concatenate([np.full(2 * 100, 0.5, np.float32), np.zeros(2, np.float32)])
    g.add_voice(0, buf, 2, SR)
    g.add_effect(0, _capi.FX_REVERB, reverb_seeds=workloads.reverb_seeds(0))
    outs = [np.zeros(2048, np.float32) for _ in range(4)]
    for i, o in enumerate(outs):
        assert g.write(o, 1024 * i) == 2048
    assert np.abs(outs[3]).max() > 1e-6  # reverb tail still audible (tail ~ 270k frames)


def test_submixer_silence_gate_two_seconds():
    """A sub-mixer without sources returns 0 samples; it counts as silent and stops contributing to `audible_input`
    after 2 s (submixer.rs:47-77) — observable through the bus effect's bypass: Gain leaves the cleared buffer alone."""
    g = oracle.OracleGraph(SR, 2)
    g.add_mixer()
    out = np.full(2 * 4096, 1.0, np.float32)
    for i in range(30):
        assert g.write(out, 4096 * i) == 8192
        assert np.all(out == 0.0)


def test_render_is_chunking_invariant_for_sources():
    """write() in 1 x 4096 or 4 x 1024 frames gives identical source output (state carried across calls)."""
    def build():
        g = oracle.OracleGraph(SR, 2)
        for i in range(3):
            g.add_voice(0, workloads.tone_buffer(i, 44100, 0.05), 2, 44100, volume=0.4, panning=workloads.voice_pan(i), has_repeat=1,
                        repeat=_capi.PG_REPEAT_FOREVER)
        return g
    a = build().render(1, 4096)
    b = build().render(4, 1024)
    assert np.array_equal(a, b)


def test_normalized_parameter_updates():
    """Normalized updates go through the parameter scaling (float.rs:137-141, scaling.rs:45-74)."""
    e = oracle.OracleEffect(_capi.FX_FILTER)
    e.initialize(SR, 2, 4096)
    e.set_parameter("cuto", 0.5, normalized=True)  # 20 + 0.5^2.5 * 19980 = 3552.0...
    x = workloads.test_signal(4096, seed=1)
    e.process(x)
    e2 = oracle.OracleEffect(_capi.FX_FILTER)
    e2.initialize(SR, 2, 4096)
    e2.set_parameter("cuto", float(np.float32(20.0) + np.float32(0.5) ** np.float32(2.5) * np.float32(19980.0)))
    y = workloads.test_signal(4096, seed=1)
    e2.process(y)
    np.testing.assert_allclose(x, y, atol=1e-6)


def test_nested_mixers_without_effects_equal_the_flat_sum():
    """Player::add_mixer(parent) (player.rs:771-822): a chain of effect-less sub-mixers only passes the block up (x + 0 is exact),
    so main -> A -> B -> voice renders exactly what main -> voice does; unknown parents are rejected."""
    buf = workloads.tone_buffer(9, 44100, 0.2)
    flat, nested = oracle.OracleGraph(SR, 2), oracle.OracleGraph(SR, 2)
    flat.add_voice(0, buf, 2, 44100, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    a = nested.add_mixer()
    b = nested.add_mixer(a)
    assert b != a
    nested.add_voice(b, buf, 2, 44100, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    try:
        nested.add_mixer(77)
        assert False, "unknown parent accepted"
    except Exception:
        pass
    for blk in range(4):
        x, y = np.zeros(2048, np.float32), np.zeros(2048, np.float32)
        assert flat.write(x, blk * 1024) == 2048 and nested.write(y, blk * 1024) == 2048
        assert np.array_equal(x, y) and float(np.abs(x).max()) > 0.01


def test_move_and_remove_effect_messages():
    """MixerMessage::MoveEffect / RemoveEffect (mixed.rs:433-462) with memoryless effects and a constant input: a hard-clip Distortion
    (clamp to +-1/gain, then x gain, then the RMS compensation factor c of that drive; distortion.rs hard_clip) and a Gain of 0.1.
    [Distortion, Gain]: 0.5 clips -> 1.0 c -> 0.1 c.   [Gain, Distortion]: 0.05 stays under the threshold -> 0.05 gain c.
    The moved chain must give exactly what a chain built in that order gives. Direction offsets clamp to the chain; a removed
    effect's id is unknown afterwards; the emptied chain passes the input through."""
    import pytest

    buf = np.concatenate([np.full(2 * 20000, 0.5, np.float32), np.zeros(2, np.float32)])
    out = np.zeros(2 * 1024, np.float32)

    def fresh(order):
        g = oracle.OracleGraph(SR, 2)
        g.add_voice(0, buf, 2, SR)
        ids = {}
        for k in order:
            ids[k] = g.add_effect(0, _capi.FX_DISTORTION, params={"type": 1, "driv": 1.0, "mix ": 1.0}) if k == "dist" else g.add_effect(0, _capi.FX_GAIN, params={"gain": 0.1})
        return g, ids

    def level(g, pos):
        g.write(out, pos)
        return float(out[-1])

    dg = level(fresh(["dist", "gain"])[0], 0)
    gd = level(fresh(["gain", "dist"])[0], 0)
    assert 0.005 < dg < 0.1 and gd > dg * 1.01  # 0.1 c against 0.05 gain c with gain > 2
    g, ids = fresh(["dist", "gain"])
    dist, gain = ids["dist"], ids["gain"]
    assert level(g, 0) == dg
    g.move_effect(gain, 0, _capi.MOVE_START)            # [Gain, Distortion]
    assert level(g, 1024) == gd
    g.move_effect(gain, 0, _capi.MOVE_DIRECTION, 5)     # clamped to the end: [Distortion, Gain]
    assert level(g, 2048) == dg
    g.move_effect(dist, 0, _capi.MOVE_END)              # [Gain, Distortion]
    assert level(g, 3072) == gd
    g.move_effect(dist, 0, _capi.MOVE_DIRECTION, -9)    # clamped to the start: [Distortion, Gain]
    assert level(g, 4096) == dg
    g.remove_effect(dist)                               # [Gain]
    assert abs(level(g, 5120) - 0.05) < 1e-7
    with pytest.raises(Exception):
        g.remove_effect(dist)
    with pytest.raises(Exception):
        g.schedule_param(dist, "driv", 0.5, 6000)
    with pytest.raises(Exception):
        g.move_effect(gain, 1, _capi.MOVE_END)          # not that mixer's effect
    g.remove_effect(gain)                               # []
    assert level(g, 6144) == 0.5


def test_compressor_gain_computer_is_discontinuous_at_the_upper_knee_edge():
    """Why a GPU-vs-oracle tolerance cannot hold on every input behind a Compressor: the reference's gain computer (compressor.rs:258-270)
    has the branches `t - w/2 < envelope < t + w/2` (knee) and `envelope > t + w/2` (line) and nothing for `envelope == t + w/2`, where the
    reduction falls to 0 dB — a one-frame click of (w/2)(1 - 1/ratio) = 1.3 dB at the defaults. A slowly released envelope moves by less than
    30 ulps per frame, so it lands on the edge exactly in a few percent of its crossings. This is seed 301229 of the fuzz family `rates`
    (Distortion -> Compressor after a move_effect; 44.1 kHz): the oracle's own output changes by 16 % in one frame when its input changes by
    1e-7 — the size of the difference between the device's and glibc's tanh in the Distortion in front. The device reproduces the click
    whenever its envelope is bit-equal (pg_log10f restates glibc's log10f for that: fuzz seeds 734, 888), not when an upstream effect
    differs in the last place."""
    sizes = [2048, 4096, 4096, 64, 2048, 1365, 1]

    def render(volscale):
        g = oracle.OracleGraph(44100, 2, 4096)
        m = g.add_mixer()
        g.add_effect(m, _capi.FX_COMPRESSOR, params={"attk": 0.04995590075850487, "rels": 0.8362696766853333})
        dist = g.add_effect(m, _capi.FX_DISTORTION)
        g.add_voice(m, workloads.tone_buffer(54, 22050, 0.12), 2, 22050, volume=0.4557180730637565 * volscale, panning=0.8919428354856447, has_repeat=1,
                    repeat=_capi.PG_REPEAT_FOREVER)
        chunks, pos = [], 0
        for b, n in enumerate(sizes):
            if b == 3:
                g.move_effect(dist, m, _capi.MOVE_DIRECTION, -3)
            o = np.zeros(2 * n, np.float32)
            g.write(o, pos)
            chunks.append(o)
            pos += n
        return np.concatenate(chunks).astype(np.float64)

    a, b = render(1.0), render(1.0 + 1e-7)
    d = np.abs(a - b)
    i = int(np.argmax(d))
    assert i // 2 == 13308 and d[i] > 0.1 * abs(a[i])                         # one frame, 16 % apart
    # the oracle's knee-edge log (what the GPU parity tests classify such clicks with) names that frame, and few others
    near = oracle.knee_edge_frames(lambda: render(1.0))
    assert 13308 in near and len(near) < 64, near[:16]
    d[2 * 13308:2 * 13308 + 2] = 0.0
    assert d.max() < 1e-6                                                     # every other frame follows the 1e-7
    ratio = a[2 * 13308 + 1] / b[2 * 13308 + 1]
    assert abs(20.0 * np.log10(ratio) - 1.5 * (1.0 - 1.0 / 8.0)) < 0.01       # the click is the full reduction at the knee's upper edge
