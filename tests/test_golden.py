"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle must reproduce its frozen answers bit for bit (guards the checker itself against drift).
GPU: the HIP path, through the C ABI, must match the frozen answers within the parity tolerance (1e-5 RMS, the bound
BASELINE.json states) — these run without building or loading the oracle."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "effects.npz"))
GR = np.load(os.path.join(HERE, "golden", "graphs.npz"))


def close(a, b, rms_tol=1e-5, max_tol=1e-4):
    assert a.shape == b.shape and np.isfinite(a).all()
    d = a.astype(np.float64) - b.astype(np.float64)
    assert float(np.sqrt(np.mean(d * d))) <= rms_tol
    assert float(np.abs(d).max()) <= max_tol


@pytest.mark.parametrize("case", mg.EFFECT_CASES, ids=[c[0] for c in mg.EFFECT_CASES])
def test_oracle_reproduces_effect_vectors(case):
    import oracle

    x, y = mg.run_effect(oracle.OracleEffect, case)
    assert np.array_equal(x, FX[case[0] + "_in"])
    assert np.array_equal(y, FX[case[0] + "_out"])
    assert not np.array_equal(x, y)


@pytest.mark.parametrize("case", mg.GRAPH_CASES, ids=[c[0] for c in mg.GRAPH_CASES])
def test_oracle_reproduces_graph_vectors(case):
    import oracle

    assert np.array_equal(mg.run_graph(oracle.OracleGraph, case), GR[case[0]])
    assert np.abs(GR[case[0]]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.EFFECT_CASES, ids=[c[0] for c in mg.EFFECT_CASES])
def test_gpu_effect_matches_golden(case):
    import phonic_amd

    x, y = mg.run_effect(lambda kind, params, seeds: phonic_amd.Effect(kind, params, seeds), case)
    assert np.array_equal(x, FX[case[0] + "_in"])
    close(y, FX[case[0] + "_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.GRAPH_CASES, ids=[c[0] for c in mg.GRAPH_CASES])
def test_gpu_graph_matches_golden(case):
    from phonic_amd.graph import Graph

    close(mg.run_graph(lambda sr, ch, mf: Graph(sr, ch, mf, 0), case), GR[case[0]])


# ---- the C++ oracle vs the independent Python restatement (tests/golden/numpy_restatement.py -> independent.npz) ----------
IND = np.load(os.path.join(HERE, "golden", "independent.npz"))


@pytest.mark.parametrize("name", ["up", "down", "mono"])
def test_oracle_cubic_resampler_equals_independent_restatement(name):
    import ctypes as C

    import oracle

    x, want = IND[f"cubic_{name}_in"], IND[f"cubic_{name}_out"]
    in_rate, out_rate, nch, consumed_want = (int(v) for v in IND[f"cubic_{name}_meta"])
    out = np.zeros(500 * nch, np.float32)
    consumed = C.c_size_t(0)
    f32p = C.POINTER(C.c_float)
    produced = oracle.lib().po_cubic_resample(x.ctypes.data_as(f32p), x.size, in_rate, out_rate, nch, out.ctypes.data_as(f32p), out.size, 128 * nch, C.byref(consumed))
    assert produced == want.size and consumed.value == consumed_want
    assert np.array_equal(out[:produced], want)


@pytest.mark.parametrize("name", ["mid", "small_wet"])
def test_oracle_reverb_equals_independent_restatement(name):
    """Two restatements of reverb.rs written separately (C++ oracle, Python scalars) agree bit for bit over 1200 frames: the
    rings, vibrato phases, Householder feedback, the three low-pass biquads and the sin/asin shaping."""
    import oracle
    import workloads
    from phonic_amd import _capi

    x, want = IND[f"reverb_{name}_in"], IND[f"reverb_{name}_out"]
    room, wet, seed, block = IND[f"reverb_{name}_meta"]
    e = oracle.OracleEffect(_capi.FX_REVERB, {"room": float(room), "wet ": float(wet)}, workloads.reverb_seeds(int(seed)))
    e.initialize(48000, 2, 4096)
    y = x.copy()
    block = int(block)
    for b0 in range(0, y.size // 2, block):
        e.process(y[2 * b0:2 * (b0 + block)])
    assert np.array_equal(y, want)
    assert not np.array_equal(y, x)


# ---- the C++ oracle vs the second part of the independent restatement: the other eight effects (numpy_restatement_fx.py) -----------------
import numpy_restatement_fx as rfx  # noqa: E402

IND_FX = np.load(os.path.join(HERE, "golden", "independent_fx.npz"))


@pytest.mark.parametrize("case", rfx.CASES, ids=[c[0] for c in rfx.CASES])
def test_oracle_effect_equals_independent_restatement(case):
    """Gain, Panning, Filter, Eq5, Delay (all deterministic LFO shapes), Chorus, Compressor / limiter, Gate and Distortion — with the
    smoothers, TPT-SVF / biquad coefficient formulas, DC filter, envelope follower, interpolated and look-ahead delay lines they are built
    from — restated a second time, in Python scalars, from the Rust sources: the C++ oracle must agree bit for bit at the default
    parameters and through parameter ramps (smoother targets set between blocks, per-frame coefficient branches, type switches)."""
    import oracle

    name, kind, _cls, params, updates, _sig = case
    x, want = IND_FX[name + "_in"], IND_FX[name + "_out"]
    e = oracle.OracleEffect(kind, params, None)
    e.initialize(rfx.SR, 2, 4096)
    y = x.copy()
    for blk in range(rfx.BLOCKS):
        for pid, v in updates.get(blk, []):
            e.set_parameter(pid, v, False)
        e.process(y[blk * rfx.FRAMES * 2:(blk + 1) * rfx.FRAMES * 2])
    assert np.array_equal(y, want)


def test_independent_restatement_vectors_are_current():
    """The committed vectors are what the restatement script produces (a sample of the cases: the full set takes ~5 s)."""
    for case in rfx.CASES[::4]:
        x, y = rfx.run_case(rfx._adapt(case))
        assert np.array_equal(x, IND_FX[case[0] + "_in"]) and np.array_equal(y, IND_FX[case[0] + "_out"]), case[0]


# ---- the C++ oracle's GRAPH vs the third part of the independent restatement (numpy_restatement_graph.py) ---------------------------------
import numpy_restatement_graph as rgr  # noqa: E402

IND_GR = np.load(os.path.join(HERE, "golden", "independent_graph.npz"))


def _oracle_scenario(sc, g=None, returns=None):
    """The scenario of numpy_restatement_graph.py on the C++ oracle's graph (or the graph given), through the API every graph test uses."""
    from phonic_amd import _capi

    if g is None:
        import oracle
        g = oracle.OracleGraph(rgr.SR, 2, 512)
    fx_ids = {}
    mixers = [0]
    for mi, chain in enumerate(sc["mixers"]):
        parent = sc.get("parents", [0] * len(sc["mixers"]))[mi]
        m = g.add_mixer() if parent == 0 else g.add_mixer(mixers[parent])
        mixers.append(m)
        for fi, (name, params) in enumerate(chain):
            fx_ids[(mi + 1, fi)] = g.add_effect(m, rgr.FX[name][2], params=params)
    voices = []
    for v in sc["voices"]:
        i, rate, seconds, nch = v["tone"]
        rep = v["repeat"]
        loop = dict(has_loop_range=1, loop_start=v["loop"][0], loop_end=v["loop"][1]) if v.get("loop") else {}
        if v.get("source_rate"):
            loop["source_rate"] = v["source_rate"]
        if not v.get("transient", True):
            loop["non_transient"] = 1
        voices.append(g.add_voice(mixers[v["mixer"]], rgr.tone(i, rate, seconds, nch), nch, rate, volume=v["volume"], panning=v["panning"], start_time=v["start"],
                                  has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER if rep == rgr.USIZE_MAX else rep, **loop))
    outs, pos = [], 0
    for b, n in enumerate(sc["blocks"]):
        for kind, who, val, t in sc["actions"].get(b, []):
            if kind == "stop":
                g.stop_voice(voices[who], t)
            elif kind == "remove":
                g.remove_voice(voices[who])
            elif kind == "stop_all":
                g.stop_all_voices()
            elif kind == "volume":
                g.set_voice_volume(voices[who], val, t)
            elif kind == "panning":
                g.set_voice_panning(voices[who], val, t)
            elif kind == "speed":
                g.set_voice_speed(voices[who], val[0], t, glide=val[1])
            elif kind == "seek":
                g.seek_voice(voices[who], val[0], t)
            else:
                g.schedule_param(fx_ids[who], val[0], val[1], t)
        o = np.zeros(2 * n, np.float32)
        w = g.write(o, pos)
        assert w in (0, 2 * n)
        if returns is not None:
            returns.append(w)
        outs.append(o)
        pos += n
    return np.concatenate(outs)


@pytest.mark.parametrize("name", sorted(rgr.SCENARIOS))
def test_oracle_graph_equals_independent_restatement(name):
    """The graph level — PreloadedFileSource (repeat, end of file, stop with fade-out), mono -> stereo mapping, smoothed source volume and
    panning, MixedSource::write (sample-time events splitting blocks, start / stop times, removal of exhausted sources, sub-mixers, the effect
    chain; loop range with a finite repeat count, pitch glide in 64-frame steps, seek, immediate speed change), EffectProcessor's auto-bypass with
    known tails and with silence detection, SubMixerProcessor's 2 s silence gate — restated a
    second time in Python from the Rust sources: the C++ oracle's graph must agree bit for bit over the whole run (the sub-mixer scenario
    runs 43 520 frames at 8 kHz: the voice ends, the Gain bypasses at once, the Filter after its tail, the Delay after 2 s of silence, the
    sub-mixer's gate 2 s after its output fell silent)."""
    want = IND_GR[name]
    got = _oracle_scenario(rgr.SCENARIOS[name])
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        bad = np.flatnonzero(got != want)
        raise AssertionError(f"{name}: {bad.size} samples differ, first at frame {bad[0] // 2} (oracle {got[bad[0]]!r}, restatement {want[bad[0]]!r}), "
                             f"max |d| {float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())}")


def test_independent_graph_vectors_are_current():
    """The committed vectors are what the restatement script produces (the short scenario; the long one takes a few seconds)."""
    assert np.array_equal(rgr.run_scenario(rgr.SCENARIOS["sources"]), IND_GR["sources"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(rgr.SCENARIOS))
def test_gpu_graph_matches_independent_restatement(name):
    """The HIP graph against the restatement's vectors directly (no oracle in between), at the scenarios' 8 kHz mixer rate: sources with start /
    stop / fade-out / volume and panning events, and the 43 520-frame bypass scenario (known tails, silence detection, the sub-mixer's gate)."""
    from phonic_amd.graph import Graph

    want = IND_GR[name]
    got = _oracle_scenario(rgr.SCENARIOS[name], Graph(rgr.SR, 2, 512, 0))
    close(got, want)
    if name == "submixer_bypass":   # behind every gate of the long scenario both sides are exactly silent
        assert not np.any(got[2 * 23000:]) and not np.any(want[2 * 23000:])


def test_non_transient_sources_write_returns():
    """What MixedSource::write returns over the non-transient scenario (mixed.rs:664-670): the block's length for as long as the mixer holds a
    source — also an exhausted one it keeps — and 0 once RemoveSource has taken the last: oracle and restatement agree call for call."""
    import copy

    sc = copy.deepcopy(rgr.SCENARIOS["non_transient"])
    sc["returns"] = []
    rgr.run_scenario(sc)
    got = []
    _oracle_scenario(rgr.SCENARIOS["non_transient"], returns=got)
    assert got == sc["returns"]
    assert got[:20] == [512] * 20 and got[20:] == [0] * 10     # voice 0 ended after 800 frames and was kept until block 20


@pytest.mark.gpu
def test_gpu_non_transient_sources_and_remove_voice():
    """pg_voice_options::non_transient + pg_graph_remove_voice (MixerMessage::AddSource{is_transient: false} / RemoveSource, mixed.rs:117-123,
    149-151,298-305,400-402,612-616,715) on the HIP graph and through the sharded handle: against the restatement's vectors, and the returns of
    write call for call (a kept source keeps the mixer alive; once the last is removed write returns 0)."""
    from phonic_amd.graph import Graph, ShardedGraph

    want = IND_GR["non_transient"]
    for g in (Graph(rgr.SR, 2, 512, 0), ShardedGraph([0, 0], rgr.SR, 2, 512)):
        rets = []
        got = _oracle_scenario(rgr.SCENARIOS["non_transient"], g, returns=rets)
        close(got, want)
        assert rets[:20] == [512] * 20 and rets[21:] == [0] * 9, rets
        import phonic_amd
        with pytest.raises(phonic_amd.PhonicError):
            g.remove_voice(0)       # gone: PG_ERR_NOT_FOUND
