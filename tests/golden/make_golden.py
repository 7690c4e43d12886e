#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: seeded inputs and the expected outputs of the CPU oracle (oracle/, the C++ restatement of
the reference's per-block DSP path, pinned against the reference's own known-answer tests in tests/test_oracle_kats.py).

The reference is Rust and has no toolchain in this image, so it cannot be run to produce vectors; these fixtures freeze the
oracle's answers instead: `tests/test_golden.py` checks (CPU) that today's oracle still reproduces them bit for bit — an
accidental change of the oracle shows up as a diff here — and (GPU) that the HIP path matches them within the parity tolerance
without needing the oracle at run time.

    python tests/golden/make_golden.py        # rewrites effects.npz and graphs.npz next to this file
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
import workloads  # noqa: E402
from phonic_amd import _capi  # noqa: E402

SR = 48000

# (name, kind, params, reverb seed index, [(block, fourcc, value, normalized)])
EFFECT_CASES = [
    ("gain", _capi.FX_GAIN, {"gain": 0.5, "dcfm": 2}, None, [(2, "gain", 0.9, True)]),
    ("panning", _capi.FX_PANNING, {"pan ": -0.3, "wdth": 1.5, "invr": 1}, None, [(2, "pan ", 0.8, False)]),
    ("filter", _capi.FX_FILTER, {"type": 0, "cuto": 2000.0, "fltq": 0.707}, None, [(2, "cuto", 800.0, False)]),
    ("eq5", _capi.FX_EQ5, {"gan1": 6.0, "gan2": -3.0, "gan3": 4.0, "gan4": -6.0, "gan5": 2.0, "bw_2": 1.5}, None, [(2, "frq2", 500.0, False)]),
    ("delay", _capi.FX_DELAY, {"mode": 1, "dlay": 20.0, "fdbk": 0.7, "driv": 0.5, "ftyp": 2, "wdth": 1.0}, None, [(2, "dlay", 60.0, False)]),
    ("reverb", _capi.FX_REVERB, {"room": 0.6, "wet ": 0.5}, 3, [(2, "room", 0.9, False)]),
    ("chorus", _capi.FX_CHORUS, {"rate": 3.0, "dpth": 0.8, "fdbk": -0.6, "dlay": 0.5, "fltt": 1, "fltf": 300.0, "fltq": 0.4}, None, [(2, "rate", 4.0, False)]),
    ("compressor", _capi.FX_COMPRESSOR, {"thrs": -0.01, "rato": 20.0, "knee": 0.0, "attk": 0.02, "rels": 2.0, "gain": 0.0, "look": 0.02}, None, [(2, "thrs", -30.0, False)]),
    ("gate", _capi.FX_GATE, {"thrs": -20.0, "attk": 0.002, "hold": 0.01, "rels": 0.05, "rnge": -40.0}, None, [(2, "thrs", -10.0, False)]),
    ("distortion", _capi.FX_DISTORTION, {"type": 4, "driv": 4.0}, None, [(2, "mix ", 0.3, False)]),
]
EFFECT_BLOCKS, EFFECT_FRAMES = 4, 384


def run_effect(make, case):
    name, kind, params, seed, updates = case
    e = make(kind, params, workloads.reverb_seeds(seed) if seed is not None else None)
    e.initialize(SR, 2, 4096)
    x = workloads.test_signal(EFFECT_BLOCKS * EFFECT_FRAMES, seed=100 + kind, kind="noise")
    y = x.copy()
    for blk in range(EFFECT_BLOCKS):
        for (b, id4, val, norm) in updates:
            if b == blk:
                e.set_parameter(id4, val, norm)
        sl = slice(blk * EFFECT_FRAMES * 2, (blk + 1) * EFFECT_FRAMES * 2)
        e.process(y[sl])
    return x, y


# mixer graphs: (name, builder, blocks, block frames)
def g_resampler(g):
    for i, rate in enumerate((44100, 32000, 16000, 48000, 96000)):
        g.add_voice(0, workloads.tone_buffer(i, rate, 0.05), 2, rate, volume=0.4, panning=workloads.voice_pan(i), has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)


def g_headline(g):
    workloads.build_headline(g, 4, seconds=0.2)


def g_chain_and_bus(g):
    m = g.add_mixer()
    g.add_effect(m, _capi.FX_GAIN, params={"gain": 0.7})
    g.add_effect(m, _capi.FX_PANNING, params={"pan ": -0.3})
    g.add_effect(m, _capi.FX_REVERB, params={"room": 0.45, "wet ": 0.6}, reverb_seeds=workloads.reverb_seeds(77))
    for i in range(2):
        g.add_voice(m, workloads.tone_buffer(20 + i, 44100, 0.2), 2, 44100, volume=0.4, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    m2 = g.add_mixer()
    g.add_effect(m2, _capi.FX_FILTER, params={"type": 0, "cuto": 1500.0})
    g.add_effect(m2, _capi.FX_CHORUS)
    g.add_voice(m2, workloads.tone_buffer(30, 48000, 0.1, channels=1), 1, 48000, volume=0.5, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    g.add_effect(0, _capi.FX_EQ5, params={"gan2": 3.0})
    g.add_effect(0, _capi.FX_COMPRESSOR, params={"thrs": -12.0, "rato": 4.0})


GRAPH_CASES = [("resampler", g_resampler, 4, 512), ("headline", g_headline, 4, 1024), ("chain_and_bus", g_chain_and_bus, 4, 1024)]


def run_graph(make, case):
    name, build, blocks, frames = case
    g = make(SR, 2, 1024)
    build(g)
    return g.render(blocks, frames)


def main():
    fx = {}
    for case in EFFECT_CASES:
        x, y = run_effect(oracle.OracleEffect, case)
        fx[case[0] + "_in"] = x
        fx[case[0] + "_out"] = y
    np.savez_compressed(os.path.join(HERE, "effects.npz"), **fx)
    gr = {}
    for case in GRAPH_CASES:
        gr[case[0]] = run_graph(oracle.OracleGraph, case)
    np.savez_compressed(os.path.join(HERE, "graphs.npz"), **gr)
    for f in ("effects.npz", "graphs.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
