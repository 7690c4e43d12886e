"""The parameter descriptors of all ten stock effects against tests/golden/params.json — ids, names, types, ranges, defaults, scalings and
enum sizes as the reference's sources declare them, in the order `Effect::parameters()` lists them (the fixture is written by
tests/golden/make_params.py from src/effect/*.rs in the build container; it is data, the .rs text never ships).

  CPU : the whole pg_effect_kind_param table of the library, field by field.
  GPU : every parameter of every effect driven to Normalized 0 / 0.5 / 1 (src/parameter/float.rs:131-141, scaling.rs:45-108, enum.rs:151-155) on
        the device and in the oracle: same resolved value, same audio."""
import json
import os

import numpy as np
import pytest

from phonic_amd import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = json.load(open(os.path.join(HERE, "golden", "params.json")))
TYPE = {"float": 0, "enum": 1, "bool": 2}
SCALING = {"linear": 0, "exponential": 1, "decibel": 2}


def test_fixture_covers_the_ten_effects():
    assert [e["name"] for e in FIXTURE["effects"]] == _capi.FX_NAMES
    assert sum(len(e["parameters"]) for e in FIXTURE["effects"]) == 63


def test_descriptor_table_equals_the_reference_consts():
    from phonic_amd.graph import effect_parameters

    lib = _capi.load()
    for kind, e in enumerate(FIXTURE["effects"]):
        assert lib.pg_effect_kind_name(kind).decode() == e["name"]
        assert lib.pg_effect_kind_weight(kind) == e["weight"]
        have = effect_parameters(kind)
        assert len(have) == len(e["parameters"]) == lib.pg_effect_kind_param_count(kind)
        for h, want in zip(have, e["parameters"]):
            ctx = (e["name"], want["id"])
            assert h["fourcc"] == _capi.fourcc(want["id"]), ctx
            assert h["name"] == want["name"], ctx
            assert h["type"] == TYPE[want["type"]], ctx
            assert np.float32(h["min"]) == np.float32(want["min"]) and np.float32(h["max"]) == np.float32(want["max"]), (ctx, h["min"], h["max"])
            assert np.float32(h["default"]) == np.float32(want["default"]), (ctx, h["default"])
            assert h["scaling"] == SCALING[want["scaling"]], ctx
            args = list(want["scaling_args"]) + [0.0, 0.0]
            assert np.float32(h["scaling_args"][0]) == np.float32(args[0]) and np.float32(h["scaling_args"][1]) == np.float32(args[1]), ctx
            assert h["n_values"] == want["n_values"], ctx


def expected_raw(p, norm):
    """ParameterValueUpdate::Normalized -> raw, as the reference resolves it: float.rs:131-141 (denormalize through the scaling, clamp),
    scaling.rs:45-74 (Linear: x; Exponential(f): x^f; Decibel(lo, hi): db_to_linear(lo + x (hi - lo)) rescaled to the range),
    enum.rs:151-155 (round(x (n - 1))), boolean: x >= 0.5."""
    if p["type"] == "enum":
        return float(int(np.floor(np.float32(norm) * np.float32(p["n_values"] - 1) + np.float32(0.5))))
    if p["type"] == "bool":
        return 1.0 if norm >= 0.5 else 0.0
    return None   # floats: compared through the oracle (same formulas, f32 rounding included)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", range(10))
def test_every_parameter_at_normalized_0_half_1_against_the_oracle(kind):
    """One effect instance per (parameter, normalized value) on the device and in the oracle: set the parameter Normalized, render six blocks of a
    broadband test signal, compare. A wrong range, default, scaling or enum size moves the resolved value and with it the audio (cutoffs, gains,
    times, shapes); the delay's two random LFO shapes are exercised through their explicit seed (they are deterministic given the seed)."""
    import oracle
    import phonic_amd
    import workloads

    e = FIXTURE["effects"][kind]
    n, blocks = 512, 6
    x = workloads.test_signal(n * blocks, seed=7 + kind, kind="noise") if hasattr(workloads, "test_signal") else None
    if x is None:
        rng = np.random.default_rng(7 + kind)
        x = (0.25 * rng.standard_normal(2 * n * blocks)).astype(np.float32)
    x = np.ascontiguousarray(x, np.float32).reshape(-1)[:2 * n * blocks]
    seeds = workloads.reverb_seeds(3) if kind == _capi.FX_REVERB else None
    checked = 0
    for p in e["parameters"]:
        for norm in (0.0, 0.5, 1.0):
            dev, ref = phonic_amd.Effect(kind, None, seeds), oracle.OracleEffect(kind, None, seeds)
            outs = []
            for fx in (dev, ref):
                fx.initialize(48000, 2, n)
                fx.set_parameter(p["id"], norm, normalized=True)
                y = x.copy()
                for b in range(blocks):
                    fx.process(y[2 * n * b:2 * n * (b + 1)])
                outs.append(y)
            d = outs[0].astype(np.float64) - outs[1].astype(np.float64)
            scale = max(1.0, float(np.abs(outs[1]).max()))
            assert np.isfinite(outs[0]).all(), (e["name"], p["id"], norm)
            assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale and float(np.abs(d).max()) <= 1e-4 * scale, (e["name"], p["id"], norm, float(np.sqrt(np.mean(d * d))))
            checked += 1
    assert checked == 3 * len(e["parameters"])
