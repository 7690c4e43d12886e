"""Seeded random graphs against the oracle: random effect chains (every kind, random in-range parameters) on sub-mixers and on
the bus, random source rates, ragged block sizes, a few scheduled parameter / voice events. Exercises the kernel selection logic
(lean / wide fused kernels, staged kernels, generic kernel hand-over) and every time-parallel effect path in combination."""
import os
import struct

import numpy as np
import pytest

import oracle
import workloads
from phonic_amd import _capi

pytestmark = pytest.mark.gpu
SR = 48000
# A wider campaign than the suite's default (run by hand on a GPU box, e.g. PHONIC_FUZZ_SEEDS=1500 PHONIC_FUZZ_BASE=48): more seeds, other seeds.
FUZZ_SEEDS = int(os.environ.get("PHONIC_FUZZ_SEEDS", "0"))
FUZZ_BASE = int(os.environ.get("PHONIC_FUZZ_BASE", "0"))


def fourcc_str(v):
    return struct.pack(">I", v).decode("latin1")


def random_params(rng, kind, descs):
    params = {}
    for d in descs:
        if rng.random() < 0.5:
            continue
        lo, hi = d["min"], d["max"]
        if d["type"] == 0:  # float
            t = float(rng.random())
            v = lo + (hi - lo) * (t * t if d["scaling"] else t)
            if kind == _capi.FX_DELAY and fourcc_str(d["fourcc"]) in ("lfdt", "lfdf", "ldfb") and rng.random() < 0.7:
                v = 0.0   # mostly leave the delay's LFO depths at zero (the time-parallel path); sometimes modulated (serial path)
            params[fourcc_str(d["fourcc"])] = float(np.float32(v))
        else:  # enum / bool / int: integral raw value
            params[fourcc_str(d["fourcc"])] = float(int(rng.integers(int(lo), int(hi) + 1)))
    return params


_PLAN_CACHE = {}


def outer_rate_of(tone):
    """One voice in five sits behind a ResampledSource: a function of the tone index, so that every seed's other draws stay what they were."""
    return {7: 32000, 3: 96000}.get(tone % 10, 0)


def make_plan(seed):
    """The random graph of a seed — the first of its draws (salt 0, 1, ...) that is AUDIBLE in the oracle: a plan whose every path ends in a gate that
    never opens or a gain of -120 dB compares silence with silence (round 2 skipped such seeds; now every seed of the suite is a real comparison)."""
    import copy

    if seed not in _PLAN_CACHE:
        for salt in range(12):
            plan = _make_plan(seed, salt)
            if float(np.abs(render_plan(copy.deepcopy(plan), oracle.OracleGraph(SR, 2, 1024))).max()) > 1e-3:
                break
        _PLAN_CACHE[seed] = plan
    return copy.deepcopy(_PLAN_CACHE[seed])


def _make_plan(seed, salt=0):
    """One draw: sub-mixer chains with voices, bus chain, ragged block sizes, the block that carries the events."""
    from phonic_amd.graph import effect_parameters

    rng = np.random.default_rng(1000 + seed + 1000003 * salt)
    descs = {k: effect_parameters(k) for k in range(10)}
    plan = {"mixers": [], "bus": [], "seed": seed, "descs": descs}
    for m in range(int(rng.integers(1, 5))):
        chain = []
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(0, 10))
            chain.append((k, random_params(rng, k, descs[k]), int(rng.integers(0, 1000))))
        if rng.random() < 0.5:
            chain.append((_capi.FX_REVERB, random_params(rng, _capi.FX_REVERB, descs[_capi.FX_REVERB]), int(rng.integers(0, 1000))))
        voices = [(int(rng.integers(0, 60)), int(rng.choice([44100, 48000, 32000, 22050, 96000])), float(rng.uniform(0.2, 0.8)), float(rng.uniform(-1, 1)))
                  for _ in range(int(rng.integers(1, 4)))]
        plan["mixers"].append((chain, voices))
    for _ in range(int(rng.integers(0, 3))):
        k = int(rng.integers(0, 10))
        plan["bus"].append((k, random_params(rng, k, descs[k]), int(rng.integers(0, 1000))))
    plan["sizes"] = [int(rng.choice([1024, 1024, 512, 700, 64, 333, 1000])) for _ in range(10)]
    plan["ev_block"] = int(rng.integers(2, 8))
    return plan


def render_plan(plan, g, split=0, events_at_call_start=False, mutations=True):
    """Build the plan's graph on `g` (the HIP graph or the oracle's) and render its calls, events and chain mutations included. split: pull
    every call in pieces of that many frames (diagnostics: a HIP graph walks a long write in the reference's 4096-frame chunks whatever its
    max_frames is, so both sides are pulled in the same calls)."""
    seed, descs, sizes, ev_block = plan["seed"], plan["descs"], plan["sizes"], plan["ev_block"]
    fx_ids, voice_ids, fx_mixer = [], [], {}
    for chain, voices in plan["mixers"]:
        m = g.add_mixer()
        for (k, p, s) in chain:
            fx_ids.append((g.add_effect(m, k, params=p, reverb_seeds=workloads.reverb_seeds(s) if k == _capi.FX_REVERB else None), k))
            fx_mixer[fx_ids[-1][0]] = m
        for (ti, rate, vol, pan) in voices:
            voice_ids.append(g.add_voice(m, workloads.tone_buffer(ti, rate, 0.12), 2, rate, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER,
                                         source_rate=outer_rate_of(ti)))
    for (k, p, s) in plan["bus"]:
        fx_ids.append((g.add_effect(0, k, params=p, reverb_seeds=workloads.reverb_seeds(s) if k == _capi.FX_REVERB else None), k))
        fx_mixer[fx_ids[-1][0]] = 0
    chunks, pos = [], 0
    rng2 = np.random.default_rng(9000 + seed)  # chain mutations (Player::move_effect / remove_effect): the same draws for both sides
    for b, n in enumerate(sizes):
        if b == ev_block:
            if voice_ids:
                g.set_voice_volume(voice_ids[0], 0.3, pos if events_at_call_start else pos + 17)
            if fx_ids:
                fid, k = fx_ids[seed % len(fx_ids)]
                d = descs[k][0]
                if d["type"] == 0:
                    g.schedule_param(fid, fourcc_str(d["fourcc"]), 0.35, pos if events_at_call_start else pos + n // 2, normalized=True)
        if mutations and b in (ev_block + 1, ev_block + 2) and fx_ids and seed % 3 != 0:
            fid, k = fx_ids[int(rng2.integers(0, len(fx_ids)))]
            if fid in fx_mixer:
                if rng2.random() < 0.6:
                    g.move_effect(fid, fx_mixer[fid], _capi.MOVE_DIRECTION, int(rng2.integers(-3, 4)))
                else:
                    g.remove_effect(fid)
                    del fx_mixer[fid]
        o = np.zeros(2 * n, np.float32)
        for off in range(0, n, split or n):
            m = min(split or n, n - off)
            assert g.write(o[2 * off:2 * (off + m)], pos + off) in (0, 2 * m)
        chunks.append(o)
        pos += n
    return np.concatenate(chunks)


def reference_is_discontinuous_here(plan, make_oracle, b, tol_rms, tol_max):
    """The oracle against itself with every source 1e-7 louder. Where that alone moves the output by more than the tolerance the case sits on a
    discontinuity of the reference — so far always the Compressor's gain computer, which has no branch for an envelope exactly on the knee's
    upper edge (tests/test_oracle_graph.py) — and no device-vs-oracle tolerance can hold."""
    import copy

    p2 = copy.deepcopy(plan)
    p2["mixers"] = [(chain, [(ti, rate, float(np.float32(vol * (1.0 + 1e-7))) if np.float32(vol * (1.0 + 1e-7)) != np.float32(vol) else vol * (1.0 + 2e-7), pan)
                             for (ti, rate, vol, pan) in voices]) for chain, voices in plan["mixers"]]
    d = render_plan(p2, make_oracle()).astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d))) > tol_rms or float(np.abs(d).max()) > tol_max


def difference_sits_on_knee_edges(plan, make_oracle, a, b, tol_rms, tol_max):
    """The other half of the same discontinuity: the DEVICE's envelope lands on the knee's upper edge (its upstream arithmetic differs from the
    oracle's in the last place) and the oracle's does not, so perturbing the oracle shows nothing. The oracle logs every frame at which a
    Compressor's envelope comes within 64 ulps of that edge (oracle.knee_edge_frames): when the difference is confined to such frames — without
    them the rest is inside the tolerance — the case is the reference's click, one frame long, placed differently (long-call seed 210084:
    Distortion -> Compressor, ONE frame 25 % apart in 24 729)."""
    import copy

    frames = oracle.knee_edge_frames(lambda: render_plan(copy.deepcopy(plan), make_oracle()))
    if not frames:
        return False
    d = a.astype(np.float64) - b.astype(np.float64)
    n_frames = len(d) // 2
    for f in frames:
        if f < n_frames:
            d[2 * f:2 * f + 2] = 0.0
    return float(np.sqrt(np.mean(d * d))) <= tol_rms and float(np.abs(d).max()) <= tol_max


# 888: a Gate whose envelope crosses the threshold where the device's own log10f and the host's differ in the last bit used to open a frame late;
# 734 (super-block test below): a Compressor whose envelope landed exactly on the upper knee edge — where the reference's gain computer has no
# branch — on the device only. The level detectors now use the host libm's log10f restated (pg_log10f).
FLAT_REGRESSION_SEEDS = [] if FUZZ_SEEDS else [888]


@pytest.mark.parametrize("seed", list(range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS or 48))) + FLAT_REGRESSION_SEEDS)
def test_random_graph_matches_oracle(seed):
    from phonic_amd.graph import Graph

    plan = make_plan(seed)
    sizes = plan["sizes"]
    g = Graph(SR, 2, 1024, 0)
    outs = [render_plan(plan, g), render_plan(plan, oracle.OracleGraph(SR, 2, 1024))]
    a, b = outs
    assert np.isfinite(a).all() and g.device_errors() == 0   # (no kernel met an effect state its time-parallel paths decline)
    assert float(np.abs(b).max()) > 1e-4   # (make_plan draws until the oracle's render is audible)
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    what = {"chains": [[(_capi.FX_NAMES[k], p) for (k, p, _) in chain] for chain, _ in plan["mixers"]], "bus": [(_capi.FX_NAMES[k], p) for (k, p, _) in plan["bus"]],
            "rms_per_block": [float(np.sqrt(np.mean(x * x))) for x in np.array_split(d, len(sizes))], "peak": float(np.abs(b).max())}
    if (float(np.sqrt(np.mean(d * d))) > 1e-5 * scale or float(np.abs(d).max()) > 1e-4 * scale) and (
            reference_is_discontinuous_here(plan, lambda: oracle.OracleGraph(SR, 2, 1024), b, 1e-5 * scale, 1e-4 * scale) or
            difference_sits_on_knee_edges(plan, lambda: oracle.OracleGraph(SR, 2, 1024), a, b, 1e-5 * scale, 1e-4 * scale)):
        pytest.skip("the reference is discontinuous at this input (DESIGN §2 (10))")
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale, f"rms {np.sqrt(np.mean(d * d))} (scale {scale}) {what}"
    assert float(np.abs(d).max()) <= 1e-4 * scale, what


@pytest.mark.parametrize("seed", list(range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 4 or 16))) + ([] if FUZZ_SEEDS else [734]))
def test_random_graph_superblock_writes(seed):
    """The same random graphs pulled in calls of one to four whole blocks with super-block launches enabled (pg_graph_set_max_blocks_per_launch):
    units enter and leave the steady state (events anywhere inside the calls, chain mutations, voices that end, tails, gates), and the host must
    fall back to single launches exactly where it has to — the render equals the one without super-block launches BIT FOR BIT. A call is walked in
    the reference's chunks (<= 4096 frames from the call's start and from every main-mixer event, mixed.rs:679-712) whatever max_frames is, so the
    oracle is pulled in the SAME calls."""
    from phonic_amd.graph import Graph

    rng = np.random.default_rng(11000 + seed)
    sizes = [1024 * int(rng.integers(1, 5)) for _ in range(7)]
    for salt in range(8):   # (make_plan draws until audible in ITS calls; in these the next plans are tried until one is)
        plan = make_plan(seed + 100003 * salt)
        plan["sizes"] = sizes
        b = render_plan(plan, oracle.OracleGraph(SR, 2, 1024))
        if float(np.abs(b).max()) > 1e-3:
            break
    g = Graph(SR, 2, 1024, 0)
    g.set_max_blocks_per_launch(4)
    a = render_plan(plan, g)
    a1 = render_plan(plan, Graph(SR, 2, 1024, 0))           # the same calls, one launch sequence per piece
    assert np.array_equal(a, a1), f"super-block render differs from the piece-by-piece one in {int(np.count_nonzero(a != a1))} samples, sizes {plan['sizes']}"
    assert np.isfinite(a).all() and g.device_errors() == 0
    assert float(np.abs(b).max()) > 1e-4
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    if (float(np.sqrt(np.mean(d * d))) > 1e-5 * scale or float(np.abs(d).max()) > 1e-4 * scale) and reference_is_discontinuous_here(
            plan, lambda: oracle.OracleGraph(SR, 2, 1024), b, 1e-5 * scale, 1e-4 * scale):
        pytest.skip("the reference is discontinuous at this input (DESIGN §2 (10))")
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale, f"rms {np.sqrt(np.mean(d * d))} (scale {scale}) sizes {plan['sizes']}"
    assert float(np.abs(d).max()) <= 1e-4 * scale


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 4 or 12)))
def test_random_graph_on_three_shards(seed):
    """The same random graphs behind ONE pg_sharded_* handle with three shards (on one device here): sub-mixers and main-mixer sources are placed
    on the least loaded shard, events travel to the shard that owns their target, the partial buses meet on the root in shard order in front of
    the bus chain. Against the oracle (the sum over shards reassociates the f32 master-bus sum: tolerance, not bit-equality). Odd seeds: calls of
    one to four blocks rendered as super-blocks, events anywhere inside them, the oracle pulled in the same calls (every shard walks a call in
    the reference's chunks)."""
    from phonic_amd.graph import ShardedGraph

    g = ShardedGraph([0, 0, 0], SR, 2, 1024)
    sizes = None
    if seed % 2:   # odd seeds: whole-block calls of one to four blocks rendered as super-blocks (bus decisions per chunk, flags OR-ed over the shards)
        rng = np.random.default_rng(13000 + seed)
        sizes = [1024 * int(rng.integers(1, 5)) for _ in range(7)]
        g.set_max_blocks_per_launch(4)
    for salt in range(8):   # (the first of the seed's plans that is audible in these calls)
        plan = make_plan(seed + 100003 * salt)
        if sizes:
            plan["sizes"] = sizes
        b = render_plan(plan, oracle.OracleGraph(SR, 2, 1024))
        if float(np.abs(b).max()) > 1e-3:
            break
    a = render_plan(plan, g)
    assert np.isfinite(a).all() and g.device_errors() == 0
    assert float(np.abs(b).max()) > 1e-4
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale, f"rms {np.sqrt(np.mean(d * d))} (scale {scale})"
    assert float(np.abs(d).max()) <= 1e-4 * scale


def make_nested_plan(seed):
    """The first audible draw of a seed's random mixer tree (see make_plan)."""
    import copy

    key = ("nested", seed)
    if key not in _PLAN_CACHE:
        for salt in range(12):
            plan = _make_nested_plan(seed, salt)
            if float(np.abs(render_nested_plan(copy.deepcopy(plan), oracle.OracleGraph(SR, 2, 1024))).max()) > 1e-3:
                break
        _PLAN_CACHE[key] = plan
    return copy.deepcopy(_PLAN_CACHE[key])


def _make_nested_plan(seed, salt=0):
    """Random mixer trees up to depth 4 with events on mixers that have sub-mixers of their own."""
    from phonic_amd.graph import effect_parameters

    rng = np.random.default_rng(5000 + seed + 1000003 * salt)
    descs = {k: effect_parameters(k) for k in range(10)}
    mixers = []  # (parent index into `mixers` or -1 for main, chain, voices)
    for m in range(int(rng.integers(2, 7))):
        parent = -1 if m == 0 or rng.random() < 0.3 else int(rng.integers(0, m))
        chain = []
        for _ in range(int(rng.integers(0, 3))):
            k = int(rng.integers(0, 10))
            chain.append((k, random_params(rng, k, descs[k]), int(rng.integers(0, 1000))))
        if rng.random() < 0.4:
            chain.append((_capi.FX_REVERB, random_params(rng, _capi.FX_REVERB, descs[_capi.FX_REVERB]), int(rng.integers(0, 1000))))
        voices = [(int(rng.integers(0, 60)), int(rng.choice([44100, 48000, 32000])), float(rng.uniform(0.2, 0.8)), float(rng.uniform(-1, 1)))
                  for _ in range(int(rng.integers(0, 3)))]
        mixers.append((parent, chain, voices))
    if not any(v for _, _, v in mixers):
        mixers[-1] = (mixers[-1][0], mixers[-1][1], [(5, 44100, 0.5, 0.0)])
    sizes = [int(rng.choice([1024, 1024, 512, 700, 333])) for _ in range(8)]
    n_events = int(rng.integers(2, 9))
    ev_plan = [(int(rng.integers(1, len(sizes))), float(rng.random()), int(rng.integers(0, 1 << 30)), float(rng.uniform(0.1, 0.9))) for _ in range(n_events)]
    return {"seed": seed, "descs": descs, "mixers": mixers, "sizes": sizes, "ev_plan": ev_plan}


def render_nested_plan(plan, g, mutations=True, events=True):
    seed, descs, mixers, sizes, ev_plan = plan["seed"], plan["descs"], plan["mixers"], plan["sizes"], plan["ev_plan"]
    ids, fx_ids, voice_ids, fx_mixer = [], [], [], {}
    rng2 = np.random.default_rng(7000 + seed)  # chain mutations: the same draws for both sides
    for parent, chain, voices in mixers:
        m = g.add_mixer(None if parent < 0 else ids[parent])
        ids.append(m)
        for (k, p, s) in chain:
            fx_ids.append((g.add_effect(m, k, params=p, reverb_seeds=workloads.reverb_seeds(s) if k == _capi.FX_REVERB else None), k))
            fx_mixer[fx_ids[-1][0]] = m
        for (ti, rate, vol, pan) in voices:
            voice_ids.append(g.add_voice(m, workloads.tone_buffer(ti, rate, 0.12), 2, rate, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER,
                                         source_rate=outer_rate_of(ti)))
    chunks, pos = [], 0
    for b, n in enumerate(sizes):
        for (eb, frac, pick, val) in ev_plan:
            if eb != b or not events:
                continue
            t = pos + int(frac * n)
            if fx_ids and pick % 2 == 0:
                fid, k = fx_ids[(pick >> 1) % len(fx_ids)]
                d = descs[k][0]
                if d["type"] == 0 and fid in fx_mixer:
                    g.schedule_param(fid, fourcc_str(d["fourcc"]), val, t, normalized=True)
                    continue
            g.set_voice_volume(voice_ids[(pick >> 1) % len(voice_ids)], val, t)
        if mutations and b in (3, 5) and fx_ids and seed % 2 == 1:  # Player::move_effect / remove_effect between blocks
            fid, k = fx_ids[int(rng2.integers(0, len(fx_ids)))]
            if fid in fx_mixer:
                if rng2.random() < 0.6:
                    g.move_effect(fid, fx_mixer[fid], _capi.MOVE_DIRECTION, int(rng2.integers(-3, 4)))
                else:
                    g.remove_effect(fid)
                    del fx_mixer[fid]
        o = np.zeros(2 * n, np.float32)
        assert g.write(o, pos) in (0, 2 * n)
        chunks.append(o)
        pos += n
    return np.concatenate(chunks)


# seeds a wide campaign found: 2141 — a parameter event scheduled for an effect that is removed before the event comes due (the event must still
# split the block: the effect processors' tail counters count calls); 3442 — a room that shrinks leaves a ring position above the new ring end,
# and the generic kernel must not hand that block to the time-parallel reverb; 1853 — a threshold update of a compressor used to make the device
# recompute the envelope follower's exp(-1 / (t fs)) coefficients with its own expf (one ulp off the host's = 0.5 % of a 2 s time constant)
NESTED_REGRESSION_SEEDS = [] if FUZZ_SEEDS else [2141, 3442, 1853]


@pytest.mark.parametrize("seed", list(range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 2 or 24))) + NESTED_REGRESSION_SEEDS)
def test_random_nested_graph_matches_oracle(seed):
    """Player::add_mixer(parent): random mixer trees up to depth 4. Events on a mixer with sub-mixers (effect parameters, voice
    volume) split its block, and with it the write() calls its sub-mixers see (per-call silence gate and bypass logic)."""
    from phonic_amd.graph import Graph

    plan = make_nested_plan(seed)
    g = Graph(SR, 2, 1024, 0)
    a = render_nested_plan(plan, g)
    b = render_nested_plan(plan, oracle.OracleGraph(SR, 2, 1024))
    assert np.isfinite(a).all() and g.device_errors() == 0
    assert float(np.abs(b).max()) > 1e-4   # (make_nested_plan draws until the oracle's render is audible)
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    what = {"mixers": [(parent, [(_capi.FX_NAMES[k], p) for (k, p, _) in chain], len(voices)) for parent, chain, voices in plan["mixers"]],
            "rms_per_block": [float(np.sqrt(np.mean(x * x))) for x in np.array_split(d, len(plan["sizes"]))], "peak": float(np.abs(b).max())}
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale, f"rms {np.sqrt(np.mean(d * d))} (scale {scale}) {what}"
    assert float(np.abs(d).max()) <= 1e-4 * scale, what


def make_voice_plan(seed):
    """Random file sources on the main mixer and on one plain sub-mixer: rates on both sides of the mixer's (resampler ratios from 0.17 to 7; some behind a ResampledSource at a third rate),
    mono / stereo, one-shots, finite and endless repeats, loop ranges, start times inside blocks, fade-out lengths, and a schedule of stop /
    volume / panning / speed (immediate and glide) / seek calls at sample times."""
    rng = np.random.default_rng(21000 + seed)
    rng2 = np.random.default_rng(22000 + seed)
    voices = []
    for _ in range(int(rng.integers(1, 6))):
        rate = int(rng.choice([8000, 11025, 22050, 32000, 44100, 48000, 64000, 96000]))
        nch = int(rng.choice([1, 2]))
        seconds = float(rng.uniform(0.02, 0.25))
        frames = int(rate * seconds)
        opt = dict(volume=float(rng.uniform(0.1, 0.9)), panning=float(rng.uniform(-1, 1)), start_time=int(rng.choice([0, 0, int(rng.integers(1, 3000))])),
                   fade_out_seconds=float(rng.choice([0.05, 0.0, 0.01, 0.2])))
        mode = int(rng.integers(0, 4))
        if mode == 1:
            opt.update(has_repeat=1, repeat=int(rng.integers(1, 4)))
        elif mode == 2:
            opt.update(has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
        elif mode == 3 and frames > 64:
            a = int(rng.integers(0, frames // 2))
            opt.update(has_repeat=1, repeat=(2 if rng.random() < 0.5 else _capi.PG_REPEAT_FOREVER), has_loop_range=1, loop_start=a, loop_end=int(rng.integers(a + 16, frames + 1)))
        if rng.random() < 0.3:
            opt["speed"] = float(rng.choice([0.5, 0.75, 1.25, 1.5, 2.0]))
        # (a generator of its own: the draws above stay what they were for every seed) a ResampledSource behind the file source, speeds up to
        # two octaves above the file's pitch (the resampler schedule's [2, 4) range)
        if rng2.random() < 0.25:
            opt["source_rate"] = int(rng2.choice([22050, 32000, 44100, 96000]))
        if rng2.random() < 0.15:
            opt["speed"] = float(rng2.choice([2.5, 3.0, 3.5]))
        if rng2.random() < 0.2:
            opt["fade_in_seconds"] = float(rng2.choice([0.005, 0.02, 0.1]))
        voices.append(dict(mixer=int(rng.integers(0, 2)), tone=(int(rng.integers(0, 60)), rate, seconds, nch), opt=opt))
    sizes = [int(rng.choice([1024, 1024, 512, 700, 64, 333, 1000, 1])) for _ in range(12)]
    total = sum(sizes)
    actions = []
    for _ in range(int(rng.integers(0, 9))):
        kind = str(rng.choice(["stop", "volume", "panning", "speed", "glide", "seek"]))
        actions.append((int(rng.integers(0, len(sizes))), kind, int(rng.integers(0, len(voices))), float(rng.random()), int(rng.integers(0, total))))
    return {"voices": voices, "sizes": sizes, "actions": actions}


def render_voice_plan(plan, g):
    m1 = g.add_mixer()
    ids = []
    for v in plan["voices"]:
        i, rate, seconds, nch = v["tone"]
        ids.append(g.add_voice(m1 if v["mixer"] else 0, workloads.tone_buffer(i, rate, seconds, channels=nch), nch, rate, **v["opt"]))
    chunks, pos = [], 0
    for b, n in enumerate(plan["sizes"]):
        for (ab, kind, vi, x, t) in plan["actions"]:
            if ab != b:
                continue
            t = max(t, pos)   # handles schedule from "now" on
            if kind == "stop":
                g.stop_voice(ids[vi], t)
            elif kind == "volume":
                g.set_voice_volume(ids[vi], x, t)
            elif kind == "panning":
                g.set_voice_panning(ids[vi], 2.0 * x - 1.0, t)
            elif kind == "speed":
                g.set_voice_speed(ids[vi], 0.5 + 1.5 * x, t)
            elif kind == "glide":
                g.set_voice_speed(ids[vi], 0.5 + 1.5 * x, t, glide=6.0 + 60.0 * x)
            else:
                # Seek positions on a frame boundary: PreloadedFileSource::seek turns seconds into a SAMPLE index (seconds * rate * channels,
                # truncated, preloaded.rs:137-145); an odd index in a stereo file leaves half a frame in front of the loop end that no
                # resampler call can consume — write_buffer then spins forever in the reference (and in the oracle; the device breaks out).
                rate = plan["voices"][vi]["tone"][1]
                g.seek_voice(ids[vi], (int(0.2 * x * rate) + 0.25) / rate, t)
        o = np.zeros(2 * n, np.float32)
        assert g.write(o, pos) in (0, 2 * n)
        chunks.append(o)
        pos += n
    return np.concatenate(chunks)


# 9028: two panning events for one source come due in front of the same chunk — the reference's one-slot message queue keeps the last one only,
# and the first one would have snapped the nearly settled smoother onto its target
# 743, 2600: a ResampledSource-backed mono voice that runs out exactly at an event inside the block (the mixer keeps calling an inactive source until
# its write ends); 2452: the same inside a sub-mixer whose call ends at a main-mixer event; 2008, 1439: fade-outs at source rates other than 48 kHz
# (expf of the fader's inertia, one ulp apart between two libms)
@pytest.mark.parametrize("seed", list(range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 2 or 24))) + ([] if FUZZ_SEEDS else [9028, 743, 2600, 2452, 2008, 1439]))
def test_random_voice_features_match_oracle(seed):
    """PreloadedFileSource / FileSourceImpl / VolumeFader / ChannelMapped / Amplified / Panned through MixedSource's source loop, no effects:
    the arithmetic is f32 and identical on both sides (resampler schedule, Hermite taps, fades, smoothed gain and panning), only the order of
    the f32 sum over sources differs — 1e-6 RMS."""
    from phonic_amd.graph import Graph

    plan = make_voice_plan(seed)
    a = render_voice_plan(plan, Graph(SR, 2, 1024, 0))
    b = render_voice_plan(plan, oracle.OracleGraph(SR, 2, 1024))
    assert np.isfinite(a).all()
    d = a.astype(np.float64) - b.astype(np.float64)
    what = {"voices": [(v["mixer"], v["tone"][1:], v["opt"]) for v in plan["voices"]], "actions": plan["actions"], "sizes": plan["sizes"],
            "rms_per_block": [float(np.sqrt(np.mean(x * x))) if x.size else 0.0 for x in np.split(d, np.cumsum([2 * n for n in plan["sizes"]])[:-1])]}
    assert float(np.sqrt(np.mean(d * d))) <= 1e-6 and float(np.abs(d).max()) <= 1e-5, what


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS or 40)))
def test_random_standalone_effect_sequences(seed):
    """The `trait Effect` surface (pg_effect_*): one effect of a random kind with random construction parameters, then a random sequence of
    process calls of ragged sizes with parameter updates (raw and normalized, any parameter, several between two calls) and reset messages in
    between, on a signal that alternates noise, bursts and silence."""
    import phonic_amd
    from phonic_amd.graph import effect_parameters

    rng = np.random.default_rng(31000 + seed)
    kind = int(rng.integers(0, 10))
    descs = effect_parameters(kind)
    params = random_params(rng, kind, descs) if rng.random() < 0.7 else None
    seeds = workloads.reverb_seeds(int(rng.integers(0, 1000))) if kind == _capi.FX_REVERB else None
    e_gpu, e_cpu = phonic_amd.Effect(kind, params, seeds), oracle.OracleEffect(kind, params, seeds)
    # (a generator of its own: the other draws stay what they were) the kinds that take any channel count, at 1, 3, 4 or 6 channels half of the time
    rng3 = np.random.default_rng(32000 + seed)
    C = 2
    if kind in (_capi.FX_FILTER, _capi.FX_EQ5, _capi.FX_GAIN, _capi.FX_DISTORTION) and rng3.random() < 0.5:
        C = int(rng3.choice([1, 3, 4, 6]))
    for e in (e_gpu, e_cpu):
        e.initialize(SR, C, 1024)
    sizes = [int(rng.choice([1024, 1024, 512, 700, 64, 333, 1000, 1, 0])) for _ in range(14)]
    x = workloads.test_signal((C * sum(sizes) + 1) // 2 + 1, seed=seed, kind=str(rng.choice(["noise", "burst"])))[: C * sum(sizes)].copy()
    quiet = int(rng.integers(0, len(sizes)))
    a, b = x.copy(), x.copy()
    pos = 0
    log = []
    for i, n in enumerate(sizes):
        if i == quiet:
            a[C * pos:] *= 0.0
            b[C * pos:] *= 0.0
        for _ in range(int(rng.choice([0, 0, 1, 1, 2, 4]))):
            d = descs[int(rng.integers(0, len(descs)))]
            name = fourcc_str(d["fourcc"])
            norm = bool(rng.random() < 0.5)
            if d["type"] == 0:
                v = float(rng.random()) if norm else float(np.float32(d["min"] + (d["max"] - d["min"]) * rng.random()))
            else:
                v = float(rng.random()) if norm else float(int(rng.integers(int(d["min"]), int(d["max"]) + 1)))
            log.append((i, name, v, norm))
            e_gpu.set_parameter(name, v, norm)
            e_cpu.set_parameter(name, v, norm)
        if kind in (_capi.FX_DELAY, _capi.FX_REVERB, _capi.FX_CHORUS) and rng.random() < 0.1:
            log.append((i, "reset"))
            e_gpu.reset()
            e_cpu.reset()
        e_gpu.process(a[C * pos:C * (pos + n)])
        e_cpu.process(b[C * pos:C * (pos + n)])
        pos += n
    assert np.isfinite(a).all()
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    what = {"kind": _capi.FX_NAMES[kind], "channels": C, "params": params, "sizes": sizes, "quiet from call": quiet, "log": log}
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale and float(np.abs(d).max()) <= 1e-4 * scale, (float(np.sqrt(np.mean(d * d))), float(np.abs(d).max()), what)


@pytest.mark.parametrize("seed", list(range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 4 or 16))) + ([] if FUZZ_SEEDS else [301229]))
def test_random_graph_other_rates_and_block_sizes(seed):
    """The flat random graphs at other mixer rates (22.05 / 44.1 / 96 kHz: delay-line lengths, filter coefficients, smoother and tail constants,
    resampler ratios on the other side of one all move) and other max_frames (256 ... 4096: chunking of the time-parallel paths, the staged
    kernels' 1024-frame limit, LDS plans)."""
    from phonic_amd.graph import Graph

    rng = np.random.default_rng(41000 + seed)
    sr = int(rng.choice([22050, 44100, 96000]))
    mf = int(rng.choice([256, 512, 2048, 4096]))
    sizes = [int(rng.choice([mf, mf, mf // 2, max(1, mf // 3), 64, 1])) for _ in range(9)]
    for salt in range(8):   # (make_plan draws until audible at 48 kHz in 1024-frame blocks; at this rate and these sizes the next plans are tried until one is)
        plan = make_plan(seed + 100003 * salt)
        plan["sizes"] = sizes
        b = render_plan(plan, oracle.OracleGraph(sr, 2, mf))
        if float(np.abs(b).max()) > 1e-3:
            break
    g = Graph(sr, 2, mf, 0)
    a = render_plan(plan, g)
    assert np.isfinite(a).all() and g.device_errors() == 0
    assert float(np.abs(b).max()) > 1e-4
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    what = {"sr": sr, "max_frames": mf, "sizes": plan["sizes"], "chains": [[_capi.FX_NAMES[k] for (k, _, _) in chain] for chain, _ in plan["mixers"]], "bus": [_capi.FX_NAMES[k] for (k, _, _) in plan["bus"]]}
    if (float(np.sqrt(np.mean(d * d))) > 1e-5 * scale or float(np.abs(d).max()) > 1e-4 * scale) and reference_is_discontinuous_here(
            plan, lambda: oracle.OracleGraph(sr, 2, mf), b, 1e-5 * scale, 1e-4 * scale):
        pytest.skip("the reference is discontinuous at this input (seed 301229: DESIGN §2 (10))")
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale and float(np.abs(d).max()) <= 1e-4 * scale, (float(np.sqrt(np.mean(d * d))), what)


def make_topology_plan(seed):
    """A graph that keeps changing while it plays: between blocks, sub-mixers (also nested ones) are added and removed with everything on them,
    effects are added, moved and removed, one-shot and looping voices start on any live mixer, single voices and all voices are stopped, and
    parameter / volume events are scheduled — also for things that are gone by the time the event comes due."""
    from phonic_amd.graph import effect_parameters

    rng = np.random.default_rng(51000 + seed)
    descs = {k: effect_parameters(k) for k in range(10)}
    steps = []
    for b in range(12):
        acts = []
        for _ in range(int(rng.choice([0, 1, 1, 2, 3]))):
            what = str(rng.choice(["add_mixer", "add_nested", "remove_mixer", "add_effect", "remove_effect", "move_effect", "add_voice", "add_voice", "stop_voice", "stop_all",
                                   "param", "volume"]))
            k = int(rng.integers(0, 10))
            acts.append(dict(what=what, pick=int(rng.integers(0, 1 << 30)), kind=k, params=random_params(rng, k, descs[k]), rseed=int(rng.integers(0, 1000)),
                             tone=int(rng.integers(0, 60)), rate=int(rng.choice([44100, 48000, 32000, 22050])), vol=float(rng.uniform(0.2, 0.7)), pan=float(rng.uniform(-1, 1)),
                             loop=bool(rng.random() < 0.5), frac=float(rng.random()), val=float(rng.uniform(0.1, 0.9)), off=int(rng.integers(-2, 3))))
        steps.append((int(rng.choice([1024, 1024, 512, 700, 333])), acts))
    return {"steps": steps, "descs": descs}


def render_topology_plan(plan, g):
    descs = plan["descs"]
    mixers, parent_of, fx, voices = [0], {0: None}, [], []      # live mixers (0 = main), effects [(id, kind, mixer)], voices [(id, mixer)]

    def dead_branch(m):
        out, grew = {m}, True
        while grew:
            grew = False
            for c, p in parent_of.items():
                if p in out and c not in out:
                    out.add(c)
                    grew = True
        return out

    chunks, pos = [], 0
    for n, acts in plan["steps"]:
        for a in acts:
            w, pick = a["what"], a["pick"]
            t = pos + int(a["frac"] * n)
            if w == "add_mixer" and len(mixers) < 7:
                m = g.add_mixer()
                mixers.append(m); parent_of[m] = 0
            elif w == "add_nested" and len(mixers) > 1 and len(mixers) < 7:
                p = mixers[1 + pick % (len(mixers) - 1)]
                m = g.add_mixer(p)
                mixers.append(m); parent_of[m] = p
            elif w == "remove_mixer" and len(mixers) > 1:
                m = mixers[1 + pick % (len(mixers) - 1)]
                gone = dead_branch(m)
                g.remove_mixer(m)
                mixers[:] = [x for x in mixers if x not in gone]
                fx[:] = [e for e in fx if e[2] not in gone]
                voices[:] = [v for v in voices if v[1] not in gone]
                for x in gone:
                    parent_of.pop(x, None)
            elif w == "add_effect":
                m = mixers[pick % len(mixers)]
                if sum(1 for e in fx if e[2] == m) < 3:
                    fx.append((g.add_effect(m, a["kind"], params=a["params"], reverb_seeds=workloads.reverb_seeds(a["rseed"]) if a["kind"] == _capi.FX_REVERB else None), a["kind"], m))
            elif w == "remove_effect" and fx:
                e = fx.pop(pick % len(fx))
                g.remove_effect(e[0])
            elif w == "move_effect" and fx:
                e = fx[pick % len(fx)]
                g.move_effect(e[0], e[2], _capi.MOVE_DIRECTION, a["off"])
            elif w == "add_voice" and len(voices) < 10:
                m = mixers[pick % len(mixers)]
                opt = dict(volume=a["vol"], panning=a["pan"], start_time=t, source_rate=outer_rate_of(a["tone"]))
                if a["loop"]:
                    opt.update(has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
                v = g.add_voice(m, workloads.tone_buffer(a["tone"], a["rate"], 0.1), 2, a["rate"], **opt)
                if a["loop"]:   # only voices that cannot end by themselves are addressed later: a handle of a source that has ended refuses calls
                    voices.append((v, m))   # (FilePlaybackHandle: Error::SourceNotPlaying), and when it notices is a matter of thread timing in the reference
            elif w == "stop_voice" and voices:
                g.stop_voice(voices.pop(pick % len(voices))[0], t)
            elif w == "stop_all":
                g.stop_all_voices()
                voices.clear()
            elif w == "param" and fx:
                e = fx[pick % len(fx)]
                d = descs[e[1]][(pick >> 8) % len(descs[e[1]])]
                if d["type"] == 0 and not (e[1] == _capi.FX_DELAY and fourcc_str(d["fourcc"]) in ("lfdt", "lfdf", "ldfb")):
                    g.schedule_param(e[0], fourcc_str(d["fourcc"]), a["val"], t + (n if pick & 1 else 0), normalized=True)   # some come due a block later
            elif w == "volume" and voices:
                g.set_voice_volume(voices[pick % len(voices)][0], a["val"], t)
        o = np.zeros(2 * n, np.float32)
        assert g.write(o, pos) in (0, 2 * n)
        chunks.append(o)
        pos += n
    return np.concatenate(chunks)


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 2 or 32)))
def test_random_topology_changes_while_playing(seed):
    from phonic_amd.graph import Graph

    plan = make_topology_plan(seed)
    g = Graph(SR, 2, 1024, 0)
    a = render_topology_plan(plan, g)
    b = render_topology_plan(plan, oracle.OracleGraph(SR, 2, 1024))
    assert np.isfinite(a).all() and g.device_errors() == 0
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    edges = np.cumsum([0] + [2 * n for n, _ in plan["steps"]])
    what = {"rms_per_block": [float(np.sqrt(np.mean(d[edges[i]:edges[i + 1]] ** 2))) for i in range(len(plan["steps"]))],
            "steps": [(n, [(x["what"], x["pick"] % 97, _capi.FX_NAMES[x["kind"]]) for x in acts]) for n, acts in plan["steps"]]}
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale and float(np.abs(d).max()) <= 1e-4 * scale, (float(np.sqrt(np.mean(d * d))), what)


class _AsyncWriter:
    """A Graph whose write() enqueues the render on a caller's stream into one device buffer and returns at once (pg_graph_write_device): the
    host never waits between blocks, so every graph mutation and control call of a plan meets rounds that are still in flight."""

    def __init__(self, g, total_frames):
        import torch

        self._g, self._torch = g, torch
        self._stream = torch.cuda.Stream()
        self._buf = torch.zeros(2 * total_frames, dtype=torch.float32, device="cuda")
        self._views = []

    def __getattr__(self, name):
        return getattr(self._g, name)

    def write(self, out, pos):
        off = 2 * pos
        w = self._g.write_device(self._buf.data_ptr() + 4 * off, out.size, pos, self._stream.cuda_stream)
        self._views.append((out, off, w))
        return w if w else out.size   # (0 = nothing to do: the slice stays zero, as a host write leaves its buffer)

    def finish(self):
        self._torch.cuda.synchronize()
        host = self._buf.cpu().numpy()
        for out, off, w in self._views:
            if w:
                out[:] = host[off:off + out.size]


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 2 or 24)))
def test_random_topology_changes_on_an_asynchronous_stream(seed):
    """The same changing graphs rendered without a single host wait between blocks: pg_graph_write_device on a caller's stream returns while the
    round is in flight, and the add / remove / move / stop calls that follow must order themselves against it (they drain the stream the last
    write used before they touch tables a running round reads)."""
    from phonic_amd.graph import Graph

    plan = make_topology_plan(seed)
    g = Graph(SR, 2, 1024, 0)
    w = _AsyncWriter(g, sum(n for n, _ in plan["steps"]))
    render_topology_plan(plan, w)
    w.finish()
    a = np.concatenate([out for out, _, _ in w._views])   # (the blocks' host arrays are filled when the stream has drained)
    b = render_topology_plan(plan, oracle.OracleGraph(SR, 2, 1024))
    assert np.isfinite(a).all() and g.device_errors() == 0
    d = a.astype(np.float64) - b.astype(np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale and float(np.abs(d).max()) <= 1e-4 * scale, float(np.sqrt(np.mean(d * d)))


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 4 or 12)))
def test_random_graph_on_the_exact_serial_kernels(seed):
    """pg_graph_set_fast_math(0): every unit on the generic kernel's exact serial code (the path that otherwise only renders command blocks and the
    few states without a time-parallel form) — flat and changing graphs alike, same bar."""
    from phonic_amd.graph import Graph

    for make, render in ((make_plan, render_plan), (make_topology_plan, render_topology_plan)):
        plan = make(seed)
        g = Graph(SR, 2, 1024, 0)
        g.set_fast_math(0)
        a = render(plan, g)
        b = render(plan, oracle.OracleGraph(SR, 2, 1024))
        assert np.isfinite(a).all() and g.device_errors() == 0
        d = a.astype(np.float64) - b.astype(np.float64)
        scale = max(1.0, float(np.abs(b).max()))
        assert float(np.sqrt(np.mean(d * d))) <= 1e-5 * scale and float(np.abs(d).max()) <= 1e-4 * scale, (make.__name__, float(np.sqrt(np.mean(d * d))))


# Call lengths of the long-call family: below, at and above max_frames and the reference's 4096-frame chunk, not multiples of either
LONG_CALLS = [1024, 2048, 3072, 4096, 5000, 700, 2500, 8192, 6144, 333, 4097, 1, 9000, 1500]


def _long_call_render(family, seed):
    """(render(graph) -> samples, call sizes, tolerance (rms, max-abs) relative to the oracle's peak, the flat family's plan for the discontinuity
    classifier | None) of a family's plan pulled in long calls."""
    import copy

    rng = np.random.default_rng(77000 + seed)
    if family == "flat":
        plan = make_plan(seed)
        plan["sizes"] = [int(rng.choice(LONG_CALLS)) for _ in range(9)]
        return (lambda g: render_plan(copy.deepcopy(plan), g)), plan["sizes"], (1e-5, 1e-4), plan
    if family == "nested":
        plan = make_nested_plan(seed)
        plan["sizes"] = [int(rng.choice(LONG_CALLS)) for _ in range(len(plan["sizes"]))]
        return (lambda g: render_nested_plan(copy.deepcopy(plan), g)), plan["sizes"], (1e-5, 1e-4), None
    if family == "voices":
        plan = make_voice_plan(seed)
        old_total = sum(plan["sizes"])
        plan["sizes"] = [int(rng.choice(LONG_CALLS)) for _ in range(len(plan["sizes"]))]
        stretch = sum(plan["sizes"]) / max(1, old_total)
        plan["actions"] = [(ab, kind, vi, x, int(t * stretch)) for (ab, kind, vi, x, t) in plan["actions"]]
        return (lambda g: render_voice_plan(copy.deepcopy(plan), g)), plan["sizes"], (1e-6, 1e-5), None
    plan = make_topology_plan(seed)
    plan["steps"] = [(int(rng.choice(LONG_CALLS)), acts) for (_, acts) in plan["steps"]]
    return (lambda g: render_topology_plan(copy.deepcopy(plan), g)), [n for n, _ in plan["steps"]], (1e-5, 1e-4), None


# what the first campaigns found (DESIGN §2 "Round 4: the chunk grid"): the fader's arrival test (voices 2, topology 3), is_exhausted of a ResampledSource
# at a piece boundary (voices 37), the file's last frame (voices 46, 70), the per-call ramp branch of Eq5 / Filter (nested 1210)
LONG_CALL_CASES = [(f, s) for f in ("flat", "nested", "voices", "topology") for s in range(FUZZ_BASE, FUZZ_BASE + (FUZZ_SEEDS // 4 or 10))] + (
    [] if FUZZ_SEEDS else [("voices", 2), ("topology", 3), ("voices", 37), ("voices", 46), ("voices", 70), ("nested", 1210)])


@pytest.mark.parametrize("family,seed", LONG_CALL_CASES)
def test_random_graphs_in_long_calls(family, seed):
    """MixedSource::write walks a call in chunks of min(remaining, 4096) frames from the call's start and from every event (mixed.rs:216,679-712);
    sources, effect processors and sub-mixers are called once per chunk and take their per-call decisions there (bypass and tail counters,
    silence gates, the chorus' call-end phase bookkeeping, the fader's arrival test, is_exhausted). The graph keeps that grid whatever its
    max_frames is: a chunk is rendered as pieces of at most max_frames frames. Calls of 1 to 9000 frames with events anywhere, the four plan
    families:  (1) on the exact serial kernels max_frames 1024 / 256 / 1000 equal max_frames 4096 — one piece per chunk — BIT FOR BIT;
    (2) on the time-parallel kernels super-block launches equal single launches bit for bit;  (3) every configuration agrees with the oracle
    pulled in the SAME calls."""
    from phonic_amd.graph import Graph

    render, sizes, (tol_rms, tol_max), flat_plan = _long_call_render(family, seed)
    ref = render(oracle.OracleGraph(SR, 2, 1024))
    scale = 1.0 if family == "voices" else max(1.0, float(np.abs(ref).max()))
    outs = {}
    for key, mf, fast, blocks in (("s4096", 4096, 0, 1), ("s1024", 1024, 0, 1), ("s256", 256, 0, 1), ("s1000", 1000, 0, 1),
                                  ("f1024", 1024, 1, 1), ("f1024x8", 1024, 1, 8), ("f512x16", 512, 1, 16)):
        g = Graph(SR, 2, mf, 0)
        if not fast:
            g.set_fast_math(0)
        if blocks > 1:
            g.set_max_blocks_per_launch(blocks)
        outs[key] = render(g)
        assert np.isfinite(outs[key]).all() and g.device_errors() == 0, key
        d = outs[key].astype(np.float64) - ref.astype(np.float64)
        if flat_plan is not None and (float(np.sqrt(np.mean(d * d))) > tol_rms * scale or float(np.abs(d).max()) > tol_max * scale) and (
                reference_is_discontinuous_here(flat_plan, lambda: oracle.OracleGraph(SR, 2, 1024), ref, tol_rms * scale, tol_max * scale) or
                difference_sits_on_knee_edges(flat_plan, lambda: oracle.OracleGraph(SR, 2, 1024), outs[key], ref, tol_rms * scale, tol_max * scale)):
            pytest.skip("the reference is discontinuous at this input (DESIGN §2 (10))")   # (seed 210084: a Compressor's envelope on its knee's upper edge on the device only)
        assert float(np.sqrt(np.mean(d * d))) <= tol_rms * scale and float(np.abs(d).max()) <= tol_max * scale, (key, sizes, float(np.sqrt(np.mean(d * d))), float(np.abs(d).max()))
    for key in ("s1024", "s256", "s1000"):
        assert np.array_equal(outs[key], outs["s4096"]), (key, sizes, int(np.flatnonzero(outs[key] != outs["s4096"])[0]) // 2)
    assert np.array_equal(outs["f1024x8"], outs["f1024"]), (sizes, int(np.flatnonzero(outs["f1024x8"] != outs["f1024"])[0]) // 2)
