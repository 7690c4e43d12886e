"""Parity of the ten stock effects on the GPU (through the C ABI, include/phonic_gpu.h) against the CPU oracle,
on the same seeded inputs: default parameters, `with_parameters` constructions and ramped-parameter cases
(per-frame coefficient branches). Tolerance: <= 1e-5 RMS and <= 1e-4 max-abs (signals are O(0.1..1)); integer
state (counters, indices) is compared for equality where the ABI exposes it."""
import numpy as np
import pytest

import oracle
import workloads
from phonic_amd import _capi

pytestmark = pytest.mark.gpu

SR = 48000
RMS_TOL = 1e-5
MAX_TOL = 1e-4


def gpu_effect(kind, params=None, seeds=None):
    import phonic_amd

    return phonic_amd.Effect(kind, params, seeds)


def run_pair(kind, params=None, seeds=None, blocks=6, frames=512, signal="noise", updates=None, sr=SR):
    """Process `blocks` blocks through both implementations; `updates` = {block_index: [(fourcc, value, normalized)]}."""
    e_gpu = gpu_effect(kind, params, seeds)
    e_cpu = oracle.OracleEffect(kind, params, seeds)
    for e in (e_gpu, e_cpu):
        e.initialize(sr, 2, 4096)
    x = workloads.test_signal(blocks * frames, seed=kind + 11, kind=signal)
    a, b = x.copy(), x.copy()
    for blk in range(blocks):
        for (id4, val, norm) in (updates or {}).get(blk, []):
            e_gpu.set_parameter(id4, val, norm)
            e_cpu.set_parameter(id4, val, norm)
        sl = slice(blk * frames * 2, (blk + 1) * frames * 2)
        e_gpu.process(a[sl])
        e_cpu.process(b[sl])
    return a, b, e_gpu, e_cpu


def check(a, b, rms_tol=RMS_TOL, max_tol=MAX_TOL):
    assert np.isfinite(a).all()
    d = a.astype(np.float64) - b.astype(np.float64)
    rms = float(np.sqrt(np.mean(d * d)))
    mx = float(np.abs(d).max())
    assert rms <= rms_tol, f"rms {rms}"
    assert mx <= max_tol, f"max {mx}"
    return rms, mx


CASES = [
    ("gain_default", _capi.FX_GAIN, None, None),
    ("gain_dc", _capi.FX_GAIN, {"gain": 0.5, "dcfm": 2}, None),
    ("pan_default", _capi.FX_PANNING, None, None),
    ("pan_params", _capi.FX_PANNING, {"pan ": -0.3, "wdth": 1.5, "invr": 1}, None),
    ("filter_default", _capi.FX_FILTER, None, None),
    ("filter_lp", _capi.FX_FILTER, {"type": 0, "cuto": 2000.0, "fltq": 0.707}, None),
    ("filter_hp", _capi.FX_FILTER, {"type": 3, "cuto": 500.0, "fltq": 2.0}, None),
    ("filter_bp", _capi.FX_FILTER, {"type": 1, "cuto": 1200.0, "fltq": 1.0}, None),
    ("filter_notch", _capi.FX_FILTER, {"type": 2, "cuto": 3000.0, "fltq": 0.5}, None),
    ("eq5_default", _capi.FX_EQ5, None, None),
    ("eq5_gains", _capi.FX_EQ5, {"gan1": 6.0, "gan2": -3.0, "gan3": 4.0, "gan4": -6.0, "gan5": 2.0, "bw_2": 1.5}, None),
    ("delay_default", _capi.FX_DELAY, None, None),
    ("delay_pingpong", _capi.FX_DELAY, {"mode": 1, "dlay": 20.0, "fdbk": 0.7, "driv": 0.5, "ftyp": 2, "wdth": 1.0}, None),
    ("delay_lfo", _capi.FX_DELAY, {"dlay": 10.0, "lfdt": 0.1, "ldfb": 0.3, "lfdf": 0.5, "lfor": 5.0, "lfos": 1}, None),
    ("reverb_default", _capi.FX_REVERB, None, workloads.reverb_seeds(3)),
    ("reverb_small", _capi.FX_REVERB, {"room": 0.0, "wet ": 1.0}, workloads.reverb_seeds(4)),
    ("reverb_big", _capi.FX_REVERB, {"room": 1.0, "wet ": 0.5}, workloads.reverb_seeds(5)),
    ("chorus_default", _capi.FX_CHORUS, None, None),
    ("chorus_params", _capi.FX_CHORUS, {"rate": 3.0, "dpth": 0.8, "fdbk": -0.6, "dlay": 0.5, "fltt": 1, "fltf": 300.0, "fltq": 0.4}, None),
    ("comp_default", _capi.FX_COMPRESSOR, None, None),
    ("limiter", _capi.FX_COMPRESSOR, {"thrs": -0.01, "rato": 20.0, "knee": 0.0, "attk": 0.02, "rels": 2.0, "gain": 0.0, "look": 0.02}, None),
    ("gate_default", _capi.FX_GATE, None, None),
    ("gate_params", _capi.FX_GATE, {"thrs": -20.0, "attk": 0.002, "hold": 0.01, "rels": 0.05, "rnge": -40.0}, None),
    ("dist_default", _capi.FX_DISTORTION, None, None),
    ("dist_soft", _capi.FX_DISTORTION, {"type": 0, "driv": 2.0}, None),
    ("dist_hard", _capi.FX_DISTORTION, {"type": 1, "driv": 3.0, "mix ": 0.5}, None),
    ("dist_diode", _capi.FX_DISTORTION, {"type": 2, "driv": 1.0}, None),
    ("dist_fuzz", _capi.FX_DISTORTION, {"type": 3, "driv": 2.5}, None),
    ("dist_fold", _capi.FX_DISTORTION, {"type": 4, "driv": 4.0}, None),
]


@pytest.mark.parametrize("name,kind,params,seeds", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("signal", ["noise", "burst"])
def test_effect_parity(name, kind, params, seeds, signal):
    a, b, _, _ = run_pair(kind, params, seeds, blocks=6, frames=512, signal=signal)
    check(a, b)
    if name not in ("gain_default", "pan_default", "dist_default", "filter_default", "eq5_default"):
        x = workloads.test_signal(6 * 512, seed=kind + 11, kind=signal)
        assert not np.array_equal(a, x), "effect left the signal untouched"


RAMPS = [
    ("gain", _capi.FX_GAIN, None, None, {1: [("gain", 0.25, False)], 3: [("gain", 0.9, True)]}),
    ("pan", _capi.FX_PANNING, None, None, {1: [("pan ", 0.8, False), ("wdth", 0.2, False)], 3: [("invl", 1.0, False)]}),
    ("filter", _capi.FX_FILTER, None, None, {1: [("cuto", 800.0, False)], 2: [("fltq", 3.0, False), ("type", 3, False)], 4: [("cuto", 0.9, True)]}),
    ("eq5", _capi.FX_EQ5, None, None, {1: [("gan2", 9.0, False), ("frq2", 500.0, False)], 3: [("bw_3", 0.7, False), ("gan5", -12.0, False)]}),
    ("delay", _capi.FX_DELAY, {"dlay": 30.0}, None, {1: [("dlay", 60.0, False), ("fdbk", 0.8, False)], 3: [("cuto", 1000.0, False), ("lfor", 4.0, False), ("lfdt", 0.2, False)]}),
    ("reverb", _capi.FX_REVERB, None, workloads.reverb_seeds(9), {1: [("room", 0.9, False)], 3: [("wet ", 0.8, False)]}),
    ("chorus", _capi.FX_CHORUS, None, None, {1: [("rate", 4.0, False), ("phas", 1.0, False)], 2: [("dlay", 30.0, False), ("fltf", 2000.0, False)], 4: [("dpth", 0.9, False)]}),
    ("comp", _capi.FX_COMPRESSOR, None, None, {1: [("gain", -6.0, False), ("thrs", -30.0, False)], 3: [("look", 0.01, False), ("rato", 20.0, False)]}),
    ("gate", _capi.FX_GATE, None, None, {1: [("thrs", -10.0, False)], 3: [("hold", 0.0, False), ("rnge", -20.0, False)]}),
    ("dist", _capi.FX_DISTORTION, None, None, {1: [("driv", 3.0, False)], 2: [("mix ", 0.3, False)], 4: [("type", 4, False)]}),
]


@pytest.mark.parametrize("name,kind,params,seeds,updates", RAMPS, ids=[c[0] for c in RAMPS])
def test_effect_parameter_ramps(name, kind, params, seeds, updates):
    """Parameter updates between blocks: smoother targets, per-frame coefficient branches, type switches."""
    a, b, _, _ = run_pair(kind, params, seeds, blocks=6, frames=700, signal="noise", updates=updates)
    check(a, b)


def test_effect_reset_messages():
    for kind, seeds in ((_capi.FX_DELAY, None), (_capi.FX_REVERB, workloads.reverb_seeds(1)), (_capi.FX_CHORUS, None)):
        e_gpu = gpu_effect(kind, None, seeds)
        e_cpu = oracle.OracleEffect(kind, None, seeds)
        for e in (e_gpu, e_cpu):
            e.initialize(SR, 2, 4096)
        x = workloads.test_signal(3 * 1024, seed=5)
        a, b = x.copy(), x.copy()
        for blk in range(3):
            if blk == 2:
                e_gpu.reset()
                e_cpu.reset()
            sl = slice(blk * 2048, (blk + 1) * 2048)
            e_gpu.process(a[sl])
            e_cpu.process(b[sl])
        check(a, b)


def test_effect_tail_and_errors():
    import phonic_amd

    for kind in range(10):
        seeds = workloads.reverb_seeds(0) if kind == _capi.FX_REVERB else None
        e_gpu = gpu_effect(kind, None, seeds)
        e_cpu = oracle.OracleEffect(kind, None, seeds)
        e_gpu.initialize(SR, 2, 1024)
        e_cpu.initialize(SR, 2, 1024)
        assert e_gpu.process_tail() == e_cpu.process_tail(), _capi.FX_NAMES[kind]
        with pytest.raises(phonic_amd.PhonicError):
            e_gpu.set_parameter("zzzz", 0.0)
    with pytest.raises(phonic_amd.PhonicError):
        phonic_amd.Effect(_capi.FX_REVERB).initialize(SR, 1, 1024)  # "ReverbEffect only supports stereo I/O"
    with pytest.raises(phonic_amd.PhonicError):
        phonic_amd.Effect(_capi.FX_GAIN).reset()  # no message type


def test_block_size_edge_cases():
    """Empty, single-frame, ragged and maximum-size blocks."""
    for kind, seeds in ((_capi.FX_REVERB, workloads.reverb_seeds(2)), (_capi.FX_CHORUS, None), (_capi.FX_EQ5, None)):
        e_gpu = gpu_effect(kind, {"gan2": 5.0} if kind == _capi.FX_EQ5 else None, seeds)
        e_cpu = oracle.OracleEffect(kind, {"gan2": 5.0} if kind == _capi.FX_EQ5 else None, seeds)
        for e in (e_gpu, e_cpu):
            e.initialize(SR, 2, 4096)
        sizes = [0, 1, 3, 4096, 17, 1000, 2, 333]
        x = workloads.test_signal(sum(sizes), seed=3)
        a, b = x.copy(), x.copy()
        off = 0
        for n in sizes:
            sl = slice(off * 2, (off + n) * 2)
            e_gpu.process(a[sl])
            e_cpu.process(b[sl])
            off += n
        check(a, b)


def test_reverb_long_run_state():
    """20 blocks of 1024: vibrato phases advance by repeated f64 addition in the reference; outputs must track the oracle
    over a long run (feedback network, 13 delay lines wrapping several times)."""
    seeds = workloads.reverb_seeds(7)
    a, b, _, _ = run_pair(_capi.FX_REVERB, {"room": 0.3, "wet ": 0.6}, seeds, blocks=20, frames=1024, signal="sine")
    check(a, b)


@pytest.mark.parametrize("mode,ftyp,dlay", [(0, 0, 2.0), (1, 1, 3.5), (0, 2, 1.4), (1, 0, 375.0)])
def test_delay_time_parallel_path_short_delays_and_lfo_phase(mode, ftyp, dlay):
    """The time-parallel DelayEffect path (LFO depths 0, nothing ramping): delays of a few ms force many chunks per block
    (chunk <= floor(delay_samples) - 1), both routing modes and the three feedback-filter types. The LFO only advances its phase
    there (closed-form f32 accumulation); switching the LFO depths on afterwards hands over to the serial path, which must
    continue from exactly the phase the reference would have — any drift shows up in the modulated blocks. The 375 ms case
    (18 000 frames, the default) runs 45 blocks so that two echoes pass through the wet path before the hand-over."""
    params = {"mode": mode, "dlay": dlay, "fdbk": 0.6, "ftyp": ftyp, "driv": 0.3, "lfor": 3.7}
    long_run = dlay > 100.0
    blocks = 45 if long_run else 7
    updates = {(40 if long_run else 4): [("lfdt", 0.02, False), ("lfdf", 0.3, False)]}
    a, b, _, _ = run_pair(_capi.FX_DELAY, params, None, blocks=blocks, frames=1024, signal="noise", updates=updates)
    check(a, b)
    if long_run:
        assert_wet_path_audible(a, params, blocks, 1024, "noise")


def assert_wet_path_audible(out, params, blocks, frames, signal, first_echo_frame=18000):
    """Against the same Delay with wet = 0 (oracle) the output must differ by > 1e-3 after the first echo: the delay-line read,
    feedback filter, saturation and DC filter at the config's own delay time are really compared, not zeros."""
    dry_params = dict(params or {}, **{"wet_": 0.0})
    e = oracle.OracleEffect(_capi.FX_DELAY, dry_params, None)
    e.initialize(SR, 2, 4096)
    x = workloads.test_signal(blocks * frames, seed=_capi.FX_DELAY + 11, kind=signal)
    for blk in range(blocks):
        e.process(x[blk * frames * 2:(blk + 1) * frames * 2])
    late = slice(2 * first_echo_frame, None)
    assert float(np.abs(out[late].astype(np.float64) - x[late].astype(np.float64)).max()) > 1e-3, "the delay's wet path is inaudible in this run"


@pytest.mark.parametrize("signal", ["noise", "burst"])
def test_delay_default_two_echoes(signal):
    """DelayEffect::new() — 375 ms = 18 000 frames at 48 kHz, feedback 0.5 (src/effect/delay.rs:124-177) — over 45 blocks of 1024
    frames: the first echo returns in block 17, its feedback copy in block 35."""
    a, b, _, _ = run_pair(_capi.FX_DELAY, None, None, blocks=45, frames=1024, signal=signal)
    check(a, b)
    assert_wet_path_audible(a, None, 45, 1024, signal)


@pytest.mark.parametrize("params", [
    {"rate": 0.7, "dpth": 0.5, "fdbk": 0.5, "dlay": 12.0},
    {"rate": 9.0, "dpth": 1.0, "fdbk": -0.8, "dlay": 3.0, "fltt": 1, "fltf": 400.0, "phas": 2.0},
    {"rate": 0.05, "dpth": 0.2, "fdbk": 0.9, "dlay": 40.0, "wet_": 1.0},
])
def test_chorus_time_parallel_path(params):
    """The time-parallel ChorusEffect path over 40 blocks: LFO phase arrays from the exact piecewise closed form (many wraps and
    binade crossings at 9 Hz), chunking by the shortest line lag (3 ms -> chunks of ~140 frames), negative feedback, the
    high-pass pre-filter; then a rate change (serial path while it ramps) that must continue from the exact LFO state."""
    updates = {30: [("rate", 2.0, False)]}
    a, b, _, _ = run_pair(_capi.FX_CHORUS, params, None, blocks=40, frames=1024, signal="noise", updates=updates)
    check(a, b)


@pytest.mark.parametrize("params", [
    {"thrs": -0.01, "rato": 20.0, "knee": 0.0, "attk": 0.02, "rels": 2.0, "gain": 0.0, "look": 0.02},   # brick-wall limiter
    {"thrs": -18.0, "rato": 4.0, "knee": 6.0, "attk": 0.005, "rels": 0.1, "gain": 3.0, "look": 0.04},     # compressor with soft knee
    {"thrs": -30.0, "rato": 20.0, "knee": 2.0, "attk": 0.001, "rels": 0.1, "gain": 6.0, "look": 0.001},  # 48-frame look-ahead
])
def test_compressor_time_parallel_path(params):
    """The time-parallel compressor / limiter: sliding-window maximum by doubling (look-ahead windows of 48, 960 and 1920 frames),
    parallel gain law, serial envelope only. 24 blocks of bursty input (the tracked peak expires and is re-found many times); then
    the ratio crosses 20 (compressor <-> limiter switch uses the peak state the fast path maintained) and the makeup gain ramps."""
    flip = 4.0 if params["rato"] >= 20.0 else 20.0
    updates = {12: [("rato", flip, False)], 18: [("gain", -3.0, False)]}
    a, b, _, _ = run_pair(_capi.FX_COMPRESSOR, params, None, blocks=24, frames=1024, signal="burst", updates=updates)
    check(a, b)
    a, b, _, _ = run_pair(_capi.FX_COMPRESSOR, params, None, blocks=9, frames=700, signal="noise")
    check(a, b)


INDEX_CASES = [
    ("reverb_default", _capi.FX_REVERB, None, workloads.reverb_seeds(3), 16),
    ("reverb_big_room", _capi.FX_REVERB, {"room": 1.0, "wet ": 0.5}, workloads.reverb_seeds(5), 16),
    ("reverb_small_room", _capi.FX_REVERB, {"room": 0.0, "wet ": 1.0}, workloads.reverb_seeds(4), 16),
    ("delay_lfo_time", _capi.FX_DELAY, {"dlay": 10.0, "lfdt": 0.1, "ldfb": 0.3, "lfor": 5.0, "lfos": 1}, None, 2),
    ("delay_short", _capi.FX_DELAY, {"mode": 1, "dlay": 3.5, "fdbk": 0.6}, None, 2),
    ("chorus_default", _capi.FX_CHORUS, None, None, 2),
    ("chorus_fast_lfo", _capi.FX_CHORUS, {"rate": 9.0, "dpth": 1.0, "fdbk": -0.8, "dlay": 3.0}, None, 2),
]
INDEX_FLIP_RATE_MAX = 1e-4   # flips per logged read; a flip moves one tap by one frame (its weight is the interpolation fraction ~ 0 there)


@pytest.mark.parametrize("name,kind,params,seeds,per_frame", INDEX_CASES, ids=[c[0] for c in INDEX_CASES])
def test_delay_line_read_index_streams(name, kind, params, seeds, per_frame):
    """no index is derived from a transcendental EXCEPT the read positions of the reverb's vibrato lines
    (floor(count + (sin(phase) + 1) * 7), reverb.rs:563-567) and of the interpolated delay lines of Delay / Chorus (floor(write_pos -
    delay), dsp/delay.rs:120-126): device libm (and, in the reverb, the angle-addition form of sin) may land on the other side of an
    integer where glibc does not. Those index streams are exported by both sides (pg_effect_debug_index_log / po_index_log_*) and compared
    on their own — a flipped index must not hide inside the RMS figure. 24 blocks of 1024 frames: every read of every line is logged."""
    import ctypes as C

    blocks, frames = 24, 1024
    words = frames * per_frame
    e_gpu = gpu_effect(kind, params, seeds)
    e_cpu = oracle.OracleEffect(kind, params, seeds)
    for e in (e_gpu, e_cpu):
        e.initialize(SR, 2, 4096)
    lib = e_gpu._lib
    x = workloads.test_signal(blocks * frames, seed=kind + 31, kind="noise")
    a, b = x.copy(), x.copy()
    flips = total = 0
    worst = 0
    for blk in range(blocks):
        sl = slice(blk * frames * 2, (blk + 1) * frames * 2)
        assert lib.pg_effect_debug_index_log(e_gpu._h, None, words) == 0
        e_gpu.process(a[sl])
        got = np.zeros(words, np.int32)
        assert lib.pg_effect_debug_index_log(e_gpu._h, got.ctypes.data_as(C.POINTER(C.c_int32)), words) == 0
        oracle.lib().po_index_log_begin()
        e_cpu.process(b[sl])
        want = np.zeros(words, np.int32)
        n = oracle.lib().po_index_log_end(want.ctypes.data_as(C.POINTER(C.c_int32)), words)
        assert n == words, (n, words)
        assert (got >= 0).all(), "the time-parallel path did not log every read (serial fallback?)"
        diff = got != want
        flips += int(diff.sum())
        total += words
        if diff.any():
            worst = max(worst, int(np.abs(got[diff].astype(np.int64) - want[diff]).max()))
    rate = flips / total
    print(f"{name}: {flips} index flips in {total} reads (rate {rate:.2e}, largest distance {worst})")
    assert rate <= INDEX_FLIP_RATE_MAX, f"{flips} flips in {total} reads"
    check(a, b)


def test_parameter_updates_survive_empty_process_calls_and_long_queues():
    """pg_effect_set_parameter queues commands that the next launch applies at the head of the frames it renders. A process call of zero frames
    renders nothing — the queue must stay (it used to be cleared: the updates were lost), and more updates than the queue's first allocation
    holds must all arrive, in order (the queue grows; a flush launch of zero frames cannot apply them)."""
    for kind, pid, lo, hi in ((_capi.FX_FILTER, "cuto", 200.0, 9000.0), (_capi.FX_GAIN, "gain", 0.1, 2.0), (_capi.FX_COMPRESSOR, "thrs", -50.0, -5.0)):
        e_gpu, e_cpu = gpu_effect(kind), oracle.OracleEffect(kind, None, None)
        for e in (e_gpu, e_cpu):
            e.initialize(SR, 2, 1024)
        x = workloads.test_signal(4 * 512, seed=3)
        a, b = x.copy(), x.copy()
        empty = np.zeros(0, np.float32)
        for blk in range(4):
            for e in (e_gpu, e_cpu):
                if blk == 1:
                    e.set_parameter(pid, lo, False)
                    e.process(empty)                      # zero frames: the update above must still apply to block 1
                if blk == 2:
                    for k in range(150):                   # far more than the 64 commands the queue starts with; the last one wins
                        e.set_parameter(pid, lo + (hi - lo) * ((k * 37) % 150) / 149.0, False)
            sl = slice(blk * 1024, (blk + 1) * 1024)
            e_gpu.process(a[sl])
            e_cpu.process(b[sl])
        check(a, b)
        assert not np.array_equal(a[1024:2048], x[1024:2048])


@pytest.mark.parametrize("channels", [1, 3, 6])
@pytest.mark.parametrize("kind,params,updates", [
    (_capi.FX_GAIN, {"gain": 0.7, "dcfm": 2}, {1: [("gain", 2.0, False)], 3: [("dcfm", 3, False)]}),
    (_capi.FX_FILTER, {"type": 0, "cuto": 1800.0, "fltq": 0.9}, {1: [("cuto", 6000.0, False)], 2: [("fltq", 2.5, False), ("type", 3, False)]}),
    (_capi.FX_EQ5, {"gan1": 4.0, "gan3": -6.0}, {1: [("gan2", 9.0, False), ("frq4", 5000.0, False)], 3: [("bw_3", 0.7, False)]}),
    (_capi.FX_DISTORTION, {"type": 0, "driv": 1.5, "mix ": 0.6}, {1: [("driv", 3.0, False)], 2: [("mix ", 1.0, False)], 4: [("type", 4, False)]}),
])
def test_channel_counts_other_than_stereo(kind, params, updates, channels):
    """Filter, Eq5, Gain and Distortion process ANY channel count in the reference (filter.rs:144-201, eq5.rs:297-326, gain.rs:143-166,
    distortion.rs:326-361): channels are independent, the smoothers step once per frame. Mono, three and six channels (an odd count leaves a
    half-filled channel pair), each channel with its own signal, parameter ramps that start mid-run, ragged and empty blocks — against the oracle's
    N-channel loops. The stereo-only effects keep the reference's error for anything but two channels (e.g. reverb.rs:399-403)."""
    import phonic_amd

    sizes = [256, 1, 0, 511, 1024, 300]
    rng = np.random.default_rng(100 * kind + channels)
    x = (0.3 * rng.standard_normal(sum(sizes) * channels)).astype(np.float32)
    for c in range(channels):
        x[c::channels] *= np.float32(0.4 + 0.3 * c)          # (and a DC offset on one channel: the Gain's DC filter has something to remove)
    x[0::channels] += np.float32(0.05)
    e_gpu, e_cpu = phonic_amd.Effect(kind, params), oracle.OracleEffect(kind, params)
    outs = []
    for e in (e_gpu, e_cpu):
        e.initialize(SR, channels, 1024)
        y, off = x.copy(), 0
        for blk, n in enumerate(sizes):
            for (id4, val, norm) in updates.get(blk, []):
                e.set_parameter(id4, val, norm)
            e.process(y[off:off + n * channels])
            off += n * channels
        outs.append(y)
    check(outs[0], outs[1])
    assert not np.array_equal(outs[0], x)
    for c in range(1, channels):
        assert not np.array_equal(outs[0][c::channels], outs[0][0::channels])         # every channel carries its own audio
    with pytest.raises(phonic_amd.PhonicError) as ei:
        e_gpu.process(np.zeros(channels * 1024 + channels, np.float32))                 # more than max_frames
    assert ei.value.code == _capi.PG_ERR_PARAMETER
    if channels > 1:
        with pytest.raises(phonic_amd.PhonicError):
            e_gpu.process(np.zeros(channels + 1, np.float32))                           # not a whole number of frames


def test_stereo_only_effects_reject_other_channel_counts():
    import phonic_amd

    for kind in (_capi.FX_PANNING, _capi.FX_DELAY, _capi.FX_REVERB, _capi.FX_CHORUS, _capi.FX_COMPRESSOR, _capi.FX_GATE):
        for ch in (1, 3):
            with pytest.raises(phonic_amd.PhonicError) as ei:
                phonic_amd.Effect(kind).initialize(SR, ch, 1024)
            assert ei.value.code == _capi.PG_ERR_PARAMETER and "only supports stereo I/O" in str(ei.value)
    for kind in (_capi.FX_GAIN, _capi.FX_FILTER, _capi.FX_EQ5, _capi.FX_DISTORTION):
        with pytest.raises(phonic_amd.PhonicError):
            phonic_amd.Effect(kind).initialize(SR, 0, 1024)
