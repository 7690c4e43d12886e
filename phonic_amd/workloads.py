"""Deterministic synthetic inputs and graph builders shared by bench.py, smoke() and the parity tests
(SURVEY.md §8d). Pure numpy; drives either the product graph or the oracle graph (tests only) through the common
GraphHandle API."""
import numpy as np

from phonic_amd import _capi

MASK64 = (1 << 64) - 1


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def reverb_seeds(voice):
    """fpd_l/fpd_r = next u32 >= 16386, vib_phase[line][ch] = u53 * 2*pi — from splitmix64(0x5EED0000 + voice)."""
    s = (0x5EED0000 + voice) & MASK64
    fpd = []
    while len(fpd) < 2:
        s, z = splitmix64(s)
        v = z & 0xFFFFFFFF
        if v >= 16386:
            fpd.append(v)
    ph = []
    for _ in range(16):
        s, z = splitmix64(s)
        ph.append((z >> 11) * (1.0 / (1 << 53)) * 2.0 * np.pi)
    return fpd[0], fpd[1], ph


def voice_freq(i):
    return 55.0 * 2.0 ** ((i % 61) / 12.0)


def tone_buffer(i, sr_src, seconds=2.0, channels=2):
    """File-like voice i: L = 0.05 sin(2 pi f n / sr), R = 0.05 sin(2 pi 1.01 f n / sr + 0.5), f64 -> f32, plus the extra
    zero frame symphonia decoding appends (reference src/source/file/buffer.rs:103-104)."""
    n = int(round(seconds * sr_src))
    t = np.arange(n, dtype=np.float64)
    f = voice_freq(i)
    l = 0.05 * np.sin(2 * np.pi * f * t / sr_src)
    if channels == 1:
        buf = l.astype(np.float32)
        return np.concatenate([buf, np.zeros(1, np.float32)])
    r = 0.05 * np.sin(2 * np.pi * 1.01 * f * t / sr_src + 0.5)
    buf = np.stack([l, r], axis=1).astype(np.float32).reshape(-1)
    return np.concatenate([buf, np.zeros(2, np.float32)])


def voice_pan(i):
    return -1.0 + 2.0 * ((i * 37) % 101) / 100.0


def voice_level(total_voices):
    """Per-voice volume: 1/sqrt(V) up to 1024 voices, constant above. A 0.05-peak tone at 1/sqrt(V) falls below the
    sub-mixer silence gate (SILENCE_THRESHOLD 0.001, src/source/mixed/submixer.rs:47-77) from ~2500 voices on: after 2 s of audio
    every per-voice sub-mixer would be dropped from the sum and an 8192-voice run would go silent. Holding the level at 1/32 keeps
    every voice above the gate; the bus of V > 1024 incoherent voices then peaks near 0.05 * sqrt(V) / 32 (0.14 at 8192)."""
    return float(np.float32(1.0 / np.sqrt(min(int(total_voices), 1024))))


def build_headline(g, n_voices, first_voice=0, total_voices=None, seconds=2.0):
    """H: stereo 44.1 kHz looped voices -> cubic -> volume 1/sqrt(V), pan -> per-voice Reverb (own sub-mixer) -> sum."""
    total = total_voices or n_voices
    vol = voice_level(total)
    for k in range(n_voices):
        i = first_voice + k
        m = g.add_mixer()
        g.add_effect(m, _capi.FX_REVERB, reverb_seeds=reverb_seeds(i))
        g.add_voice(m, tone_buffer(i, 44100, seconds), 2, 44100, volume=vol, panning=float(np.float32(voice_pan(i))), has_repeat=1,
                    repeat=_capi.PG_REPEAT_FOREVER)


def build_c2(g, n_voices=64, seconds=2.0):
    """C2: 64 stereo 48 kHz file voices on the main mixer, Eq5 + Reverb on the bus."""
    vol = voice_level(n_voices)
    for i in range(n_voices):
        g.add_voice(0, tone_buffer(i, 48000, seconds), 2, 48000, volume=vol, panning=float(np.float32(voice_pan(i))), has_repeat=1,
                    repeat=_capi.PG_REPEAT_FOREVER)
    g.add_effect(0, _capi.FX_EQ5, params={"gan1": 3.0, "gan3": -4.0, "gan5": 2.0})
    g.add_effect(0, _capi.FX_REVERB, reverb_seeds=reverb_seeds(0))


def build_c3(g, n_voices=1024, first_voice=0, total_voices=None, seconds=2.0):
    """C3: mono sine voices, per-voice Filter(Lowpass 2 kHz, Q 0.707) + Chorus, mixer reduce."""
    total = total_voices or n_voices
    vol = voice_level(total)
    for k in range(n_voices):
        i = first_voice + k
        m = g.add_mixer()
        g.add_effect(m, _capi.FX_FILTER, params={"type": 0, "cuto": 2000.0, "fltq": 0.707})
        g.add_effect(m, _capi.FX_CHORUS)
        g.add_voice(m, tone_buffer(i, 48000, seconds, channels=1), 1, 48000, volume=vol, panning=float(np.float32(voice_pan(i))),
                    has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)


def build_c4(g, n_voices=256, seconds=2.0):
    """C4: 256 stereo 44.1 kHz voices -> cubic -> main mixer, limiter on the bus."""
    vol = voice_level(n_voices)
    for i in range(n_voices):
        g.add_voice(0, tone_buffer(i, 44100, seconds), 2, 44100, volume=vol, panning=float(np.float32(voice_pan(i))), has_repeat=1,
                    repeat=_capi.PG_REPEAT_FOREVER)
    # CompressorEffect::new_limiter(): threshold -0.01, ratio 20, knee 0, makeup 0, look-ahead = attack (compressor.rs:114-157)
    g.add_effect(0, _capi.FX_COMPRESSOR, params={"thrs": -0.01, "rato": 20.0, "knee": 0.0, "attk": 0.02, "rels": 2.0, "gain": 0.0, "look": 0.02})


def build_c5(g, n_voices=8192, first_voice=0, total_voices=None, seconds=2.0):
    """C5: per-voice Filter -> Eq5 -> Delay -> Reverb on stereo 48 kHz voices."""
    total = total_voices or n_voices
    vol = voice_level(total)
    for k in range(n_voices):
        i = first_voice + k
        m = g.add_mixer()
        g.add_effect(m, _capi.FX_FILTER, params={"type": 0, "cuto": 8000.0, "fltq": 0.707})
        g.add_effect(m, _capi.FX_EQ5, params={"gan1": 3.0, "gan3": -4.0, "gan5": 2.0})
        g.add_effect(m, _capi.FX_DELAY)
        g.add_effect(m, _capi.FX_REVERB, reverb_seeds=reverb_seeds(i))
        g.add_voice(m, tone_buffer(i, 48000, seconds), 2, 48000, volume=vol, panning=float(np.float32(voice_pan(i))), has_repeat=1,
                    repeat=_capi.PG_REPEAT_FOREVER)


def test_signal(n_frames, seed=1, kind="noise"):
    """Interleaved stereo test signals for the per-effect parity tests."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        x = rng.standard_normal(n_frames * 2) * 0.25
    elif kind == "sine":
        t = np.arange(n_frames, dtype=np.float64)
        x = np.stack([0.5 * np.sin(2 * np.pi * 1000.0 * t / 48000.0), 0.4 * np.sin(2 * np.pi * 1500.0 * t / 48000.0 + 0.3)], axis=1).reshape(-1)
    elif kind == "impulse":
        x = np.zeros(n_frames * 2)
        x[0] = 1.0
        x[1] = -0.5
    elif kind == "burst":  # loud burst then silence: exercises envelope followers / gates
        x = rng.standard_normal(n_frames * 2) * 0.5
        env = np.repeat((np.arange(n_frames) % 2000 < 700).astype(np.float64), 2)
        x = x * env
    else:
        raise ValueError(kind)
    return x.astype(np.float32)
