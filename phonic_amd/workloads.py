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


def build_dyn(g, n_voices, audio_seconds, churn_pct_per_s=0.0, silent_pct=0.0, first_frame=0, seed=1234, sample_rate=48000):
    """The headline layout off its steady state (bench.py --workload dyn): per-voice sub-mixers with a Reverb, where
    `silent_pct` % of the voices are one-shots that ended long ago (a 0.05 s sample at time 0: once the reverb's tail — 5.6 s at the default
    room size, reverb.rs:449-467 — and the sub-mixer's 2 s silence gate have passed, EffectProcessor and SubMixerProcessor bypass them,
    src/source/mixed/effect.rs:56-84, submixer.rs:47-77), and `churn_pct_per_s` % of the voices per second of audio are notes that END and
    RESTART at random sample times inside [first_frame, first_frame + audio_seconds): the playing source is stopped there (StopSource: its
    fade-out) and a successor — added up front with that start time, as Player::play_file_source schedules a source (player.rs:519-602) —
    takes over a random gap later. Returns {"mixers", "reverbs", "voices" (the voice playing at first_frame, per unit), "silent" (set of units),
    "restarts": [(stop_time, unit, old_voice, new_voice)] sorted by time}."""
    rng = np.random.default_rng(seed)
    vol = voice_level(n_voices)
    n_silent = int(round(n_voices * silent_pct / 100.0))
    silent = set(int(u) for u in rng.permutation(n_voices)[:n_silent])
    out = {"mixers": [], "reverbs": [], "voices": [], "silent": silent, "restarts": []}
    for i in range(n_voices):
        m = g.add_mixer()
        out["mixers"].append(m)
        out["reverbs"].append(g.add_effect(m, _capi.FX_REVERB, reverb_seeds=reverb_seeds(i)))
        if i in silent:
            out["voices"].append(g.add_voice(m, tone_buffer(i, 44100, 0.05), 2, 44100, volume=vol, panning=float(np.float32(voice_pan(i)))))
        else:
            out["voices"].append(g.add_voice(m, tone_buffer(i, 44100, 2.0), 2, 44100, volume=vol, panning=float(np.float32(voice_pan(i))), has_repeat=1,
                                             repeat=_capi.PG_REPEAT_FOREVER))
    live = [u for u in range(n_voices) if u not in silent]
    n_restarts = int(round(len(live) * churn_pct_per_s / 100.0 * audio_seconds))
    if n_restarts and live:
        times = np.sort(rng.integers(first_frame, first_frame + int(audio_seconds * sample_rate), n_restarts))
        current = {u: out["voices"][u] for u in live}
        busy_until = {}
        for t in times:
            t = int(t)
            u = int(live[int(rng.integers(0, len(live)))])
            if busy_until.get(u, -1) >= t:   # (its successor has not started yet: another unit takes this restart)
                free = [v for v in live if busy_until.get(v, -1) < t]
                if not free:
                    continue
                u = int(free[int(rng.integers(0, len(free)))])
            gap = int(rng.integers(0, int(0.2 * sample_rate)))
            nv = g.add_voice(out["mixers"][u], tone_buffer(u + 7, 44100, 2.0), 2, 44100, volume=vol, panning=float(np.float32(voice_pan(u))), has_repeat=1,
                             repeat=_capi.PG_REPEAT_FOREVER, start_time=t + gap)
            out["restarts"].append((t, u, current[u], nv))
            current[u] = nv
            busy_until[u] = t + gap
    return out


class DynDriver:
    """Schedules the dynamic workload's commands call by call (bench.py --workload dyn and its parity test drive the product graph and the
    oracle through the same object): before the call that renders frames [t0, t1) — `events_per_s` x (t1 - t0) / sample_rate commands on average
    (Poisson), each at a random sample time inside the call on a random audible unit: the reverb's `wet`, the playing source's volume, its
    panning, equal shares — and the StopSource messages of the planned restarts that fall into the call."""

    def __init__(self, plan, events_per_s, seed, sample_rate=48000, kinds=(0, 1, 2)):
        self.plan, self.events_per_s, self.sr, self.kinds = plan, float(events_per_s), sample_rate, tuple(kinds)
        self.rng = np.random.default_rng(seed)
        self.live = [u for u in range(len(plan["voices"])) if u not in plan["silent"]]
        self.current = {u: plan["voices"][u] for u in self.live}
        self.restart_i = 0

    def schedule(self, g, t0, t1):
        n_cmds = 0
        rng, plan = self.rng, self.plan
        n_ev = int(rng.poisson(self.events_per_s * (t1 - t0) / self.sr)) if self.events_per_s > 0 and self.live else 0
        for _ in range(n_ev):
            u = int(self.live[int(rng.integers(0, len(self.live)))])
            t = int(rng.integers(t0, t1))
            kind = self.kinds[int(rng.integers(0, len(self.kinds)))]
            if kind == 0:
                g.schedule_param(plan["reverbs"][u], "wet ", float(np.float32(rng.uniform(0.2, 0.5))), t)
            elif kind == 1:
                g.set_voice_volume(self.current[u], float(np.float32(rng.uniform(0.5, 1.0) / 32.0)), t)
            else:
                g.set_voice_panning(self.current[u], float(np.float32(rng.uniform(-1.0, 1.0))), t)
            n_cmds += 1
        rs = plan["restarts"]
        while self.restart_i < len(rs) and rs[self.restart_i][0] < t1:
            t, u, old, new = rs[self.restart_i]
            if t >= t0:
                g.stop_voice(old, t)
                self.current[u] = new
                n_cmds += 1
            self.restart_i += 1
        return n_cmds


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
