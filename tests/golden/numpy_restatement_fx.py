#!/usr/bin/env python3
"""Independent restatement, part 2: the remaining eight stock effects and the primitives they share — written from the Rust sources in
plain Python scalars (`float` = f64, `numpy.float32` = f32, every f32 operation rounded separately), separately from the C++ oracle
("per effect at default + one ramped-parameter case").

  primitives   ExponentialSmoothedValue / LinearSmoothedValue / SpringSmoothedValue      src/utils/smoothing.rs:130-228,245-402,422-534
               SmoothedParameterValue::apply_update (Raw, clamped)                        src/parameter/smoothed.rs:136-157
               BiquadFilterCoefficients::apply / BiquadFilter::process_sample             src/utils/dsp/filters/biquad.rs:153-271,314-322
               SvfFilterCoefficients::apply / SvfFilter::process_sample                   src/utils/dsp/filters/svf.rs:137-168,211-222
               DcFilter                                                                    src/utils/dsp/filters/dc.rs:54-88
               Lfo (deterministic shapes) + sine_approx                                    src/utils/dsp/lfo.rs:9-19,63-169,234-239
               EnvelopeFollower                                                            src/utils/dsp/envelope.rs:13-60
               InterpolatedDelayLine, LookupDelayLine                                      src/utils/dsp/delay.rs:79-156,172-271
               db_to_linear, panning_factors                                               src/utils.rs:41-62
  effects      Gain (gain.rs:123-205)  Panning (pan.rs:84-191)  Filter (filter.rs:82-237)  Eq5 (eq5.rs:152-363)  Delay (delay.rs:70-454)
               Chorus (chorus.rs:139-394)  Compressor (compressor.rs:95-294)  Gate (gate.rs:48-195)  Distortion (distortion.rs:60-361)

f32 libm calls (expf, log10f, powf, atanf, sinf, fmodf) go to the same glibc the oracle links (ctypes), f64 ones through `math`
(the same libm): the comparison in tests/test_golden.py is bit for bit. Run as a script it writes tests/golden/independent_fx.npz.
"""
import ctypes
import math
import os
import sys

import numpy as np

F = np.float32
_libm = ctypes.CDLL("libm.so.6")
for _n in ("expf", "log10f", "atanf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ("powf", "fmodf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float, ctypes.c_float]


def expf(x):
    return F(_libm.expf(float(x)))


def log10f(x):
    return F(_libm.log10f(float(x)))


def atanf(x):
    return F(_libm.atanf(float(x)))


def sinf(x):
    return F(_libm.sinf(float(x)))


def powf(x, y):
    return F(_libm.powf(float(x), float(y)))


def fmodf(x, y):
    return F(_libm.fmodf(float(x), float(y)))


def fabs(x):
    return F(abs(x))


def fclamp(x, lo, hi):  # f32::clamp
    x = F(x)
    if x < lo:
        return F(lo)
    if x > hi:
        return F(hi)
    return x


def fmax(a, b):
    return F(a) if a > b else F(b)


def fmin(a, b):
    return F(a) if a < b else F(b)


def round_half_away(x):  # f32::round
    x = float(x)
    return math.copysign(math.floor(abs(x) + 0.5), x)


EPS100 = F(F(1.1920929e-07) * F(100.0))
PI32 = F(3.14159274101257324)
TAU32 = F(6.28318548202514648)


# ---- smoothers ----------------------------------------------------------------------------------------------------------------
class ExpSm:  # ExponentialSmoothedValue
    def __init__(self, value, sr, inertia=None):
        self.current = self.target = F(value)
        self.inertia = F(1.0) / F(256.0) if inertia is None else F(inertia)
        self.comp = F(44100.0) / F(sr)

    def need_ramp(self):
        add = F(F(F(self.target - self.current) * self.inertia) * self.comp)
        return fabs(add) > EPS100

    def ramp(self):
        self.current = F(self.current + F(F(F(self.target - self.current) * self.inertia) * self.comp))

    def init(self, v):
        self.target = self.current = F(v)

    def set_target(self, t):
        self.target = F(t)
        if not self.need_ramp():
            self.current = self.target

    def next(self):
        if self.need_ramp():
            self.ramp()
            return self.current
        return self.target


class LinSm:  # LinearSmoothedValue::default().with_step(step), then init(value), then set_sample_rate(sr)
    def __init__(self, value, sr, step=0.01):
        self.step = F(step)
        self.current = self.target = F(value)
        self.pending = 0
        self.comp = F(44100.0) / F(sr)
        self.current_step = F(F(-self.step) * self.comp) if self.current > self.target else F(self.step * self.comp)

    def need_ramp(self):
        return self.pending > 0

    def ramp(self):
        if self.pending > 0:
            self.current = F(self.current + self.current_step)
            self.pending -= 1
            if self.pending == 0:
                self.current = self.target

    def init(self, v):
        self.target = self.current = F(v)
        self.pending = 0

    def set_target(self, t):
        self.target = F(t)
        if self.current == self.target:
            self.pending = 0
            return
        self.current_step = F(F(-self.step) * self.comp) if self.current > self.target else F(self.step * self.comp)
        steps = F(F(self.target - self.current) / self.current_step)
        self.pending = int(max(round_half_away(steps), 0.0))
        if self.pending == 0:
            self.current = self.target

    def next(self):
        if self.need_ramp():
            self.ramp()
            return self.current
        return self.target


class SpringSm:  # SpringSmoothedValue::default().with_duration(d)
    def __init__(self, value, sr, duration):
        self.current = self.target = F(value)
        self.velocity = F(0.0)
        self.omega = F(5.5) / F(float(duration))
        self.comp = F(44100.0) / F(sr)

    def need_ramp(self):
        return fabs(self.velocity) > EPS100 or fabs(F(self.target - self.current)) > EPS100

    def ramp(self):
        omega = F(self.omega * self.comp)
        k = F(omega * omega)
        d = F(F(2.0) * omega)
        self.velocity = F(self.velocity + F(F(F(self.target - self.current) * k) - F(self.velocity * d)))
        self.current = F(self.current + self.velocity)

    def init(self, v):
        self.current = self.target = F(v)
        self.velocity = F(0.0)

    def set_target(self, t):
        self.target = F(t)

    def next(self):
        if self.need_ramp():
            self.ramp()
            return self.current
        return self.target


# ---- filters --------------------------------------------------------------------------------------------------------------------
LOWPASS, HIGHPASS, BANDPASS, NOTCH, PEAK, ALLPASS, BELL, LOWSHELF, HIGHSHELF = range(9)


class BiquadCoefs:  # BiquadFilterCoefficients
    def __init__(self):
        self.key = (LOWPASS, 0, F(0.0), F(0.0), F(0.0))
        self.a1 = self.a2 = self.a3 = self.m0 = self.m1 = self.m2 = 0.0

    def set(self, ftype, sr, cutoff, q, gain):
        key = (ftype, sr, F(cutoff), F(q), F(gain))
        if key != self.key:
            self.key = key
            self.apply()

    def cutoff(self):
        return self.key[2]

    def set_cutoff(self, c):
        self.set(self.key[0], self.key[1], c, self.key[3], self.key[4])

    def set_type(self, t):
        self.set(t, self.key[1], self.key[2], self.key[3], self.key[4])

    def apply(self):
        ftype, sr, cutoff, q, gain = self.key
        assert sr > 0 and q > 0 and cutoff <= F(sr) / F(2.0)
        tg = math.tan(math.pi * float(cutoff) / float(sr))
        if ftype in (LOWPASS, HIGHPASS, BANDPASS, NOTCH, PEAK, ALLPASS):
            g = tg
            k = 1.0 / float(q)
            m = {LOWPASS: (0.0, 0.0, 1.0), HIGHPASS: (1.0, -k, -1.0), BANDPASS: (0.0, 1.0, 0.0), NOTCH: (1.0, -k, 0.0), PEAK: (1.0, -k, -2.0),
                 ALLPASS: (1.0, -2.0 * k, 0.0)}[ftype]
        elif ftype == BELL:
            a = math.pow(10.0, float(gain) / 40.0)
            g = tg
            k = 1.0 / (float(q) * a)
            m = (1.0, k * (a * a - 1.0), 0.0)
        elif ftype == LOWSHELF:
            a = math.pow(10.0, float(gain) / 40.0)
            g = tg / math.sqrt(a)
            k = 1.0 / float(q)
            m = (1.0, k * (a - 1.0), a * a - 1.0)
        else:
            a = math.pow(10.0, float(gain) / 40.0)
            g = tg * math.sqrt(a)
            k = 1.0 / float(q)
            m = (a * a, k * (1.0 - a) * a, 1.0 - a * a)
        self.a1 = 1.0 / (1.0 + g * (g + k))
        self.a2 = g * self.a1
        self.a3 = g * self.a2
        self.m0, self.m1, self.m2 = m


class Biquad:  # BiquadFilter
    def __init__(self):
        self.ic1 = self.ic2 = 0.0

    def tick(self, c, x):
        v0 = x
        v3 = v0 - self.ic2
        v1 = c.a1 * self.ic1 + c.a2 * v3
        v2 = self.ic2 + c.a2 * self.ic1 + c.a3 * v3
        self.ic1 = 2.0 * v1 - self.ic1
        self.ic2 = 2.0 * v2 - self.ic2
        return c.m0 * v0 + c.m1 * v1 + c.m2 * v2


SVF_LP, SVF_HP, SVF_BP = 0, 1, 2  # SvfFilterType: Lowpass, Highpass, Bandpass


class SvfCoefs:
    def __init__(self, ftype, sr, cutoff, res):
        self.key = None
        self.set(ftype, sr, cutoff, res)

    def set(self, ftype, sr, cutoff, res):
        key = (ftype, sr, F(cutoff), F(res))
        if key != self.key:
            self.key = key
            assert 0.0 <= res <= 1.0 and cutoff <= F(sr) / F(2.0)
            self.g = math.tan(math.pi * float(F(cutoff)) / float(sr))
            self.k = max(2.0 * (1.0 - float(F(res)) * 0.97), 0.03)
            self.a1 = 1.0 / (1.0 + self.g * (self.g + self.k))
            self.a2 = self.g * self.a1
            self.a3 = self.g * self.a2
            self.ftype = ftype


class Svf:
    def __init__(self):
        self.ic1 = self.ic2 = 0.0

    def tick(self, c, x):
        v3 = x - self.ic2
        v1 = c.a1 * self.ic1 + c.a2 * v3
        v2 = self.ic2 + c.a2 * self.ic1 + c.a3 * v3
        self.ic1 = 2.0 * v1 - self.ic1
        self.ic2 = 2.0 * v2 - self.ic2
        if c.ftype == SVF_LP:
            return v2
        if c.ftype == SVF_BP:
            return v1
        return x - c.k * v1 - v2


class Dc:  # DcFilter::new(sample_rate, mode)
    def __init__(self, sr, hz):
        self.y1 = self.x1 = 0.0
        self.set(sr, hz)

    def set(self, sr, hz):
        self.r = 1.0 - (math.tau * hz / float(sr))

    def reset(self):
        self.y1 = self.x1 = 0.0

    def tick(self, x):
        self.y1 = x - self.x1 + self.r * self.y1
        self.x1 = x
        return self.y1


def sine_approx(x):
    x = F(x)
    B = F(4.0) / PI32
    C = F(-4.0) / F(PI32 * PI32)
    P = F(0.225)
    y = F(F(B * x) + F(F(C * x) * fabs(x)))
    return F(F(P * F(F(y * fabs(y)) - y)) + y)


class Lfo:  # deterministic waveforms: 0 Sine, 1 Triangle, 2 RampUp, 3 RampDown, 4 Square
    def __init__(self, sr, rate, waveform=0):
        self.phase = F(0.0)
        self.inc = F(float(rate) / float(sr))
        self.wf = waveform

    def set_rate(self, sr, rate):
        self.inc = F(float(rate) / float(sr))

    def set_phase_degrees(self, ph):  # (despite its name: radians / TAU), rem_euclid(1.0)
        p = F(F(ph) / TAU32)
        r = fmodf(p, F(1.0))
        if r < 0.0:
            r = F(r + F(1.0))
        self.phase = r

    def run(self):
        ph = self.phase
        if self.wf == 0:
            p = F(ph * TAU32) if ph < F(0.5) else F(F(ph - F(1.0)) * TAU32)
            v = sine_approx(p)
        elif self.wf == 1:
            v = F(ph * F(4.0)) if ph < F(0.25) else (F(F(2.0) - F(ph * F(4.0))) if ph < F(0.75) else F(F(ph * F(4.0)) - F(4.0)))
        elif self.wf == 2:
            v = F(F(ph * F(2.0)) - F(1.0))
        elif self.wf == 3:
            v = F(F(1.0) - F(ph * F(2.0)))
        else:
            v = F(1.0) if ph < F(0.5) else F(-1.0)
        self.phase = F(self.phase + self.inc)
        if self.phase >= F(1.0):
            self.phase = F(self.phase - F(1.0))
        return v


class Envelope:  # EnvelopeFollower
    def __init__(self, sr, attack, release):
        self.sr = sr
        self.cur = F(0.0)
        self.set_times(attack, release)

    def set_times(self, attack, release):
        self.att = expf(F(F(-1.0) / F(F(attack) * F(self.sr)))) if attack > 0.0 else F(0.0)
        self.rel = expf(F(F(-1.0) / F(F(release) * F(self.sr)))) if release > 0.0 else F(0.0)

    def run(self, x):
        x = F(x)
        c = self.att if x > self.cur else self.rel
        self.cur = F(x + F(c * F(self.cur - x)))
        return self.cur


def next_pow2(n):
    p = 1
    while p < n:
        p <<= 1
    return p


class InterpDelay:  # InterpolatedDelayLine<1>
    def __init__(self, max_size):
        n = next_pow2(max_size)
        self.buf = [0.0] * n
        self.mask = n - 1
        self.wp = 0

    def process(self, x, feedback, delay):
        read_pos = float(self.wp) - float(F(delay))
        fl = math.floor(read_pos)
        frac = read_pos - fl
        i1 = int(fl)
        v1 = self.buf[i1 & self.mask]
        v2 = self.buf[(i1 + 1) & self.mask]
        out = F(v1 + (v2 - v1) * frac)
        self.buf[self.wp & self.mask] = float(F(x)) + float(out) * float(F(feedback))
        self.wp = (self.wp + 1) & self.mask
        return out


class LookupDelay:  # LookupDelayLine<2>
    def __init__(self, sr, delay_time):
        self.delay = int(math.ceil(float(F(F(delay_time) * F(sr)))))
        n = next_pow2(self.delay) if self.delay > 0 else 0
        self.buf = [[0.0, 0.0] for _ in range(n)]
        self.mask = n - 1 if n else 0
        self.wp = 0
        self.peak = 0.0
        self.peak_pos = 0

    def process(self, l, r):
        if self.delay == 0:
            return F(l), F(r)
        n = len(self.buf)
        ri = (self.wp + n - self.delay) & self.mask
        d = (F(self.buf[ri][0]), F(self.buf[ri][1]))
        self.buf[self.wp & self.mask] = [float(F(l)), float(F(r))]
        expired = self.peak_pos == ri
        new_peak = max(max(0.0, float(fabs(l))), float(fabs(r)))
        if new_peak >= self.peak:
            self.peak = new_peak
            self.peak_pos = self.wp
        elif expired:
            self.peak = 0.0
            for i in range(self.delay):
                fi = (self.wp + n - i) & self.mask
                fp = max(max(0.0, abs(self.buf[fi][0])), abs(self.buf[fi][1]))
                if fp >= self.peak:
                    self.peak = fp
                    self.peak_pos = fi
        self.wp = (self.wp + 1) & self.mask
        return d

    def peak_value(self):
        return F(self.peak)


def db_to_linear(v):
    v = F(v)
    k = F(F(2.30258509299404568402) / F(20.0))  # LN_10 / 20 in f32
    if v != v:
        return F(np.nan)
    if v == F(0.0):
        return F(1.0)
    if v > F(-200.0):
        return expf(F(v * k))
    return F(0.0)


def panning_factors(p):
    power = F(0.707106781186547524400844362104849039)
    n = F(F(fclamp(p, F(-1.0), F(1.0)) + F(1.0)) / F(2.0))
    return F(np.sqrt(F(F(1.0) - n)) / power), F(np.sqrt(n) / power)


# ---- effects (stereo, interleaved f32 buffers processed in place; `set` = process_parameter_update with a Raw value) -------------------------
class Gain:
    def __init__(self, sr, gain=1.0, dc_mode=0):
        self.sr = sr
        self.gain = ExpSm(gain, sr)
        self.mode = dc_mode
        hz = {1: 1.0, 2: 5.0, 3: 20.0}.get(dc_mode, 5.0)
        self.dc = [Dc(sr, hz), Dc(sr, hz)]

    def set(self, pid, v):
        if pid == "gain":
            self.gain.set_target(fclamp(v, F(0.000001), F(15.848932)))
        else:
            self.mode = int(v)
            if self.mode:
                for f in self.dc:
                    f.set(self.sr, {1: 1.0, 2: 5.0, 3: 20.0}[self.mode])
            else:
                for f in self.dc:
                    f.reset()

    def process(self, b):
        n = len(b) // 2
        if self.mode != 0:
            for ch in range(2):
                for i in range(n):
                    b[2 * i + ch] = F(self.dc[ch].tick(float(b[2 * i + ch])))
        if self.gain.need_ramp():
            for i in range(n):
                g = self.gain.next()
                b[2 * i] = F(b[2 * i] * g)
                b[2 * i + 1] = F(b[2 * i + 1] * g)
        else:
            g = self.gain.target
            for i in range(2 * n):
                b[i] = F(b[i] * g)


class Panning:
    def __init__(self, sr, pan=0.0, width=1.0, invl=0, invr=0):
        self.pan, self.width = ExpSm(pan, sr), ExpSm(width, sr)
        self.invl, self.invr = bool(invl), bool(invr)

    def set(self, pid, v):
        if pid == "pan ":
            self.pan.set_target(fclamp(v, F(-1.0), F(1.0)))
        elif pid == "wdth":
            self.width.set_target(fclamp(v, F(0.0), F(2.0)))
        elif pid == "invl":
            self.invl = bool(v)
        else:
            self.invr = bool(v)

    def process(self, b):
        il = F(-1.0) if self.invl else F(1.0)
        ir = F(-1.0) if self.invr else F(1.0)
        has_inv = il < 0.0 or ir < 0.0
        pr, wr = self.pan.need_ramp(), self.width.need_ramp()
        if not has_inv and not pr and not wr and fabs(self.pan.target) < F(1e-6) and fabs(F(self.width.target - F(1.0))) < F(1e-6):
            return
        for i in range(len(b) // 2):
            l, r = F(b[2 * i] * il), F(b[2 * i + 1] * ir)
            w = self.width.next() if wr else self.width.target
            if fabs(F(w - F(1.0))) > F(1e-6):
                mid = F(F(l + r) * F(0.5))
                side = F(F(l - r) * F(0.5))
                l = F(mid + F(side * w))
                r = F(mid - F(side * w))
            p = self.pan.next() if pr else self.pan.target
            if fabs(p) > F(1e-6):
                pl, prr = panning_factors(p)
                l = F(l * pl)
                r = F(r * prr)
            b[2 * i], b[2 * i + 1] = l, r


FILTER_TO_BIQUAD = {0: LOWPASS, 1: BANDPASS, 2: NOTCH, 3: HIGHPASS}


class Filter:
    def __init__(self, sr, params=None):
        self.sr = sr
        self.coefs = BiquadCoefs()
        self.coefs.set(LOWPASS, 44100, F(22050.0), F(0.707), F(0.0))  # FilterEffect::new(): coefficients for 44100 Hz
        self.ftype = 0
        cutoff, q = F(20000.0), F(0.707)
        if params:  # with_parameters
            self.ftype, cutoff, q = int(params["type"]), F(params["cuto"]), F(params["fltq"])
            self.coefs.set(FILTER_TO_BIQUAD[self.ftype], 44100, fclamp(cutoff, F(20.0), F(44100.0) / F(2.0)), q, F(0.0))
        self.cutoff, self.q = ExpSm(cutoff, sr), LinSm(q, sr, 0.01)
        self.coefs.set_cutoff(fclamp(self.coefs.cutoff(), F(20.0), F(sr) / F(2.0)))  # initialize(): only the cutoff is re-validated
        self.f = [Biquad(), Biquad()]

    def set(self, pid, v):
        if pid == "type":
            self.ftype = int(v)
            self.coefs.set_type(FILTER_TO_BIQUAD[self.ftype])
        elif pid == "cuto":
            self.cutoff.set_target(fclamp(v, F(20.0), F(20000.0)))
        else:
            self.q.set_target(fclamp(v, F(0.001), F(4.0)))

    def process(self, b):
        n = len(b) // 2
        if self.cutoff.need_ramp() or self.q.need_ramp():
            for i in range(n):
                c = fclamp(self.cutoff.next(), F(20.0), F(self.sr) / F(2.0))
                q = self.q.next()
                self.coefs.set(FILTER_TO_BIQUAD[self.ftype], self.sr, c, q, F(0.0))
                for ch in range(2):
                    b[2 * i + ch] = F(self.f[ch].tick(self.coefs, float(b[2 * i + ch])))
        else:
            for ch in range(2):
                for i in range(n):
                    b[2 * i + ch] = F(self.f[ch].tick(self.coefs, float(b[2 * i + ch])))


class Eq5:
    DEF_F = (100.0, 1000.0, 4000.0, 8000.0, 12000.0)
    BW_MAX = (1.0, 4.0, 4.0, 4.0, 1.0)
    TYPES = (LOWSHELF, BELL, BELL, BELL, HIGHSHELF)

    def __init__(self, sr, params=None):
        params = params or {}
        self.sr = sr
        self.gains = [ExpSm(params.get(f"gan{i+1}", 0.0), sr) for i in range(5)]
        self.freqs = [ExpSm(params.get(f"frq{i+1}", self.DEF_F[i]), sr) for i in range(5)]
        self.bws = [LinSm(params.get(f"bw_{i+1}", self.BW_MAX[i]), sr, 0.01) for i in range(5)]
        self.coefs = [BiquadCoefs() for _ in range(5)]
        self.update()
        self.f = [[Biquad() for _ in range(5)] for _ in range(2)]

    def update(self):  # update_filter_coefficients: q = bandwidth
        for i in range(5):
            self.coefs[i].set(self.TYPES[i], self.sr, fclamp(self.freqs[i].current, F(20.0), F(self.sr) / F(2.0)), self.bws[i].current, self.gains[i].current)

    def set(self, pid, v):
        i = int(pid[3]) - 1
        if pid.startswith("gan"):
            self.gains[i].set_target(fclamp(v, F(-20.0), F(20.0)))
        elif pid.startswith("frq"):
            self.freqs[i].set_target(fclamp(v, F(20.0), F(20000.0)))
        else:
            self.bws[i].set_target(fclamp(v, F(0.0001), F(self.BW_MAX[i])))
        self.update()

    def process(self, b):
        need = any(s.need_ramp() for s in self.freqs) or any(s.need_ramp() for s in self.bws) or any(s.need_ramp() for s in self.gains)
        for n in range(len(b) // 2):
            if need:  # ramp_filter_coefficients: q = 1 / max(bandwidth, 0.001) for the bell bands
                for i in range(5):
                    bw = self.bws[i].next()
                    q = bw if i in (0, 4) else F(F(1.0) / fmax(bw, F(0.001)))
                    c = fclamp(self.freqs[i].next(), F(20.0), F(self.sr) / F(2.0))
                    g = self.gains[i].next()
                    self.coefs[i].set(self.TYPES[i], self.sr, c, q, g)
            for ch in range(2):
                s = b[2 * n + ch]
                for i in range(5):
                    s = F(self.f[ch][i].tick(self.coefs[i], float(s)))
                b[2 * n + ch] = s


def saturate(x, drive):
    if drive < F(0.001):
        return x
    gain = 1.0 + float(drive) * 4.0
    x = x * gain
    x2 = x * x
    return (x * (27.0 + x2) / (27.0 + 9.0 * x2)) / math.sqrt(gain)


class Delay:
    DEF = {"mode": 0, "dlay": 375.0, "fdbk": 0.5, "ftyp": 0, "cuto": 6000.0, "driv": 0.0, "wet_": 0.5, "wdth": 0.5, "lfor": 1.0, "lfos": 0, "lfdt": 0.0, "ldfb": 0.0, "lfdf": 0.0}
    RANGE = {"dlay": (1.0, 4000.0), "fdbk": (0.0, 1.0), "cuto": (20.0, 20000.0), "driv": (0.0, 1.0), "wet_": (0.0, 1.0), "wdth": (0.0, 1.0), "lfor": (0.01, 10.0),
             "lfdt": (-1.0, 1.0), "ldfb": (-1.0, 1.0), "lfdf": (-1.0, 1.0)}

    def __init__(self, sr, params=None):
        p = dict(self.DEF, **(params or {}))
        self.sr = sr
        self.mode, self.ftype, self.shape = int(p["mode"]), int(p["ftyp"]), int(p["lfos"])
        self.sm = {k: ExpSm(p[k], sr) for k in ("fdbk", "cuto", "driv", "wet_", "wdth", "lfor", "lfdt", "ldfb", "lfdf")}
        self.sm["dlay"] = SpringSm(p["dlay"], sr, 20000)
        max_delay = int(math.ceil(float(F(F(F(4000.0) + F(50.0)) * F(sr)) / F(1000.0))))
        self.lines = [InterpDelay(max_delay + 4), InterpDelay(max_delay + 4)]
        self.coefs = SvfCoefs(self.ftype, sr, fclamp(self.sm["cuto"].target, F(20.0), F(sr) / F(2.0)), F(0.302))
        self.lfo = Lfo(sr, float(self.sm["lfor"].target), self.shape)
        self.flt = [Svf(), Svf()]
        self.dc = [Dc(sr, 5.0), Dc(sr, 5.0)]
        self.fb = [F(0.0), F(0.0)]

    def set(self, pid, v):
        if pid == "mode":
            self.mode = int(v)
        elif pid == "ftyp":
            self.ftype = int(v)
        elif pid == "lfos":
            self.shape = int(v)
            self.lfo.wf = self.shape
        else:
            lo, hi = self.RANGE[pid]
            self.sm[pid].set_target(fclamp(v, F(lo), F(hi)))

    def feedback_path(self, ch, delayed, drive):
        filtered = self.flt[ch].tick(self.coefs, float(delayed))
        clean = F(self.dc[ch].tick(saturate(filtered, drive)))
        return fclamp(clean, F(-4.0), F(4.0))

    def process(self, b):
        srf = F(self.sr)
        s = self.sm
        for i in range(len(b) // 2):
            li, ri = b[2 * i], b[2 * i + 1]
            lfo = self.lfo.run()
            if s["lfor"].need_ramp():
                self.lfo.set_rate(self.sr, float(s["lfor"].next()))
            base = s["dlay"].next()
            tmod = F(F(lfo * s["lfdt"].next()) * F(50.0))
            delay_ms = fmax(F(base + tmod), F(1.0))
            dsamp = F(F(delay_ms * F(0.001)) * srf)
            fdepth = s["lfdf"].next()
            fmod = powf(F(2.0), F(F(lfo * fdepth) * F(2.0)))
            cutoff = fclamp(F(s["cuto"].next() * fmod), F(20.0), srf / F(2.0))
            self.coefs.set(self.ftype, self.sr, cutoff, F(0.302))
            bfb = s["fdbk"].next()
            fbd = s["ldfb"].next()
            fb = fclamp(F(bfb + F(F(lfo * fbd) * F(F(1.0) - fabs(bfb)))), F(0.0), F(0.999))
            drive = s["driv"].next()
            wet = s["wet_"].next()
            width = s["wdth"].next()
            if self.mode == 0:
                l_in = F(li + F(self.fb[0] * fb))
                cl = self.feedback_path(0, self.lines[0].process(l_in, F(0.0), dsamp), drive)
                self.fb[0] = cl
                r_in = F(ri + F(self.fb[1] * fb))
                cr = self.feedback_path(1, self.lines[1].process(r_in, F(0.0), dsamp), drive)
                self.fb[1] = cr
            else:
                mono = F(F(li + ri) * F(0.5))
                l_in = F(mono + F(self.fb[1] * fb))
                cl = self.feedback_path(0, self.lines[0].process(l_in, F(0.0), dsamp), drive)
                r_in = F(self.fb[0] * fb)
                cr = self.feedback_path(1, self.lines[1].process(r_in, F(0.0), dsamp), drive)
                self.fb = [cl, cr]
            dry_g = fmin(F(F(F(1.0) - wet) * F(2.0)), F(1.0))
            wet_g = fmin(F(wet * F(2.0)), F(1.0))
            ol = F(F(li * dry_g) + F(cl * wet_g))
            orr = F(F(ri * dry_g) + F(cr * wet_g))
            mid = F(F(ol + orr) * F(0.5))
            side = F(F(ol - orr) * F(0.5))
            b[2 * i] = F(mid + F(side * width))
            b[2 * i + 1] = F(mid - F(side * width))


class Chorus:
    DEF = {"rate": 1.0, "dpth": 0.25, "fdbk": 0.5, "dlay": 12.0, "wet_": 0.5, "phas": float(PI32 / F(2.0)), "fltt": 0, "fltf": 20000.0, "fltq": 0.0}
    RANGE = {"rate": (0.01, 10.0), "dpth": (0.0, 1.0), "fdbk": (-1.0, 1.0), "dlay": (0.0, 100.0), "wet_": (0.0, 1.0), "phas": (0.0, float(PI32)), "fltf": (20.0, 20000.0),
             "fltq": (0.0, 1.0)}

    def __init__(self, sr, params=None):
        p = dict(self.DEF, **(params or {}))
        self.sr = sr
        self.ftype = int(p["fltt"])
        self.sm = {k: ExpSm(p[k], sr) for k in ("dpth", "fdbk", "wet_", "fltf", "fltq")}
        self.sm["rate"] = LinSm(p["rate"], sr, 0.005)
        self.sm["phas"] = LinSm(p["phas"], sr, 0.001)
        self.sm["dlay"] = SpringSm(p["dlay"], sr, 1000)
        self.lfo_range = F(F(256.0) * F(F(sr) / F(44100.0)))
        max_depth = int(math.ceil(float(self.lfo_range)))
        max_delay = int(math.ceil(float(F(F(F(100.0) * F(sr)) / F(1000.0)))))
        size = 2 + max_delay + 2 * max_depth + 1
        self.lines = [InterpDelay(size), InterpDelay(size)]
        self.coefs = SvfCoefs(self.ftype, sr, fclamp(self.sm["fltf"].target, F(20.0), F(sr) / F(2.0)), self.sm["fltq"].target)
        self.flt = [Svf(), Svf()]
        self.current_phase = 0.0
        self.reset_lfos()

    def reset_lfos(self):
        rate = float(self.sm["rate"].current)
        self.osc = [Lfo(self.sr, rate, 0), Lfo(self.sr, rate, 0)]
        off = float(self.sm["phas"].current)
        self.osc[0].set_phase_degrees(F(self.current_phase))
        self.osc[1].set_phase_degrees(F(self.current_phase + off))

    def update_lfos(self):
        rate = float(self.sm["rate"].next())
        for o in self.osc:
            o.set_rate(self.sr, rate)
        off = float(self.sm["phas"].next())
        self.osc[0].set_phase_degrees(F(self.current_phase))
        self.osc[1].set_phase_degrees(F(self.current_phase + off))

    def set(self, pid, v):
        if pid == "fltt":
            self.ftype = int(v)
            self.coefs.set(self.ftype, self.coefs.key[1], self.coefs.key[2], self.coefs.key[3])
        else:
            lo, hi = self.RANGE[pid]
            self.sm[pid].set_target(fclamp(v, F(lo), F(hi)))

    def process(self, b):
        s = self.sm
        srf = F(self.sr)
        n = len(b) // 2
        for i in range(n):
            li, ri = b[2 * i], b[2 * i + 1]
            delay_ms = s["dlay"].next()
            depth = s["dpth"].next()
            fb = fclamp(s["fdbk"].next(), F(-0.999), F(0.999))
            wet = s["wet_"].next()
            dry = F(F(1.0) - wet)
            if s["rate"].need_ramp() or s["phas"].need_ramp():
                self.update_lfos()
            if s["fltf"].need_ramp() or s["fltq"].need_ramp():
                c = fclamp(s["fltf"].next(), F(20.0), srf / F(2.0))
                r = s["fltq"].next()
                self.coefs.set(self.ftype, self.sr, c, r)
            fl = self.flt[0].tick(self.coefs, float(li))
            fr = self.flt[1].tick(self.coefs, float(ri))
            dsamp = F(F(delay_ms * srf) * F(0.001))
            depth_s = F(self.lfo_range * depth)
            ll, rl = self.osc[0].run(), self.osc[1].run()
            lpos = F(F(F(2.0) + dsamp) + F(F(F(1.0) + ll) * depth_s))
            rpos = F(F(F(2.0) + dsamp) + F(F(F(1.0) + rl) * depth_s))
            lo = self.lines[0].process(F(fl), fb, lpos)
            ro = self.lines[1].process(F(fr), fb, rpos)
            b[2 * i] = F(F(li * dry) + F(lo * wet))
            b[2 * i + 1] = F(F(ri * dry) + F(ro * wet))
        inc = 2.0 * math.pi * float(s["rate"].current) / float(self.sr)
        self.current_phase += float(2 * n) / 2.0 * inc
        while self.current_phase >= 2.0 * math.pi:
            self.current_phase -= 2.0 * math.pi


class Compressor:
    DEF = {"thrs": -12.0, "rato": 8.0, "knee": 3.0, "attk": 0.02, "rels": 2.0, "gain": 6.0, "look": 0.04}

    def __init__(self, sr, params=None):
        p = dict(self.DEF, **(params or {}))
        self.sr = sr
        self.p = {k: F(v) for k, v in p.items() if k != "gain"}
        self.makeup = ExpSm(p["gain"], sr)
        self.line = LookupDelay(sr, self.p["look"])
        self.env = Envelope(sr, float(self.p["attk"]), float(self.p["rels"]))
        self.env.cur = F(-120.0) if self.p["rato"] >= F(20.0) else F(0.0)

    def set(self, pid, v):
        rng = {"thrs": (-60.0, 0.0), "rato": (1.0, 20.0), "knee": (0.0, 12.0), "attk": (0.001, 0.5), "rels": (0.1, 2.0), "look": (0.001, 0.2), "gain": (-24.0, 24.0)}[pid]
        v = fclamp(v, F(rng[0]), F(rng[1]))
        if pid == "gain":
            self.makeup.set_target(v)
            self.env.set_times(float(self.p["attk"]), float(self.p["rels"]))
            return
        old_look = self.p["look"]
        self.p[pid] = v
        self.env.set_times(float(self.p["attk"]), float(self.p["rels"]))
        if pid == "look" and v != old_look:
            self.line = LookupDelay(self.sr, v)

    def process(self, b):
        p = self.p
        inp = b.copy()
        for i in range(len(b) // 2):
            il, ir = inp[2 * i], inp[2 * i + 1]
            dl, dr = self.line.process(il, ir)
            if p["rato"] >= F(20.0):
                pk = self.line.peak_value()
                in_db = F(F(20.0) * log10f(pk)) if pk > F(1e-6) else F(-120.0)
            else:
                pk = fmax(fabs(il), fabs(ir))
                in_db = F(F(20.0) * log10f(pk)) if pk > F(1e-6) else F(-120.0)
            env = self.env.run(in_db)
            t, w = p["thrs"], p["knee"]
            slope = F(1.0) if p["rato"] >= F(20.0) else F(F(1.0) - F(F(1.0) / p["rato"]))
            half = F(w / F(2.0))
            if w > F(0.0) and env > F(t - half) and env < F(t + half):
                x = F(F(env - F(t - half)) / w)
                gr = F(F(F(F(x * x) * slope) * w) / F(2.0))
            elif env > F(t + half):
                gr = F(F(env - t) * slope)
            else:
                gr = F(0.0)
            mk = self.makeup.next()
            g = db_to_linear(F(mk - gr))
            b[2 * i], b[2 * i + 1] = F(dl * g), F(dr * g)


class Gate:
    DEF = {"thrs": -30.0, "attk": 0.005, "hold": 0.1, "rels": 0.2, "rnge": -60.0}

    def __init__(self, sr, params=None):
        self.sr = sr
        self.p = {k: F(v) for k, v in dict(self.DEF, **(params or {})).items()}
        self.env = Envelope(sr, float(self.p["attk"]), float(self.p["rels"]))
        self.env.cur = F(-120.0)
        self.hold = 0
        self.gain_db = self.p["rnge"]
        self.update()

    def update(self):
        self.env.set_times(float(self.p["attk"]), float(self.p["rels"]))
        sr = F(self.sr)
        self.att = expf(F(F(-1.0) / F(self.p["attk"] * sr)))
        self.rel = expf(F(F(-1.0) / F(self.p["rels"] * sr)))

    def set(self, pid, v):
        rng = {"thrs": (-60.0, 0.0), "attk": (0.001, 0.5), "hold": (0.0, 2.0), "rels": (0.01, 2.0), "rnge": (-60.0, 0.0)}[pid]
        self.p[pid] = fclamp(v, F(rng[0]), F(rng[1]))
        self.update()

    def process(self, b):
        p = self.p
        hold_samples = int(float(F(p["hold"] * F(self.sr))))
        for i in range(len(b) // 2):
            pk = fmax(fabs(b[2 * i]), fabs(b[2 * i + 1]))
            in_db = F(F(20.0) * log10f(pk)) if pk > F(1e-6) else F(-120.0)
            env = self.env.run(in_db)
            if env >= p["thrs"]:
                self.hold = hold_samples
                tgt = F(0.0)
            elif self.hold > 0:
                self.hold -= 1
                tgt = F(0.0)
            else:
                tgt = p["rnge"]
            c = self.att if tgt > self.gain_db else self.rel
            self.gain_db = F(F(c * self.gain_db) + F(F(F(1.0) - c) * tgt))
            g = F(0.0) if self.gain_db <= F(-60.0) else db_to_linear(self.gain_db)
            b[2 * i] = F(b[2 * i] * g)
            b[2 * i + 1] = F(b[2 * i + 1] * g)


MAX_DRIVE = F(4.0)


def shape(kind, x, drive):
    x, drive = F(x), F(drive)
    t = F(drive / MAX_DRIVE)
    if kind == 0:  # soft_clip
        gain = F(F(1.0) + F(F(t * t) * F(14.0)))
        y = F(x * gain)
        if y >= F(1.0):
            return F(1.0)
        if y > F(-1.0):
            if gain <= F(1.0):
                return x
            return F(F(F(3.0) / F(2.0)) * F(y - F(F(F(y * y) * y) / F(3.0))))
        return F(-1.0)
    if kind == 1:  # hard_clip
        gain = F(F(1.0) + F(F(t * t) * F(24.0)))
        th = F(F(1.0) / gain)
        return F(fclamp(x, F(-th), th) * gain)
    if kind == 2:  # diode
        curve = F(F(F(0.6) * F(t * t)) + F(F(0.4) * t))
        gain = F(F(1.0) + F(curve * F(19.0)))
        dc = F(expf(F(F(F(0.1) * x) / F(F(0.0253) * F(1.68)))) - F(1.0))
        return F(F(F(2.0) / PI32) * atanf(F(dc * gain)))
    if kind == 3:  # fuzz
        gain = F(F(1.0) + F(F(F(1.0) - expf(F(F(-3.0) * t))) * F(29.0)))
        a = F(x * gain)
        e = F(F(1.0) - expf(F(-fabs(a))))
        s = F(F(-1.0) * e) if a < F(0.0) else F(F(1.0) * e)
        return F(F(1.5) * F(s + fabs(s)))
    gain = F(F(1.0) + F(F(t * t) * F(3.0)))  # fold
    y = F(x * gain)
    th = F(F(1.0) / gain)
    if y > th or y < F(-th):
        return F(fabs(F(fmodf(fabs(F(y - th)), F(th * F(4.0))) - F(th * F(2.0)))) - th)
    return y


_LUTS = None


def dist_luts():
    global _LUTS
    if _LUTS is None:
        partials = ((1.0, 0.60), (2.7, 0.25), (5.3, 0.10), (9.1, 0.03), (14.6, 0.02))
        peak = F(0.0)
        for _, a in partials:
            peak = F(peak + F(a))
        _LUTS = np.zeros((5, 256), F)
        for kind in range(5):
            for li in range(256):
                drive = F(F(F(li) / F(255.0)) * MAX_DRIVE)
                in_sq = out_sq = F(0.0)
                for i in range(256):
                    t = F(F(TAU32 * F(F(i) + F(0.5))) / F(256.0))
                    s = F(0.0)
                    for fq, a in partials:
                        s = F(s + F(F(a) * sinf(F(F(fq) * t))))
                    smp = F(s / peak)
                    in_sq = F(in_sq + F(smp * smp))
                    o = shape(kind, smp, drive)
                    out_sq = F(out_sq + F(o * o))
                irms = F(np.sqrt(F(in_sq / F(256.0))))
                orms = F(np.sqrt(F(out_sq / F(256.0))))
                _LUTS[kind, li] = F(irms / orms) if orms > F(1e-10) else F(1.0)
    return _LUTS


class Distortion:
    def __init__(self, sr, params=None):
        p = dict({"type": 2, "driv": 0.0, "mix ": 1.0}, **(params or {}))
        self.kind = int(p["type"])
        self.drive = LinSm(p["driv"], sr, 0.01)
        self.mix = ExpSm(p["mix "], sr, inertia=0.1)
        self.luts = dist_luts()

    def set(self, pid, v):
        if pid == "type":
            self.kind = int(v)
        elif pid == "driv":
            self.drive.set_target(fclamp(v, F(0.0), F(4.0)))
        else:
            self.mix.set_target(fclamp(v, F(0.0), F(1.0)))

    def comp(self, drive):
        pos = F(fclamp(F(drive / MAX_DRIVE), F(0.0), F(1.0)) * F(255.0))
        lo = int(float(pos))
        hi = min(lo + 1, 255)
        frac = F(pos - F(lo))
        lut = self.luts[self.kind]
        return F(lut[lo] + F(F(lut[hi] - lut[lo]) * frac))

    def process(self, b):
        n = len(b) // 2
        if not self.mix.need_ramp() and self.mix.target == F(0.0):
            return
        if not self.mix.need_ramp() and self.mix.target >= F(1.0):
            if not self.drive.need_ramp():
                d = self.drive.target
                c = self.comp(d)
                for i in range(2 * n):
                    b[i] = F(shape(self.kind, b[i], d) * c)
            else:
                for i in range(n):
                    d = self.drive.next()
                    c = self.comp(d)
                    for ch in range(2):
                        b[2 * i + ch] = F(shape(self.kind, b[2 * i + ch], d) * c)
        else:
            for i in range(n):
                d = self.drive.next()
                c = self.comp(d)
                m = self.mix.next()
                for ch in range(2):
                    dry = b[2 * i + ch]
                    wet = F(shape(self.kind, dry, d) * c)
                    b[2 * i + ch] = F(F(F(F(1.0) - m) * dry) + F(m * wet))


# ---- cases: (name, effect kind index of the ABI, class, construction parameters, {block: [(id, raw value)]}, signal) --------------------------------
CASES = [
    ("gain_default", 0, Gain, None, {}, "noise"),
    ("gain_dc_ramp", 0, Gain, {"gain": 0.5, "dcfm": 2}, {1: [("gain", 1.7)], 2: [("dcfm", 3)]}, "noise"),
    ("pan_default", 1, Panning, None, {}, "noise"),
    ("pan_ramp", 1, Panning, {"pan ": -0.3, "wdth": 1.5, "invr": 1}, {1: [("pan ", 0.8), ("wdth", 0.2)], 2: [("invl", 1)]}, "noise"),
    ("filter_default", 2, Filter, None, {}, "noise"),
    ("filter_ramp", 2, Filter, {"type": 0, "cuto": 2000.0, "fltq": 0.707}, {1: [("cuto", 600.0), ("fltq", 2.5)], 2: [("type", 3)]}, "noise"),
    ("eq5_default", 3, Eq5, None, {}, "noise"),
    ("eq5_ramp", 3, Eq5, {"gan1": 6.0, "gan3": -4.0, "bw_2": 1.5}, {1: [("gan2", 9.0), ("frq2", 500.0)], 2: [("bw_3", 0.7), ("gan5", -12.0)]}, "noise"),
    ("delay_default", 4, Delay, None, {}, "noise"),
    ("delay_ramp", 4, Delay, {"mode": 1, "dlay": 20.0, "fdbk": 0.7, "driv": 0.5, "ftyp": 2, "lfdt": 0.1, "lfdf": 0.4, "lfor": 5.0, "lfos": 1},
     {1: [("dlay", 45.0), ("fdbk", 0.3)], 2: [("cuto", 900.0), ("lfor", 2.0), ("ldfb", 0.5)]}, "noise"),
    ("delay_lfo_rampup", 4, Delay, {"dlay": 8.0, "lfdt": 0.05, "ldfb": 0.3, "lfdf": -0.5, "lfor": 9.0, "lfos": 2}, {2: [("wdth", 1.0), ("wet_", 0.9)]}, "noise"),
    ("delay_lfo_rampdown", 4, Delay, {"dlay": 11.0, "lfdt": -0.08, "lfdf": 0.7, "lfor": 7.0, "lfos": 3, "ftyp": 1}, {1: [("driv", 0.8)]}, "noise"),
    ("delay_lfo_square", 4, Delay, {"mode": 1, "dlay": 6.0, "lfdt": 0.03, "ldfb": -0.4, "lfor": 10.0, "lfos": 4}, {1: [("mode", 0)]}, "noise"),
    ("chorus_default", 6, Chorus, None, {}, "noise"),
    ("chorus_ramp", 6, Chorus, {"rate": 3.0, "dpth": 0.8, "fdbk": -0.6, "dlay": 0.5, "fltt": 1, "fltf": 300.0, "fltq": 0.4},
     {1: [("rate", 5.0), ("phas", 1.0)], 2: [("dlay", 30.0), ("fltf", 2000.0), ("dpth", 0.3)]}, "noise"),
    ("comp_default", 7, Compressor, None, {}, "burst"),
    ("limiter_ramp", 7, Compressor, {"thrs": -0.01, "rato": 20.0, "knee": 0.0, "attk": 0.02, "rels": 2.0, "gain": 0.0, "look": 0.02},
     {1: [("gain", -6.0), ("thrs", -20.0)], 2: [("look", 0.01), ("rato", 4.0)]}, "burst"),
    ("gate_default", 8, Gate, None, {}, "burst"),
    ("gate_ramp", 8, Gate, {"thrs": -20.0, "attk": 0.002, "hold": 0.01, "rels": 0.05, "rnge": -40.0}, {1: [("thrs", -10.0)], 2: [("hold", 0.0), ("rnge", -20.0)]}, "burst"),
    ("dist_default", 9, Distortion, None, {}, "noise"),
    ("dist_ramp", 9, Distortion, {"type": 0, "driv": 2.0}, {1: [("driv", 3.5), ("mix ", 0.3)], 2: [("type", 4)]}, "noise"),
]
BLOCKS, FRAMES, SR = 3, 300, 48000


def signal(kind, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(BLOCKS * FRAMES * 2) * (0.25 if kind == "noise" else 0.5)
    if kind == "burst":
        x = x * np.repeat((np.arange(BLOCKS * FRAMES) % 250 < 90).astype(np.float64), 2)
    return x.astype(F)


def run_case(case):
    name, kind, cls, params, updates, sig = case
    x = signal(sig, 100 + kind)
    fx = cls(SR, **params) if cls in (Gain, Panning) and params else (cls(SR) if cls in (Gain, Panning) else cls(SR, params))
    y = x.copy()
    for blk in range(BLOCKS):
        for pid, v in updates.get(blk, []):
            fx.set(pid, v)
        fx.process(y[blk * FRAMES * 2:(blk + 1) * FRAMES * 2])
    return x, y


def _gain_pan_kwargs(cls, params):
    return params


# Gain / Panning take keyword arguments named differently from the FourCCs
def _adapt(case):
    name, kind, cls, params, updates, sig = case
    if cls is Gain and params:
        params = {"gain": params.get("gain", 1.0), "dc_mode": params.get("dcfm", 0)}
    if cls is Panning and params:
        params = {"pan": params.get("pan ", 0.0), "width": params.get("wdth", 1.0), "invl": params.get("invl", 0), "invr": params.get("invr", 0)}
    return (name, kind, cls, params, updates, sig)


def make_vectors():
    v = {}
    for case in CASES:
        x, y = run_case(_adapt(case))
        v[case[0] + "_in"] = x
        v[case[0] + "_out"] = y
    return v


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    vec = make_vectors()
    np.savez_compressed(os.path.join(here, "independent_fx.npz"), **vec)
    print("independent_fx.npz", os.path.getsize(os.path.join(here, "independent_fx.npz")), "bytes")
