#!/usr/bin/env python3
"""A second, independent restatement of two pieces of the reference — written from the Rust sources in plain Python scalars
(`float` = f64, `numpy.float32` = f32), separately from the C++ oracle — used to cross-check the oracle where the reference
holds no known-answer tests:

  * CubicInterpolator / CubicResampler          src/utils/resampler/cubic.rs:36-142,179-186
  * ReverbEffect in steady state (room/wet constant) with its DelayLine, AllpassDelayLine, ReverbDelayLine and the
    TPT-SVF low-pass biquad                      src/effect/reverb.rs:196-369,429-447,554-604; src/utils/dsp/delay.rs:47-66,
                                                 314-350; src/utils/dsp/filters/biquad.rs:175-183,314-322

Run as a script it writes tests/golden/independent.npz (inputs + this file's outputs); tests/test_golden.py compares the C++
oracle with those vectors. Both sides call the same libm (`math.sin` / `std::sin`), so the comparison is bit for bit.
"""
import math
import os
import sys

import numpy as np

f32 = np.float32


# ---- cubic resampler --------------------------------------------------------------------------------------------------------
class CubicInterpolator:
    def __init__(self, ratio):
        self.ratio = f32(ratio)
        self.input = [f32(0.0)] * 4
        self.sub_pos = f32(0.0)
        self.initialized = False

    def push(self, v):
        self.input = [f32(v), self.input[0], self.input[1], self.input[2]]

    def interpolate(self, fraction):
        ym1, y0, y1, y2 = self.input[3], self.input[2], self.input[1], self.input[0]
        fraction = f32(fraction)
        c0 = y0
        c1 = f32(f32(y1 - ym1) * f32(0.5))
        c2 = f32(f32(f32(ym1 - f32(y0 * f32(2.5))) + f32(y1 * f32(2.0))) - f32(y2 * f32(0.5)))
        c3 = f32(f32(f32(y2 - ym1) * f32(0.5)) + f32(f32(y0 - y1) * f32(1.5)))
        return f32(f32(f32(f32(f32(f32(c3 * fraction) + c2) * fraction) + c1) * fraction) + c0)

    def process(self, inp, out, ch, nch):
        num_in, num_out = len(inp) // nch, len(out) // nch
        consumed = produced = 0
        if abs(float(f32(self.ratio - f32(1.0)))) < 0.000001:
            m = min(len(inp), len(out))
            out[:m] = inp[:m]
            return m, m
        if not self.initialized and num_in >= 3:
            self.initialized = True
            for f in range(3):
                self.push(inp[f * nch + ch])
                consumed += 1
        one = f32(1.0)
        if self.ratio < one:
            while produced < num_out:
                if self.sub_pos >= one:
                    if consumed >= num_in:
                        break
                    self.push(inp[consumed * nch + ch])
                    consumed += 1
                    self.sub_pos = f32(self.sub_pos - one)
                out[produced * nch + ch] = self.interpolate(self.sub_pos)
                produced += 1
                self.sub_pos = f32(self.sub_pos + self.ratio)
        else:
            done = False
            while produced < num_out and not done:
                while self.sub_pos < self.ratio:
                    if consumed >= num_in:
                        done = True
                        break
                    self.push(inp[consumed * nch + ch])
                    consumed += 1
                    self.sub_pos = f32(self.sub_pos + one)
                if done:
                    break
                self.sub_pos = f32(self.sub_pos - self.ratio)
                out[produced * nch + ch] = self.interpolate(f32(one - self.sub_pos))
                produced += 1
        return consumed * nch, produced * nch


def cubic_resample(inp, in_rate, out_rate, nch, out_len, out_chunk):
    """Repeated CubicResampler::process calls with `out_chunk`-sample outputs (the shape of oracle.po_cubic_resample)."""
    ratio = f32(float(in_rate) / float(out_rate))  # ResamplingSpecs::input_ratio() as f32
    chans = [CubicInterpolator(ratio) for _ in range(nch)]
    out = np.zeros(out_len, f32)
    consumed = produced = 0
    while produced < out_len:
        want = min(out_chunk, out_len - produced)
        res = (0, 0)
        for ch, it in enumerate(chans):
            res = it.process(inp[consumed:], out[produced:produced + want], ch, nch)
        consumed += res[0]
        produced += res[1]
        if res[1] == 0:
            break
    return out[:produced], consumed


# ---- reverb -----------------------------------------------------------------------------------------------------------------
class Biquad:
    def __init__(self):
        self.ic1 = self.ic2 = 0.0

    def tick(self, c, x):
        a1, a2, a3, m0, m1, m2 = c
        v3 = x - self.ic2
        v1 = a1 * self.ic1 + a2 * v3
        v2 = self.ic2 + a2 * self.ic1 + a3 * v3
        self.ic1 = 2.0 * v1 - self.ic1
        self.ic2 = 2.0 * v2 - self.ic2
        return m0 * x + m1 * v1 + m2 * v2


def lowpass(sample_rate, cutoff_f32, q_f32):
    g = math.tan(math.pi * float(cutoff_f32) / float(sample_rate))
    k = 1.0 / float(q_f32)
    a1 = 1.0 / (1.0 + g * (g + k))
    a2 = g * a1
    a3 = g * a2
    return (a1, a2, a3, 0.0, 0.0, 1.0)


class PreDelay:  # DelayLine<2>
    def __init__(self, max_size):
        n = 1
        while n < max_size:
            n *= 2
        self.buf = [[0.0, 0.0] for _ in range(n)]
        self.mask = n - 1
        self.wp = 0

    def process(self, delay, x):
        self.wp &= self.mask
        self.buf[self.wp] = [x[0], x[1]]
        self.wp = (self.wp + 1) & self.mask
        if self.wp > delay:
            self.wp = 0
        return list(self.buf[self.wp])


class Allpass:  # AllpassDelayLine<2>
    def __init__(self, size):
        self.buf = [[0.0, 0.0] for _ in range(size)]
        self.delay = 0
        self.wp = 0

    def process(self, x):
        rp = self.wp + 1
        if rp > self.delay:
            rp = 0
        delayed = self.buf[rp]
        out, wf = [0.0, 0.0], [0.0, 0.0]
        for ch in range(2):
            b = x[ch] - (delayed[ch] * 0.5)
            wf[ch] = b
            out[ch] = b * 0.5
        self.buf[self.wp] = wf
        self.wp += 1
        if self.wp > self.delay:
            self.wp = 0
        nd = self.buf[self.wp]
        return [out[0] + nd[0], out[1] + nd[1]]


class Line:  # ReverbDelayLine<2>
    def __init__(self, size, depth, phases):
        self.buf = [[0.0, 0.0] for _ in range(size + 1)]
        self.count = 1
        self.delay = 1
        self.depth = depth
        self.feedback = [0.0, 0.0]
        self.phase = list(phases)

    def get(self, vib_depth, blend):
        out = [0.0, 0.0]
        for ch in range(2):
            offset = (math.sin(self.phase[ch]) + 1.0) * vib_depth
            working = float(self.count) + offset
            wfl = math.floor(working)
            frac = working - wfl
            wi = int(wfl)
            r1 = wi
            if r1 > self.delay:
                r1 -= self.delay + 1
            r2 = wi + 1
            if r2 > self.delay:
                r2 -= self.delay + 1
            v1, v2 = self.buf[r1][ch], self.buf[r2][ch]
            ip = v1 * (1.0 - frac) + v2 * frac
            out[ch] = (1.0 - blend) * ip + (v1 * blend)
        return out

    def set(self, v):
        self.buf[self.count] = [v[0] + self.feedback[0], v[1] + self.feedback[1]]

    def step(self, speed):
        self.count += 1
        if self.count > self.delay:
            self.count = 0
        for ch in range(2):
            self.phase[ch] += self.depth * speed


SIZES = [8111, 7511, 7311, 6911, 6311, 6111, 5511, 4911]
DEPTHS = [0.003251, 0.002999, 0.002917, 0.002749, 0.002503, 0.002423, 0.002146, 0.002088]
AP_SIZES = [4511, 4311, 3911, 3311]
LINE_MUL = [79.0, 73.0, 71.0, 67.0, 61.0, 59.0, 53.0, 47.0]
AP_MUL = [43.0, 41.0, 37.0, 31.0]


def reverb(x, sample_rate, room_f32, wet_f32, fpd_l, fpd_r, phases16, block):
    """ReverbEffect::process over `x` (interleaved stereo f32) in blocks of `block` frames, room / wet not ramping."""
    lines = [Line(SIZES[i], DEPTHS[i], phases16[2 * i:2 * i + 2]) for i in range(8)]
    aps = [Allpass(s) for s in AP_SIZES]
    pre = PreDelay(3111)
    bq = [[Biquad(), Biquad()] for _ in range(3)]
    y = np.array(x, dtype=f32).copy()
    n_frames = len(y) // 2
    for b0 in range(0, n_frames, block):
        room, wet = float(f32(room_f32)), float(f32(wet_f32))
        cutoff = f32(10000.0 - (room * wet * 3000.0))
        size = (room * room * 75.0) + 25.0
        t = 1.0 - (0.82 - (((1.0 - room) * 0.7) + (size * 0.002)))
        depth_factor = 1.0 - (t * t) * (t * t)   # powi(4): ((t*t)*(t*t))
        blend = 0.955 - (size * 0.007)
        regen = depth_factor * 0.5
        for i in range(8):
            lines[i].delay = min(int(LINE_MUL[i] * size), len(lines[i].buf) - 1)
        for i in range(4):
            aps[i].delay = min(int(AP_MUL[i] * size), len(aps[i].buf) - 1)
        predelay = int(29.0 * size)
        cutoff = min(max(cutoff, f32(20.0)), f32(f32(sample_rate) / f32(2.0)))
        coefs = [lowpass(sample_rate, cutoff, f32(q)) for q in (1.618034, 0.618034, 0.5)]
        for n in range(b0, min(b0 + block, n_frames)):
            inp = [float(y[2 * n]), float(y[2 * n + 1])]
            if abs(inp[0]) < 1.18e-23:
                inp[0] = float(fpd_l) * 1.18e-17
            if abs(inp[1]) < 1.18e-23:
                inp[1] = float(fpd_r) * 1.18e-17
            dry = list(inp)
            v = pre.process(predelay, inp)
            v = [bq[0][ch].tick(coefs[0], v[ch]) for ch in range(2)]
            v = [math.sin(v[ch] * wet) for ch in range(2)]
            oi = aps[0].process(v)
            oj = aps[1].process(oi)
            ok = aps[2].process(oj)
            ol = aps[3].process(ok)
            for line, src in zip(lines, (ol, ok, oj, oi, oi, oj, ok, ol)):
                line.set(src)
            for line in lines:
                line.step(0.1)
            g = [line.get(7.0, blend) for line in lines]
            for ch in range(2):
                a, b, c, d, e, f, gg, h = (g[i][ch] for i in range(8))
                lines[0].feedback[ch] = (a - (b + c + d)) * regen
                lines[1].feedback[ch] = (b - (a + c + d)) * regen
                lines[2].feedback[ch] = (c - (a + b + d)) * regen
                lines[3].feedback[ch] = (d - (a + b + c)) * regen
                lines[4].feedback[ch] = (e - (f + gg + h)) * regen
                lines[5].feedback[ch] = (f - (e + gg + h)) * regen
                lines[6].feedback[ch] = (gg - (e + f + h)) * regen
                lines[7].feedback[ch] = (h - (e + f + gg)) * regen
            out = [0.0, 0.0]
            for ch in range(2):
                s = (g[0][ch] + g[1][ch] + g[2][ch] + g[3][ch] + g[4][ch] + g[5][ch] + g[6][ch] + g[7][ch]) / 8.0
                s = bq[1][ch].tick(coefs[1], s)
                s = min(max(s, -1.0), 1.0)
                s = math.asin(s)
                s = bq[2][ch].tick(coefs[2], s)
                if wet != 1.0:
                    s += dry[ch] * (1.0 - wet)
                out[ch] = s
            y[2 * n], y[2 * n + 1] = f32(out[0]), f32(out[1])
    return y


# ---- vectors ----------------------------------------------------------------------------------------------------------------
def make_vectors():
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
    import workloads

    v = {}
    rng = np.random.default_rng(7)
    for name, in_rate, out_rate, nch in (("up", 44100, 48000, 2), ("down", 96000, 48000, 2), ("mono", 22050, 48000, 1)):
        x = (rng.standard_normal(600 * nch) * 0.3).astype(f32)
        y, consumed = cubic_resample(x, in_rate, out_rate, nch, 500 * nch, 128 * nch)
        v[f"cubic_{name}_in"] = x
        v[f"cubic_{name}_out"] = y
        v[f"cubic_{name}_meta"] = np.array([in_rate, out_rate, nch, consumed], np.int64)
    for name, room, wet, seed in (("mid", 0.6, 0.5, 3), ("small_wet", 0.0, 1.0, 4)):
        x = workloads.test_signal(3 * 400, seed=21, kind="noise")
        fl, fr, ph = workloads.reverb_seeds(seed)
        v[f"reverb_{name}_in"] = x
        v[f"reverb_{name}_out"] = reverb(x, 48000, room, wet, fl, fr, ph, 400)
        v[f"reverb_{name}_meta"] = np.array([room, wet, seed, 400], np.float64)
    return v


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    vec = make_vectors()
    np.savez_compressed(os.path.join(here, "independent.npz"), **vec)
    print("independent.npz", os.path.getsize(os.path.join(here, "independent.npz")), "bytes")
