// gfx950 device primitives: smoothers, filter coefficients, filters, LFO, dB helpers.
// Arithmetic follows the reference statement by statement (same f32/f64 widths, same operation
// order; the TU is compiled with -ffp-contract=off). Reference file:line cited per function.
#pragma once
#include <hip/hip_runtime.h>

#include "pg_dev.h"

#define DEV __host__ __device__ __forceinline__
#define DEVO __device__ __forceinline__
#define DEVN __device__ __noinline__

#ifdef PG_DIAG
#define PG_STAMP(ptr, k) do { if ((ptr) && blockIdx.x == 0 && threadIdx.x == 0) (ptr)[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define PG_STAMP_VAL(ptr, k, v) do { if ((ptr) && blockIdx.x == 0 && threadIdx.x == 0) (ptr)[k] = (unsigned long long)(v); } while (0)
// accumulating lap timer (wave 0 of workgroup 0): adds the shader clocks since the previous lap to slot k
#define PG_LAP(ptr, k, t) do { if ((ptr) && blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); lapacc[(k) - 50] += t1_ - (t); (t) = t1_; } } while (0)
#define PG_LAP_DECL(t) unsigned long long t = __builtin_amdgcn_s_memtime()
#else
#define PG_LAP(ptr, k, t) do { } while (0)
#define PG_LAP_DECL(t) do { } while (0)
#define PG_STAMP_VAL(ptr, k, v) do { } while (0)
#define PG_STAMP(ptr, k) do { } while (0)
#endif

namespace pgd {

// The lane's index in the workgroup as a value the optimiser cannot see through. Super-block launches loop over the blocks of a unit
// inside the kernel; with plain threadIdx.x every per-lane address computation of every stage is loop invariant, gets hoisted in
// front of that loop and stays alive across all three stages (80 -> 672 bytes of scratch per lane, kernel +50 %). An `asm volatile`
// is never hoisted or merged: what is derived from this value is computed where it is used (one v_mov per use site).
DEVO int pg_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }

constexpr float F32_EPS100 = 1.1920929e-07f * 100.0f;
constexpr float F32_PI = 3.14159274101257324f;
constexpr float F32_TAU = 6.28318548202514648f;
constexpr double F64_PI = 3.14159265358979323846;
constexpr double F64_TAU = 6.28318530717958647692;

// Rust float semantics -------------------------------------------------------------------------
DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }   // f32::clamp (NaN passes)
DEV double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
DEV uint32_t f2u32(float x) { return !(x > 0.0f) ? 0u : (x >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)x); }    // `as u32`
DEV uint64_t d2u64(double x) { return !(x > 0.0) ? 0ull : (x >= 18446744073709551615.0 ? PG_USIZE_MAX : (uint64_t)x); }  // `as usize`
DEV uint64_t f2u64(float x) { return d2u64((double)x); }

// src/utils.rs:41-51
DEV float db_to_linear(float value) {
  const float DB_TO_LIN_FACTOR = 2.30258509299404568402f / 20.0f;
  if (value != value) return value;
  if (value == 0.0f) return 1.0f;
  if (value > -200.0f) return expf(value * DB_TO_LIN_FACTOR);
  return 0.0f;
}
// src/utils.rs:56-62
DEV void panning_factors(float pan, float& l, float& r) {
  const float POWER = 0.707106781186547524400844362104849039f;
  float normalized = (clampf(pan, -1.0f, 1.0f) + 1.0f) / 2.0f;
  l = sqrtf(1.0f - normalized) / POWER;
  r = sqrtf(normalized) / POWER;
}

// src/utils/smoothing.rs -------------------------------------------------------------------------
DEV bool sm_need_ramp(const PgSmooth& s) {
  if (s.kind == SM_EXP) {  // :198-206
    float inertia_add = (s.target - s.current) * s.a * s.comp;
    return fabsf(inertia_add) > F32_EPS100;
  } else if (s.kind == SM_LIN) {  // :360-368
    return s.pending > 0;
  }
  return fabsf(s.b) > F32_EPS100 || fabsf(s.target - s.current) > F32_EPS100;  // :499-506
}
DEV void sm_ramp(PgSmooth& s) {
  if (s.kind == SM_EXP) {  // :208-214
    s.current += (s.target - s.current) * s.a * s.comp;
  } else if (s.kind == SM_LIN) {  // :370-382
    if (s.pending > 0) {
      s.current += s.b;
      s.pending -= 1;
      if (s.pending == 0) s.current = s.target;
    }
  } else {  // :508-518
    float omega = s.a * s.comp;
    float k = omega * omega;
    float d = 2.0f * omega;
    s.b += (s.target - s.current) * k - s.b * d;
    s.current += s.b;
  }
}
DEV float sm_next(PgSmooth& s) {  // :21-28
  if (sm_need_ramp(s)) { sm_ramp(s); return s.current; }
  return s.target;
}
// n successive values of sm_next into dst (one lane; the callers hand the sequence to all lanes): the smoother's state in registers for the
// whole walk, same operations in the same order as sm_next (smoothing.rs:198-214, 360-382, 499-518). (A loop of sm_next calls on a PgSmooth
// costs ~500 cycles per frame: the record is re-read and re-written through memory on every call. One commanded reverb's wet ramp was 0.2 ms
// per 1024-frame block that way, tools/diag_cmd.py.)
// The walk is a chain of dependent f32 operations (8-10 cycles each on a lone wave, tools/exp_smooth): what can be taken out of the chain is.
// The ramp test reads the state a step starts from, and a state that fails it does not move — so it fails for good: values are produced in
// groups of eight with the tests beside the chain (not selects inside it); the group in which a test fails is walked again one value at a
// time up to that step, and everything behind it is the target. (Written out by hand: inside the render kernels the compiler leaves the plain
// loop rolled, one value per trip with a select per value: 125 cycles per value against 45 — a commanded voice's volume ramp, two values per
// frame, was 0.1 ms of its 1024-frame block.)
#define PG_SM_GROUP 8
DEV void sm_sequence(PgSmooth& s, float* dst, int n) {
  const float t = s.target;
  int i = 0;
  if (s.kind == SM_EXP) {
    float c = s.current;
    const float a = s.a, comp = s.comp;
    for (; i + PG_SM_GROUP <= n; i += PG_SM_GROUP) {
      float o[PG_SM_GROUP];
      const float c0 = c;
      bool all = true;
#pragma unroll
      for (int k = 0; k < PG_SM_GROUP; ++k) {
        const float add = (t - c) * a * comp;
        all = all && (fabsf(add) > F32_EPS100);
        c = c + add;
        o[k] = c;
      }
      if (!all) { c = c0; break; }
#pragma unroll
      for (int k = 0; k < PG_SM_GROUP; ++k) dst[i + k] = o[k];
    }
    for (; i < n; ++i) {
      const float add = (t - c) * a * comp;
      if (!(fabsf(add) > F32_EPS100)) break;
      c = c + add;
      dst[i] = c;
    }
    s.current = c;
  } else if (s.kind == SM_LIN) {
    float c = s.current;
    const float step = s.b;
    const uint32_t pending = s.pending;
    const uint32_t m = pending < (uint32_t)n ? pending : (uint32_t)n;        // steps taken in this call
    const uint32_t plain = (m == pending && m > 0) ? m - 1 : m;              // all of them plain additions but the one that uses up `pending`
    for (; (uint32_t)i < plain; ++i) { c += step; dst[i] = c; }
    if (plain < m) { c = t; dst[i++] = c; }                                  // (c += step; pending == 0: c = target)
    s.current = c; s.pending = pending - m;
  } else {
    float c = s.current, vel = s.b;
    const float omega = s.a * s.comp;
    const float k2 = omega * omega, d = 2.0f * omega;
    for (; i + PG_SM_GROUP <= n; i += PG_SM_GROUP) {
      float o[PG_SM_GROUP];
      const float c0 = c, vel0 = vel;
      bool all = true;
#pragma unroll
      for (int k = 0; k < PG_SM_GROUP; ++k) {
        all = all && (fabsf(vel) > F32_EPS100 || fabsf(t - c) > F32_EPS100);
        vel += (t - c) * k2 - vel * d;
        c += vel;
        o[k] = c;
      }
      if (!all) { c = c0; vel = vel0; break; }
#pragma unroll
      for (int k = 0; k < PG_SM_GROUP; ++k) dst[i + k] = o[k];
    }
    for (; i < n; ++i) {
      if (!(fabsf(vel) > F32_EPS100 || fabsf(t - c) > F32_EPS100)) break;
      vel += (t - c) * k2 - vel * d;
      c += vel;
      dst[i] = c;
    }
    s.current = c; s.b = vel;
  }
  for (; i < n; ++i) dst[i] = t;   // at rest: sm_next returns the target and leaves the state alone
}
DEV void sm_init(PgSmooth& s, float v) {
  s.target = v; s.current = v;
  if (s.kind == SM_LIN) s.pending = 0;
  if (s.kind == SM_SPRING) s.b = 0.0f;
}
DEV void sm_set_target(PgSmooth& s, float t) {
  if (s.kind == SM_EXP) {  // :221-226
    s.target = t;
    if (!sm_need_ramp(s)) s.current = s.target;
  } else if (s.kind == SM_LIN) {  // :312-341 (duration None)
    s.target = t;
    if (s.current == s.target) {
      s.pending = 0;
    } else {
      s.b = (s.current > s.target) ? -s.a * s.comp : s.a * s.comp;
      float pending_steps = (s.target - s.current) / s.b;
      s.pending = f2u32(fmaxf(roundf(pending_steps), 0.0f));
      if (s.pending == 0) s.current = s.target;
    }
  } else {  // :527-530
    s.target = t;
  }
}

// src/utils/dsp/filters/biquad.rs:153-271
DEV bool biquad_apply(PgBiquadCoef& c) {
  if (c.sample_rate == 0) return false;
  if (c.q <= 0.0f) return false;
  if (c.cutoff > (float)c.sample_rate / 2.0f) return false;
  double g = tan(F64_PI * (double)c.cutoff / (double)c.sample_rate);
  double k = 1.0 / (double)c.q;
  double m0, m1, m2;
  switch (c.type) {
    case 0: m0 = 0.0; m1 = 0.0; m2 = 1.0; break;            // Lowpass
    case 1: m0 = 1.0; m1 = -k; m2 = -1.0; break;            // Highpass
    case 2: m0 = 0.0; m1 = 1.0; m2 = 0.0; break;            // Bandpass
    case 3: m0 = 1.0; m1 = -k; m2 = 0.0; break;             // Notch
    case 4: m0 = 1.0; m1 = -k; m2 = -2.0; break;            // Peak
    case 5: m0 = 1.0; m1 = -2.0 * k; m2 = 0.0; break;       // Allpass
    case 6: {                                               // Bell
      double a = pow(10.0, (double)c.gain / 40.0);
      k = 1.0 / ((double)c.q * a);
      m0 = 1.0; m1 = k * (a * a - 1.0); m2 = 0.0;
    } break;
    case 7: {                                               // Lowshelf
      double a = pow(10.0, (double)c.gain / 40.0);
      g = g / sqrt(a);
      m0 = 1.0; m1 = k * (a - 1.0); m2 = a * a - 1.0;
    } break;
    default: {                                              // Highshelf
      double a = pow(10.0, (double)c.gain / 40.0);
      g = g * sqrt(a);
      m0 = a * a; m1 = k * (1.0 - a) * a; m2 = 1.0 - a * a;
    } break;
  }
  c.a1 = 1.0 / (1.0 + g * (g + k));
  c.a2 = g * c.a1;
  c.a3 = g * c.a2;
  c.m0 = m0; c.m1 = m1; c.m2 = m2;
  return true;
}
// BiquadFilterCoefficients::set  :127-150
DEV bool biquad_set(PgBiquadCoef& c, int type, uint32_t sr, float cutoff, float q, float gain) {
  if (c.type != type || c.sample_rate != sr || c.cutoff != cutoff || c.q != q || c.gain != gain) {
    c.type = type; c.sample_rate = sr; c.cutoff = cutoff; c.q = q; c.gain = gain;
    return biquad_apply(c);
  }
  return true;
}
// BiquadFilter::process_sample :314-322
DEV double biquad_tick(const PgBiquadCoef& c, double& ic1eq, double& ic2eq, double input) {
  double v0 = input;
  double v3 = v0 - ic2eq;
  double v1 = c.a1 * ic1eq + c.a2 * v3;
  double v2 = ic2eq + c.a2 * ic1eq + c.a3 * v3;
  ic1eq = 2.0 * v1 - ic1eq;
  ic2eq = 2.0 * v2 - ic2eq;
  return c.m0 * v0 + c.m1 * v1 + c.m2 * v2;
}

// FilterEffectType {Lowpass, Bandpass, Bandstop, Highpass} -> BiquadFilterType  src/effect/filter.rs:33-41
DEV int filter_to_biquad(int t) { return t == 0 ? 0 : (t == 1 ? 2 : (t == 2 ? 3 : 1)); }
// DelayEffectFilterType / ChorusEffectFilterType are SvfFilterType {Lowpass, Highpass, Bandpass}
DEV int delay_to_svf(int v) { return v == 0 ? 0 : (v == 1 ? 1 : 2); }

// src/utils/dsp/filters/svf.rs:137-168
DEV bool svf_apply(PgSvfCoef& c) {
  if (c.sample_rate == 0) return false;
  if (c.resonance < 0.0f || c.resonance > 1.0f) return false;
  if (c.cutoff > (float)c.sample_rate / 2.0f) return false;
  c.g = tan(F64_PI * (double)c.cutoff / (double)c.sample_rate);
  c.k = fmax(2.0 * (1.0 - (double)c.resonance * 0.97), 0.03);
  c.a1 = 1.0 / (1.0 + c.g * (c.g + c.k));
  c.a2 = c.g * c.a1;
  c.a3 = c.g * c.a2;
  return true;
}
DEV bool svf_set(PgSvfCoef& c, int type, uint32_t sr, float cutoff, float res) {  // :115-134
  if (c.type != type || c.sample_rate != sr || c.cutoff != cutoff || c.resonance != res) {
    c.type = type; c.sample_rate = sr; c.cutoff = cutoff; c.resonance = res;
    return svf_apply(c);
  }
  return true;
}
DEV double svf_tick(const PgSvfCoef& c, double& ic1eq, double& ic2eq, double input) {  // :211-222
  double v3 = input - ic2eq;
  double v1 = c.a1 * ic1eq + c.a2 * v3;
  double v2 = ic2eq + c.a2 * ic1eq + c.a3 * v3;
  ic1eq = 2.0 * v1 - ic1eq;
  ic2eq = 2.0 * v2 - ic2eq;
  if (c.type == 0) return v2;            // Lowpass
  if (c.type == 2) return v1;            // Bandpass
  return input - c.k * v1 - v2;          // Highpass
}
// src/utils/dsp/filters/dc.rs:84-88
DEV double dc_tick(PgDc& d, double sample) {
  d.y1 = sample - d.x1 + d.r * d.y1;
  d.x1 = sample;
  return d.y1;
}
DEV double dc_r(double hz, uint32_t sr) { return 1.0 - (F64_TAU * hz / (double)sr); }  // :54-61

// src/utils/dsp/lfo.rs
DEV float sine_approx(float x) {  // :9-19
  const float B = 4.0f / F32_PI;
  const float C = -4.0f / (F32_PI * F32_PI);
  const float P = 0.225f;
  float y = B * x + C * x * fabsf(x);
  return P * (y * fabsf(y) - y) + y;
}
// rand ^0.9 SmallRng on 64-bit targets = Xoshiro256++ (D. Blackman, S. Vigna; public domain reference xoshiro256plusplus.c):
//   result = rotl(s0 + s3, 23) + s0;  t = s1 << 17;  s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;  s2 ^= t;  s3 = rotl(s3, 45)
// and `rng.random::<f32>()` (StandardUniform for f32: 24 bits of next_u32, which for this generator is the upper half of next_u64).
DEV uint64_t xoshiro256pp_next(uint64_t* s) {
  const uint64_t r = ((s[0] + s[3]) << 23 | (s[0] + s[3]) >> 41) + s[0];
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t;
  s[3] = (s[3] << 45) | (s[3] >> 19);
  return r;
}
DEV float rng_random_f32(uint64_t* s) { return (float)(uint32_t)(xoshiro256pp_next(s) >> 40) * (1.0f / 16777216.0f); }
DEV float lfo_random_bipolar(uint64_t* s) { return rng_random_f32(s) * 2.0f - 1.0f; }   // `rng.random::<f32>() * 2.0 - 1.0`  lfo.rs:74-76,90-92,246-249
DEV float lfo_value(const PgLfo& l) {  // :122-152 (deterministic shapes)
  float ph = l.phase;
  switch (l.waveform) {
    case 0: { float p = (ph < 0.5f) ? ph * F32_TAU : (ph - 1.0f) * F32_TAU; return sine_approx(p); }
    case 1: return (ph < 0.25f) ? ph * 4.0f : ((ph < 0.75f) ? 2.0f - ph * 4.0f : ph * 4.0f - 4.0f);
    case 2: return ph * 2.0f - 1.0f;
    case 3: return 1.0f - ph * 2.0f;
    case 4: return (ph < 0.5f) ? 1.0f : -1.0f;
    default: return 0.0f;
  }
}
DEV float lfo_run(PgLfo& l) {  // :122-169, :234-239
  float v = lfo_value(l);
  l.phase += l.phase_inc;
  if (l.phase >= 1.0f) l.phase -= 1.0f;
  return v;
}
// The DelayEffect's Lfo with all seven shapes: Random (sample & hold) and Smooth Random (cosine-interpolated jitter) read their state
// and draw new values on every phase wrap (lfo.rs:145-152,160-169,241-252)
DEV float delay_lfo_run(PgDelay& d) {
  PgLfo& l = d.lfo;
  if (l.waveform < 5) return lfo_run(l);
  float v;
  if (l.waveform == 5) v = d.lfo_sample_hold;
  else {
    const float p = 1.57079632679489661923f - l.phase * F32_PI;   // FRAC_PI_2 - phase * PI
    const float t = (1.0f - sine_approx(p)) * 0.5f;
    v = d.lfo_jitter_current + t * (d.lfo_jitter_target - d.lfo_jitter_current);
  }
  l.phase += l.phase_inc;   // advance_phase_random
  if (l.phase >= 1.0f) {
    l.phase -= 1.0f;
    d.lfo_sample_hold = lfo_random_bipolar(d.lfo_rng);
    d.lfo_jitter_current = d.lfo_jitter_target;
    d.lfo_jitter_target = lfo_random_bipolar(d.lfo_rng);
  }
  return v;
}
// Lfo::reset (lfo.rs:84-94): phase 0; the random shapes draw again
DEV void delay_lfo_reset(PgDelay& d) {
  d.lfo.phase = 0.0f;
  if (d.lfo.waveform >= 5) {
    d.lfo_sample_hold = lfo_random_bipolar(d.lfo_rng);
    d.lfo_jitter_current = d.lfo_jitter_target;
    d.lfo_jitter_target = lfo_random_bipolar(d.lfo_rng);
  }
}
// `steps` iterations of { p += d; if (p >= 1) p -= 1; } (the phase update of Lfo::run, lfo.rs:234-239) in f32, exactly, in
// O(pieces) instead of O(steps): inside a binade fl(p + d) = p + du with du = d rounded to a multiple of ulp(p) (no tie), and
// every p + m*du below the binade's end is representable; the step that leaves the binade, a wrap, a tie or a degenerate value
// takes the plain hardware step. Checked against the serial loop on the host (200k random (p, d, steps): bit-identical).
DEV void f32_phase_advance(float& p, float d, int steps) {
  while (steps > 0) {
    const uint32_t bits = __builtin_bit_cast(uint32_t, p);
    const int e = (int)((bits >> 23) & 0xff);
    bool closed = false;
    if (e > 24 && e < 0xff && p > 0.0f && d > 0.0f && p < 1.0f) {
      const double u = __builtin_bit_cast(double, (unsigned long long)(e - 127 - 23 + 1023) << 52);   // ulp(p)
      const double D = (double)d / u;                                                                        // exact (power of two)
      const double Dr = rint(D);
      if (D < 16777216.0 && Dr >= 1.0 && D - floor(D) != 0.5) {
        const double top = __builtin_bit_cast(double, (unsigned long long)(e - 127 + 1 + 1023) << 52);  // end of the binade
        const double room = (top - u) - (double)p;   // a multiple of u
        const double q = floor(room / (Dr * u));     // steps that stay inside the binade (small integers: exact)
        if (q >= 1.0) {
          const int m = q > (double)steps ? steps : (int)q;
          p = (float)((double)p + (double)m * (Dr * u));
          steps -= m;
          closed = true;
        }
      }
    }
    if (!closed) {
      p += d;
      if (p >= 1.0f) p -= 1.0f;
      steps -= 1;
    }
  }
}
DEV void lfo_set_rate(PgLfo& l, uint32_t sr, double rate) { l.phase_inc = (float)(rate / (double)sr); }  // :100-102
DEV void lfo_set_phase_degrees(PgLfo& l, float p) {  // :105-114 (rem_euclid(1.0))
  float q = p / F32_TAU;
  float r = fmodf(q, 1.0f);
  if (r < 0.0f) r += 1.0f;
  l.phase = r;
}
// src/utils/dsp/envelope.rs:51-60
DEV float env_run(float& current, float attack_coeff, float release_coeff, float input) {
  if (input > current) current = input + attack_coeff * (current - input);
  else current = input + release_coeff * (current - input);
  return current;
}
// log10f as the HOST's libm computes it. The level detectors of the Compressor and the Gate feed `20 * log10f(peak)` into decisions — the
// envelope follower's attack / release switch, the gate's `envelope >= threshold`, the edges of the compressor's knee (the reference's gain
// computer even has a hole exactly on the upper edge, compressor.rs:272-283) — so one ulp between two libm implementations can move a gate's
// opening by a frame or put a one-frame click into one side only (found by the fuzz campaigns). The reference calls the platform's
// log10f; on the Linux boxes this library runs on that is glibc's: __ieee754_log10f (fdlibm's scaling around logf, sysdeps/ieee754/flt-32/
// e_log10f.c) over the table-driven logf of ARM's optimized-routines (sysdeps/ieee754/flt-32/e_logf.c, MIT), restated here operation by operation
// for finite normal x > 0 (the callers pass peaks > 1e-6) — bit-identical to glibc 2.35 on 2e8 random arguments (tests/host/log10f_check.c).
DEV float pg_logf_glibc(float x) {
  const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
                           0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                           0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
  const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,
                           -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, -0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4,
                           0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2};
  uint32_t ix;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const int k = (int)((int32_t)tmp >> 23);
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  float zf;
  memcpy(&zf, &iz, 4);
  const double z = (double)zf;
  const double r = z * invc[i] - 1.0;
  const double y0 = logc[i] + (double)k * 0x1.62e42fefa39efp-1;
  const double r2 = r * r;
  double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
  y = -0x1.00ea348b88334p-2 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}
// expf as the HOST's libm computes it: the VolumeFader's fade-out inertia is 1 - expf(-1 / samples) (fader.rs:67-91) — a difference of two
// nearly equal numbers, so one ulp between two expf implementations is 1e-4 of the inertia and a few 1e-6 in the faded samples of a path
// that is otherwise bit-identical with the oracle (found by the fuzz over file sources at other source rates). glibc's expf
// (sysdeps/ieee754/flt-32/e_expf.c, ARM optimized-routines, MIT): x N / ln2 = k + r, 2^(k/N) from a 32-entry table, a cubic in r, all in
// double. Restated operation by operation for |x| < 88; bit-identical to the host's on a dense sweep (tests/host/expf_check.hip).
DEV float pg_expf_glibc(float x) {
  const unsigned long long T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  uint32_t ux;
  memcpy(&ux, &x, 4);
  if (((ux >> 20) & 0x7ffu) >= 0x42bu) return expf(x);  // |x| >= 88, inf, nan: the platform's (not reached by the fader)
  const double xd = (double)x;
  double z = (0x1.71547652b82fep+0 * 32.0) * xd;
  double kd = z + 0x1.8p+52;
  unsigned long long ki;
  memcpy(&ki, &kd, 8);
  kd -= 0x1.8p+52;
  const double r = z - kd;
  unsigned long long t = T[ki % 32ull];
  t += ki << (52 - 5);
  double sc;
  memcpy(&sc, &t, 8);
  z = (0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0) * r + (0x1.ebfce50fac4f3p-3 / 32.0 / 32.0);
  const double r2 = r * r;
  double y = (0x1.62e42ff0c52d6p-1 / 32.0) * r + 1.0;
  y = z * r2 + y;
  y = y * sc;
  return (float)y;
}
DEV float pg_log10f(float x) {
  uint32_t ux;
  memcpy(&ux, &x, 4);
  if (ux < 0x00800000u || ux >= 0x7f800000u) return log10f(x);  // zero, subnormal, negative, inf, nan: not reached by the level detectors
  const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
  int32_t hx = (int32_t)ux;
  const int32_t k = (hx >> 23) - 127;
  const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
  const float y = (float)(k + i);
  float xr;
  const uint32_t uh = (uint32_t)hx;
  memcpy(&xr, &uh, 4);
  const float z = y * log10_2lo + ivln10 * pg_logf_glibc(xr);
  return z + y * log10_2hi;
}
DEV float env_coeff(float t, uint32_t sr) { return (t > 0.0f) ? expf(-1.0f / (t * (float)sr)) : 0.0f; }  // :27-42

// workgroup max-reduction of |x| over an LDS buffer (max_abs_sample, src/utils/buffer.rs:150-173; order free)
// Asynchronous global -> LDS copy of one dword per lane (global_load_lds_dword): lane l of the wave writes lds_wave_base[l]. The
// instruction is issued from inline asm on purpose: the compiler counts a tracked LDS-DMA as pending LDS stores and drains it
// (s_waitcnt vmcnt(0)) at the next workgroup barrier, which would turn the copy back into a blocking load. The caller waits with
// lds_dma_wait() before a barrier that precedes the first read; in between the transfer is invisible to the compiler, whose own
// vmcnt waits stay conservative (loads return in order). M0 is saved and restored around the transfer.
__device__ __forceinline__ void lds_dma_dword(const float* g, float* lds_wave_base) {
  const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(a), "v"(g));
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

DEVO float wg_max_abs(const float* buf, int n, float* red /* LDS, >= blockDim.x/64 floats */) {
  float m = 0.0f;
  for (int i = pg_tid(); i < n; i += blockDim.x) m = fmaxf(m, fabsf(buf[i]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((pg_tid() & 63) == 0) red[pg_tid() >> 6] = m;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, red[w]);
  __syncthreads();
  return r;
}

}  // namespace pgd
