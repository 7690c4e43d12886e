// ORACLE — TEST INFRASTRUCTURE ONLY (see po_utils.hpp header). Parity unpinned beyond the
// reference KATs listed there.
//
// CPU restatement of src/utils/dsp/** and src/utils/resampler/cubic.rs of emuell/phonic.
#pragma once
#include "po_utils.hpp"

namespace po {

// ---- src/utils/dsp/filters/biquad.rs ---------------------------------------------------------
enum class BiquadType { Lowpass, Highpass, Bandpass, Notch, Peak, Allpass, Bell, Lowshelf, Highshelf };

struct BiquadCoefficients {  // :31-272
  BiquadType filter_type = BiquadType::Lowpass;
  uint32_t sample_rate = 0;
  float cutoff = 0.0f, q = 0.0f, gain = 0.0f;
  double a1 = 0, a2 = 0, a3 = 0, m0 = 0, m1 = 0, m2 = 0;

  bool set_filter_type(BiquadType t) { if (filter_type != t) { filter_type = t; return apply(); } return true; }
  bool set_cutoff(float c) { if (cutoff != c) { cutoff = c; return apply(); } return true; }
  bool set(BiquadType t, uint32_t sr, float c, float q_, float g) {  // :127-150
    if (filter_type != t || sample_rate != sr || cutoff != c || q != q_ || gain != g) {
      filter_type = t; sample_rate = sr; cutoff = c; q = q_; gain = g;
      return apply();
    }
    return true;
  }
  bool apply() {  // :153-271
    if (sample_rate == 0) return false;
    if (q <= 0.0f) return false;
    if (cutoff > (float)sample_rate / 2.0f) return false;
    const double g0 = std::tan(F64_PI * (double)cutoff / (double)sample_rate);
    switch (filter_type) {
      case BiquadType::Lowpass: {
        double g = g0, k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 0.0; m1 = 0.0; m2 = 1.0;
      } break;
      case BiquadType::Highpass: {
        double g = g0, k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 1.0; m1 = -k; m2 = -1.0;
      } break;
      case BiquadType::Bandpass: {
        double g = g0, k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 0.0; m1 = 1.0; m2 = 0.0;
      } break;
      case BiquadType::Notch: {
        double g = g0, k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 1.0; m1 = -k; m2 = 0.0;
      } break;
      case BiquadType::Peak: {
        double g = g0, k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 1.0; m1 = -k; m2 = -2.0;
      } break;
      case BiquadType::Allpass: {
        double g = g0, k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 1.0; m1 = -2.0 * k; m2 = 0.0;
      } break;
      case BiquadType::Bell: {
        double a = std::pow(10.0, (double)gain / 40.0);
        double g = g0, k = 1.0 / ((double)q * a);
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 1.0; m1 = k * (a * a - 1.0); m2 = 0.0;
      } break;
      case BiquadType::Lowshelf: {
        double a = std::pow(10.0, (double)gain / 40.0);
        double g = g0 / std::sqrt(a), k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = 1.0; m1 = k * (a - 1.0); m2 = a * a - 1.0;
      } break;
      case BiquadType::Highshelf: {
        double a = std::pow(10.0, (double)gain / 40.0);
        double g = g0 * std::sqrt(a), k = 1.0 / (double)q;
        a1 = 1.0 / (1.0 + g * (g + k)); a2 = g * a1; a3 = g * a2; m0 = a * a; m1 = k * (1.0 - a) * a; m2 = 1.0 - a * a;
      } break;
    }
    return true;
  }
};

struct BiquadFilter {  // :286-331
  double ic1eq = 0.0, ic2eq = 0.0;
  inline double process_sample(const BiquadCoefficients& c, double input) {  // :314-322
    double v0 = input;
    double v3 = v0 - ic2eq;
    double v1 = c.a1 * ic1eq + c.a2 * v3;
    double v2 = ic2eq + c.a2 * ic1eq + c.a3 * v3;
    ic1eq = 2.0 * v1 - ic1eq;
    ic2eq = 2.0 * v2 - ic2eq;
    return c.m0 * v0 + c.m1 * v1 + c.m2 * v2;
  }
  void reset() { ic1eq = 0.0; ic2eq = 0.0; }
};

// ---- src/utils/dsp/filters/svf.rs ------------------------------------------------------------
enum class SvfType { Lowpass, Highpass, Bandpass };

struct SvfCoefficients {  // :30-168
  SvfType filter_type = SvfType::Lowpass;
  uint32_t sample_rate = 0;
  float cutoff = 0.0f, resonance = 0.0f;
  double g = 0, k = 0, a1 = 0, a2 = 0, a3 = 0;
  bool set_filter_type(SvfType t) { if (filter_type != t) { filter_type = t; return apply(); } return true; }
  bool set(SvfType t, uint32_t sr, float c, float r) {  // :115-134
    if (filter_type != t || sample_rate != sr || cutoff != c || resonance != r) {
      filter_type = t; sample_rate = sr; cutoff = c; resonance = r;
      return apply();
    }
    return true;
  }
  bool apply() {  // :137-168
    if (sample_rate == 0) return false;
    if (resonance < 0.0f || resonance > 1.0f) return false;
    if (cutoff > (float)sample_rate / 2.0f) return false;
    g = std::tan(F64_PI * (double)cutoff / (double)sample_rate);
    k = rmax(2.0 * (1.0 - (double)resonance * 0.97), 0.03);
    a1 = 1.0 / (1.0 + g * (g + k));
    a2 = g * a1;
    a3 = g * a2;
    return true;
  }
};

struct SvfFilter {  // :175-230
  double ic1eq = 0.0, ic2eq = 0.0;
  inline double process_sample(const SvfCoefficients& c, double input) {  // :211-222
    double v3 = input - ic2eq;
    double v1 = c.a1 * ic1eq + c.a2 * v3;
    double v2 = ic2eq + c.a2 * ic1eq + c.a3 * v3;
    ic1eq = 2.0 * v1 - ic1eq;
    ic2eq = 2.0 * v2 - ic2eq;
    switch (c.filter_type) {
      case SvfType::Lowpass: return v2;
      case SvfType::Bandpass: return v1;
      default: return input - c.k * v1 - v2;
    }
  }
  void reset() { ic1eq = 0.0; ic2eq = 0.0; }
};

// ---- src/utils/dsp/filters/dc.rs -------------------------------------------------------------
enum class DcMode { Slow, Default, Fast };
inline double dc_mode_hz(DcMode m) { return m == DcMode::Slow ? 1.0 : (m == DcMode::Default ? 5.0 : 20.0); }  // :20-28
struct DcFilter {  // :35-89
  double y1 = 0.0, x1 = 0.0, r = 0.999;
  DcFilter() {}
  DcFilter(uint32_t sample_rate, DcMode mode) { r = 1.0 - (F64_TAU * dc_mode_hz(mode) / (double)sample_rate); }
  void reset() { x1 = 0.0; y1 = 0.0; }
  void set_mode(DcMode mode, uint32_t sample_rate) { r = 1.0 - (F64_TAU * dc_mode_hz(mode) / (double)sample_rate); }
  inline double process_sample(double sample) {  // :84-88
    y1 = sample - x1 + r * y1;
    x1 = sample;
    return y1;
  }
};

// ---- src/utils/dsp/delay.rs ------------------------------------------------------------------
inline size_t next_power_of_two(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

template <int CH>
struct DelayLine {  // :19-67
  std::vector<double> buffer;  // [frames][CH]
  size_t buffer_mask = 0, write_pos = 0;
  DelayLine() {}
  explicit DelayLine(size_t max_size) { size_t n = next_power_of_two(max_size); buffer.assign(n * CH, 0.0); buffer_mask = n - 1; }
  void flush() { std::fill(buffer.begin(), buffer.end(), 0.0); write_pos = 0; }
  inline void process(size_t delay, const double* input, double* out) {  // :47-66
    write_pos &= buffer_mask;
    for (int c = 0; c < CH; ++c) buffer[write_pos * CH + c] = input[c];
    write_pos = (write_pos + 1) & buffer_mask;
    if (write_pos > delay) write_pos = 0;
    for (int c = 0; c < CH; ++c) out[c] = buffer[write_pos * CH + c];
  }
};

template <int CH>
struct InterpolatedDelayLine {  // :79-156
  std::vector<double> buffer;
  size_t buffer_mask = 0, write_pos = 0;
  InterpolatedDelayLine() {}
  explicit InterpolatedDelayLine(size_t max_size) { size_t n = next_power_of_two(max_size); buffer.assign(n * CH, 0.0); buffer_mask = n - 1; }
  void flush() { std::fill(buffer.begin(), buffer.end(), 0.0); write_pos = 0; }
  inline void process(const float* input, float feedback, float delay, float* output) {  // :107-155
    double read_pos = (double)write_pos - (double)delay;
    double read_pos_floor = std::floor(read_pos);
    double fraction = read_pos - read_pos_floor;
    int64_t index1 = as_isize(read_pos_floor);
    int64_t index2 = index1 + 1;
    size_t read_idx1 = ((size_t)index1) & buffer_mask;
    size_t read_idx2 = ((size_t)index2) & buffer_mask;
    log_index(read_idx1);
    double f1[CH], f2[CH];
    for (int c = 0; c < CH; ++c) { f1[c] = buffer[read_idx1 * CH + c]; f2[c] = buffer[read_idx2 * CH + c]; }
    for (int c = 0; c < CH; ++c) output[c] = (float)(f1[c] + (f2[c] - f1[c]) * fraction);
    size_t w = write_pos & buffer_mask;
    for (int c = 0; c < CH; ++c) buffer[w * CH + c] = (double)input[c] + (double)output[c] * (double)feedback;
    write_pos = (write_pos + 1) & buffer_mask;
  }
};

template <int CH>
struct LookupDelayLine {  // :172-271
  std::vector<double> buffer;
  size_t write_pos = 0, buffer_mask = 0, delay_frames = 0;
  double peak_value_ = 0.0;
  size_t peak_pos = 0;
  LookupDelayLine() {}
  LookupDelayLine(uint32_t sample_rate, float delay_time) {  // :182-203
    delay_frames = as_usize(std::ceil(delay_time * (float)sample_rate));
    if (delay_frames > 0) { size_t n = next_power_of_two(delay_frames); buffer.assign(n * CH, 0.0); buffer_mask = n - 1; }
  }
  inline void process(const float* in, float* delayed) {  // :206-265
    if (delay_frames == 0) { for (int c = 0; c < CH; ++c) delayed[c] = in[c]; return; }
    size_t buffer_frames = buffer.size() / CH;
    size_t read_frame_index = (write_pos + buffer_frames - delay_frames) & buffer_mask;
    for (int c = 0; c < CH; ++c) delayed[c] = (float)buffer[read_frame_index * CH + c];
    size_t write_frame_index = write_pos & buffer_mask;
    for (int c = 0; c < CH; ++c) buffer[write_frame_index * CH + c] = (double)in[c];
    bool peak_expired = peak_pos == read_frame_index;
    double new_peak = 0.0;
    for (int c = 0; c < CH; ++c) new_peak = rmax(new_peak, (double)std::fabs(in[c]));
    if (new_peak >= peak_value_) {
      peak_value_ = new_peak;
      peak_pos = write_pos;
    } else if (peak_expired) {
      peak_value_ = 0.0;
      for (size_t i = 0; i < delay_frames; ++i) {
        size_t frame_index = (write_pos + buffer_frames - i) & buffer_mask;
        double frame_peak = 0.0;
        for (int c = 0; c < CH; ++c) frame_peak = rmax(frame_peak, std::fabs(buffer[frame_index * CH + c]));
        if (frame_peak >= peak_value_) { peak_value_ = frame_peak; peak_pos = frame_index; }
      }
    }
    write_pos = (write_pos + 1) & buffer_mask;
  }
  float peak_value() const { return (float)peak_value_; }
};

template <int CH>
struct AllpassDelayLine {  // :283-351
  std::vector<double> buffer;
  size_t delay = 0, write_pos = 0;
  AllpassDelayLine() {}
  explicit AllpassDelayLine(size_t max_size) { buffer.assign(max_size * CH, 0.0); }
  size_t frames() const { return buffer.size() / CH; }
  void flush() { std::fill(buffer.begin(), buffer.end(), 0.0); write_pos = 0; }
  void set_delay(size_t d) { delay = std::min(d, frames() - 1); }
  inline void process(const double* input, double* output) {  // :314-350
    size_t read_pos = write_pos + 1;
    if (read_pos > delay) read_pos = 0;
    double delayed[CH], write_frame[CH];
    for (int c = 0; c < CH; ++c) delayed[c] = buffer[read_pos * CH + c];
    for (int c = 0; c < CH; ++c) {
      double val_in = input[c];
      double buf = val_in - (delayed[c] * 0.5);
      write_frame[c] = buf;
      output[c] = buf * 0.5;
    }
    for (int c = 0; c < CH; ++c) buffer[write_pos * CH + c] = write_frame[c];
    write_pos += 1;
    if (write_pos > delay) write_pos = 0;
    for (int c = 0; c < CH; ++c) output[c] += buffer[write_pos * CH + c];
  }
};

// ---- src/utils/dsp/lfo.rs --------------------------------------------------------------------
inline float sine_approx(float x) {  // :9-19
  const float B = 4.0f / F32_PI;
  const float C = -4.0f / (F32_PI * F32_PI);
  const float P = 0.225f;
  float y = B * x + C * x * std::fabs(x);
  return P * (y * std::fabs(y) - y) + y;
}
enum class LfoWaveform { Sine, Triangle, RampUp, RampDown, Square, Random, SmoothRandom };
// rand ^0.9 (a dependency of the reference, not vendored): `SmallRng` on 64-bit targets is Xoshiro256++ (D. Blackman, S. Vigna, public
// domain reference xoshiro256plusplus.c; known answers for the state {1, 2, 3, 4} in tests/test_oracle_kats.py), `next_u32` its upper
// half, and `random::<f32>()` (StandardUniform) 24 bits of that times 2^-24. The reference seeds it from the OS (lfo.rs:73): the state is
// an explicit input here (pg_effect_init::lfo_rng_state), and SplitMix64(0x5EED0000) x 4 when none is given.
struct SmallRng {
  uint64_t s[4];
  SmallRng() { uint64_t z = 0x5EED0000ull; for (int i = 0; i < 4; ++i) { z += 0x9E3779B97F4A7C15ull; uint64_t x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; s[i] = x ^ (x >> 31); } }
  explicit SmallRng(const uint64_t st[4]) { for (int i = 0; i < 4; ++i) s[i] = st[i]; }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next_u64() {
    const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
  }
  uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
  float random_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

struct Lfo {  // :50-253
  float phase = 0.0f, phase_inc = 0.0f;
  LfoWaveform waveform = LfoWaveform::Sine;
  float sample_hold_value = 0.0f, jitter_current = 0.0f, jitter_target = 0.0f;
  SmallRng rng;
  Lfo() : Lfo(44100, 1.0, LfoWaveform::Sine) {}
  Lfo(uint32_t sample_rate, double rate, LfoWaveform w, const SmallRng& seeded = SmallRng())  // :70-86 (`SmallRng::from_os_rng()` -> the seeded state)
      : phase(0.0f), phase_inc((float)(rate / (double)sample_rate)), waveform(w), rng(seeded) {
    sample_hold_value = rng.random_f32() * 2.0f - 1.0f;
    jitter_current = rng.random_f32() * 2.0f - 1.0f;
    jitter_target = rng.random_f32() * 2.0f - 1.0f;
  }
  void reset() {  // :89-99
    phase = 0.0f;
    if (waveform == LfoWaveform::Random || waveform == LfoWaveform::SmoothRandom) {
      sample_hold_value = rng.random_f32() * 2.0f - 1.0f;
      jitter_current = jitter_target;
      jitter_target = rng.random_f32() * 2.0f - 1.0f;
    }
  }
  void set_rate(uint32_t sample_rate, double rate) { phase_inc = (float)(rate / (double)sample_rate); }
  void set_phase(float p) {  // rem_euclid(1.0) :105-107
    float r = std::fmod(p, 1.0f);
    if (r < 0.0f) r += 1.0f;
    phase = r;
  }
  void set_phase_degrees(float p) { set_phase(p / F32_TAU); }  // :110-114
  void set_waveform(LfoWaveform w) { waveform = w; }
  float run() {  // :122-169
    float value = 0.0f;
    switch (waveform) {
      case LfoWaveform::Sine: {
        float p = (phase < 0.5f) ? phase * F32_TAU : (phase - 1.0f) * F32_TAU;
        value = sine_approx(p);
      } break;
      case LfoWaveform::Triangle:
        if (phase < 0.25f) value = phase * 4.0f;
        else if (phase < 0.75f) value = 2.0f - phase * 4.0f;
        else value = phase * 4.0f - 4.0f;
        break;
      case LfoWaveform::RampUp: value = phase * 2.0f - 1.0f; break;
      case LfoWaveform::RampDown: value = 1.0f - phase * 2.0f; break;
      case LfoWaveform::Square: value = (phase < 0.5f) ? 1.0f : -1.0f; break;
      case LfoWaveform::Random: value = sample_hold_value; break;
      default: {  // SmoothRandom: cosine interpolation between two random values :154-159
        float p = 1.57079632679489661923f - phase * F32_PI;
        float t = (1.0f - sine_approx(p)) * 0.5f;
        value = jitter_current + t * (jitter_target - jitter_current);
      } break;
    }
    phase += phase_inc;  // advance_phase :234-239 / advance_phase_random :241-252
    if (phase >= 1.0f) {
      phase -= 1.0f;
      if (waveform == LfoWaveform::Random || waveform == LfoWaveform::SmoothRandom) {
        sample_hold_value = rng.random_f32() * 2.0f - 1.0f;
        jitter_current = jitter_target;
        jitter_target = rng.random_f32() * 2.0f - 1.0f;
      }
    }
    return value;
  }
};

// ---- src/utils/dsp/envelope.rs ---------------------------------------------------------------
struct EnvelopeFollower {  // :5-75
  float current_value = 0.0f, attack_coeff = 0.0f, release_coeff = 0.0f;
  uint32_t sample_rate = 44100;
  EnvelopeFollower() : EnvelopeFollower(44100, 0.01f, 0.1f) {}
  EnvelopeFollower(uint32_t sr, float attack, float release) : sample_rate(sr) { set_attack_time(attack); set_release_time(release); }
  void set_attack_time(float t) { attack_coeff = (t > 0.0f) ? std::exp(-1.0f / (t * (float)sample_rate)) : 0.0f; }
  void set_release_time(float t) { release_coeff = (t > 0.0f) ? std::exp(-1.0f / (t * (float)sample_rate)) : 0.0f; }
  float run(float input) {  // :51-60
    if (input > current_value) current_value = input + attack_coeff * (current_value - input);
    else current_value = input + release_coeff * (current_value - input);
    return current_value;
  }
  void reset(float v) { current_value = v; }
};

// ---- src/utils/resampler/cubic.rs -----------------------------------------------------------
struct CubicInterpolator {  // :10-143
  float input[4] = {0, 0, 0, 0};
  float sub_pos = 0.0f, ratio = 1.0f;
  bool is_initialized = false;
  void reset() { input[0] = input[1] = input[2] = input[3] = 0.0f; sub_pos = 0.0f; is_initialized = false; }
  inline void push_sample(float v) { input[3] = input[2]; input[2] = input[1]; input[1] = input[0]; input[0] = v; }
  inline float interpolate(float fraction) const {  // :125-142
    float ym1 = input[3], y0 = input[2], y1 = input[1], y2 = input[0];
    float c0 = y0;
    float c1 = (y1 - ym1) * 0.5f;
    float c2 = ym1 - y0 * 2.5f + y1 * 2.0f - y2 * 0.5f;
    float c3 = (y2 - ym1) * 0.5f + (y0 - y1) * 1.5f;
    return ((c3 * fraction + c2) * fraction + c1) * fraction + c0;
  }
  // returns (consumed samples, produced samples); in/out lengths in samples  :36-114
  void process(const float* in, size_t in_len, float* out, size_t out_len, size_t channel_index, size_t channel_count,
               size_t& consumed_samples, size_t& produced_samples) {
    size_t num_in = in_len / channel_count, num_out = out_len / channel_count;
    size_t num_consumed = 0, num_produced = 0;
    if (std::fabs(ratio - 1.0f) < 0.000001f) {
      size_t mn = std::min(in_len, out_len);
      for (size_t i = 0; i < mn; ++i) out[i] = in[i];
      consumed_samples = mn; produced_samples = mn;
      return;
    }
    if (!is_initialized && num_in >= 3) {
      is_initialized = true;
      for (size_t f = 0; f < 3; ++f) { push_sample(in[f * channel_count + channel_index]); num_consumed += 1; }
    }
    if (ratio < 1.0f) {
      while (num_produced < num_out) {
        if (sub_pos >= 1.0f) {
          if (num_consumed >= num_in) break;
          push_sample(in[num_consumed * channel_count + channel_index]);
          num_consumed += 1;
          sub_pos -= 1.0f;
        }
        out[num_produced * channel_count + channel_index] = interpolate(sub_pos);
        num_produced += 1;
        sub_pos += ratio;
      }
    } else {
      bool brk = false;
      while (num_produced < num_out && !brk) {
        while (sub_pos < ratio) {
          if (num_consumed >= num_in) { brk = true; break; }
          push_sample(in[num_consumed * channel_count + channel_index]);
          num_consumed += 1;
          sub_pos += 1.0f;
        }
        if (brk) break;
        sub_pos -= ratio;
        out[num_produced * channel_count + channel_index] = interpolate(1.0f - sub_pos);
        num_produced += 1;
      }
    }
    consumed_samples = num_consumed * channel_count;
    produced_samples = num_produced * channel_count;
  }
};

struct CubicResampler {  // :150-207
  uint32_t input_rate, output_rate;
  size_t channel_count;
  std::vector<CubicInterpolator> interpolators;
  CubicResampler(uint32_t in_rate, uint32_t out_rate, size_t ch) : input_rate(in_rate), output_rate(out_rate), channel_count(ch) {
    CubicInterpolator ci;
    ci.ratio = (float)((double)in_rate / (double)out_rate);  // spec.input_ratio() as f32  :164
    interpolators.assign(ch, ci);
  }
  void process(const float* in, size_t in_len, float* out, size_t out_len, size_t& consumed, size_t& produced) {  // :179-186
    consumed = 0; produced = 0;
    for (size_t c = 0; c < channel_count; ++c) interpolators[c].process(in, in_len, out, out_len, c, channel_count, consumed, produced);
  }
  void update(uint32_t in_rate, uint32_t out_rate) {  // :188-200
    input_rate = in_rate; output_rate = out_rate;
    float r = (float)((double)in_rate / (double)out_rate);
    for (auto& i : interpolators) i.ratio = r;
  }
  void reset() { for (auto& i : interpolators) i.reset(); }
};

}  // namespace po
