// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of emuell/phonic (v0.16.0) utility arithmetic. Every function cites the
// reference file:line it follows (paths relative to the reference repo). Scalar, same f32/f64
// widths and the same evaluation order as the Rust; build with -ffp-contract=off and without
// fast-math (Rust never contracts or reassociates).
//
// PARITY PINNING: the reference is Rust and cannot be built in this environment (no cargo /
// rustc). This restatement is pinned only by the reference's own known-answer tests
// (src/utils/buffer.rs:621-797, src/utils/smoothing.rs:556-728, src/utils.rs:94-104,
// src/source/file/preloaded.rs:486-533). Everything else is "parity unpinned".
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace po {

// ---- Rust numeric semantics helpers ---------------------------------------------------------
// f32::max / f32::min: if one argument is NaN the other is returned.
inline float rmaxf(float a, float b) { return std::fmax(a, b); }
inline float rminf(float a, float b) { return std::fmin(a, b); }
inline double rmax(double a, double b) { return std::fmax(a, b); }
inline double rmin(double a, double b) { return std::fmin(a, b); }
// f32::clamp: NaN stays NaN.
inline float rclampf(float x, float lo, float hi) {
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
inline double rclamp(double x, double lo, double hi) {
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
// `x as usize` from a float: saturating, truncating, NaN -> 0.
inline size_t as_usize(double x) {
  if (!(x > 0.0)) return 0;
  if (x >= 18446744073709551615.0) return std::numeric_limits<size_t>::max();
  return (size_t)x;
}
inline uint32_t as_u32(double x) {
  if (!(x > 0.0)) return 0;
  if (x >= 4294967295.0) return 0xFFFFFFFFu;
  return (uint32_t)x;
}
inline int64_t as_isize(double x) {
  if (x != x) return 0;
  if (x >= 9223372036854775807.0) return INT64_MAX;
  if (x <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)x;
}
constexpr float F32_PI = 3.14159274101257324f;
constexpr float F32_TAU = 6.28318548202514648f;
constexpr double F64_PI = 3.14159265358979323846;
constexpr double F64_TAU = 6.28318530717958647692;

// ---- src/utils/buffer.rs:86-173 -------------------------------------------------------------
// pulp SIMD dispatch performs the same IEEE elementwise operations; max is order free.
inline void clear_buffer(float* d, size_t n) { for (size_t i = 0; i < n; ++i) d[i] = 0.0f; }          // :86-98
inline void scale_buffer(float* d, size_t n, float v) { for (size_t i = 0; i < n; ++i) d[i] = d[i] * v; }  // :100-112
inline void add_buffers(float* d, const float* s, size_t n) { for (size_t i = 0; i < n; ++i) d[i] = d[i] + s[i]; }  // :114-130
inline void copy_buffers(float* d, const float* s, size_t n) { for (size_t i = 0; i < n; ++i) d[i] = s[i]; }  // :132-148
inline float max_abs_sample(const float* b, size_t n) {  // :150-173
  float m = 0.0f;
  for (size_t i = 0; i < n; ++i) m = rmaxf(m, std::fabs(b[i]));
  return m;
}

// src/utils/buffer.rs:183-268 (only the layouts the stereo path uses are distinct cases)
inline void remap_buffer_channels(const float* in, size_t in_ch, float* out, size_t out_ch, size_t frames) {
  if (in_ch == 1 && out_ch == 2) {
    for (size_t f = 0; f < frames; ++f) { out[2 * f] = in[f]; out[2 * f + 1] = in[f]; }
  } else if (in_ch == 2 && out_ch == 1) {
    for (size_t f = 0; f < frames; ++f) out[f] = (in[2 * f] + in[2 * f + 1]) / 2.0f;
  } else if (in_ch == out_ch) {
    copy_buffers(out, in, frames * in_ch);
  } else if (in_ch == 1) {
    for (size_t f = 0; f < frames; ++f) {
      out[f * out_ch] = in[f]; out[f * out_ch + 1] = in[f];
      for (size_t c = 2; c < out_ch; ++c) out[f * out_ch + c] = 0.0f;
    }
  } else if (out_ch == 1) {
    for (size_t f = 0; f < frames; ++f) out[f] = (in[f * in_ch] + in[f * in_ch + 1]) / 2.0f;
  } else {
    for (size_t f = 0; f < frames; ++f) {
      out[f * out_ch] = in[f * in_ch]; out[f * out_ch + 1] = in[f * in_ch + 1];
      for (size_t c = 2; c < out_ch; ++c) out[f * out_ch + c] = 0.0f;
    }
  }
}

// src/utils/buffer.rs:499-610
// Test hook: the floor()-derived ring indices of the delay-line reads, in call order, while a log is armed
// (po_index_log_begin / po_index_log_end): ReverbDelayLine::get logs read_1 per channel, InterpolatedDelayLine::process logs read_idx1.
inline std::vector<int32_t>*& index_log() { static thread_local std::vector<int32_t>* log = nullptr; return log; }
inline void log_index(size_t v) { if (index_log()) index_log()->push_back((int32_t)v); }
// Test hook: where the reference's output is discontinuous in its input. CompressorEffect's gain computer has no branch for an envelope exactly on
// the knee's upper edge (compressor.rs:258-270; tests/test_oracle_graph.py): a one-frame click that an implementation whose upstream arithmetic
// differs in the last place reproduces a frame earlier, later or not at all. While a log is armed (po_knee_log_begin / po_knee_log_end) every
// frame whose envelope comes within 64 ulps of that edge is logged with its sample time — the mixers publish their chunk's time in front of
// process_effects — so a parity test can tell such a click from a real difference.
inline uint64_t& chunk_time_now() { static thread_local uint64_t t = 0; return t; }
inline std::vector<uint64_t>*& knee_log() { static thread_local std::vector<uint64_t>* log = nullptr; return log; }
inline int32_t f32_ordered_bits(float x) { int32_t b; std::memcpy(&b, &x, 4); return b < 0 ? (int32_t)0x80000000 - b : b; }
inline void log_knee_edge(float envelope, float edge, uint64_t frame) {
  if (!knee_log()) return;
  const int64_t d = (int64_t)f32_ordered_bits(envelope) - (int64_t)f32_ordered_bits(edge);
  if (d >= -64 && d <= 64) knee_log()->push_back(frame);
}

struct TempBuffer {
  std::vector<float> buffer;
  size_t start = 0, end = 0;
  explicit TempBuffer(size_t capacity = 0) : buffer(capacity, 0.0f) {}
  bool is_empty() const { return start >= end; }
  size_t len() const { return end - start; }
  size_t capacity() const { return buffer.size(); }
  float* get() { return buffer.data() + start; }
  void set_range(size_t s, size_t e) { assert(s <= e && e <= capacity()); start = s; end = e; }
  void reset_range() { start = 0; end = buffer.size(); }
  void clear_range() { start = 0; end = 0; }
  size_t copy_to(float* other, size_t other_len) {
    size_t n = std::min(other_len, len());
    copy_buffers(other, get(), n);
    return n;
  }
  size_t copy_from(const float* other, size_t other_len) {
    size_t n = std::min(other_len, len());
    copy_buffers(get(), other, n);
    return n;
  }
  void consume(size_t samples) { assert(start + samples <= end); start += samples; }
};

// ---- src/utils.rs:26-62 ---------------------------------------------------------------------
constexpr float MINUS_INF_IN_DB = -200.0f;
inline float linear_to_db(float value) {  // :26-36
  const float LIN_TO_DB_FACTOR = 20.0f / 2.30258509299404568402f;
  if (value < 0.0f || value != value) return std::numeric_limits<float>::quiet_NaN();
  else if (value == 1.0f) return 0.0f;
  else if (value > 1e-12f) return std::log(value) * LIN_TO_DB_FACTOR;
  return MINUS_INF_IN_DB;
}
inline float db_to_linear(float value) {  // :41-51
  const float DB_TO_LIN_FACTOR = 2.30258509299404568402f / 20.0f;
  if (value != value) return std::numeric_limits<float>::quiet_NaN();
  else if (value == 0.0f) return 1.0f;
  else if (value > MINUS_INF_IN_DB) return std::exp(value * DB_TO_LIN_FACTOR);
  return 0.0f;
}
inline void panning_factors(float pan_factor, float& left, float& right) {  // :56-62
  const float POWER = 0.707106781186547524400844362104849039f;
  float normalized = (rclampf(pan_factor, -1.0f, 1.0f) + 1.0f) / 2.0f;
  left = std::sqrt(1.0f - normalized) / POWER;
  right = std::sqrt(normalized) / POWER;
}

// ---- src/utils/smoothing.rs -----------------------------------------------------------------
constexpr float F32_EPSILON = 1.1920929e-07f;
constexpr uint32_t UNINITIALIZED_SAMPLE_RATE = 66666;

// Common interface: current, target, next, need_ramp, ramp, init, set_target, set_sample_rate.
struct ExponentialSmoothedValue {  // :128-247
  float current_ = 0.0f, target_ = 0.0f, inertia_ = 1.0f / 256.0f, sample_rate_comp = 1.0f;
  ExponentialSmoothedValue() : ExponentialSmoothedValue(0.0f, UNINITIALIZED_SAMPLE_RATE) {}  // Default :238-242
  ExponentialSmoothedValue(float value, uint32_t sample_rate)
      : current_(value), target_(value), inertia_(1.0f / 256.0f), sample_rate_comp(44100.0f / (float)sample_rate) {}
  static ExponentialSmoothedValue from_f32(float v) { return ExponentialSmoothedValue(v, UNINITIALIZED_SAMPLE_RATE); }
  ExponentialSmoothedValue with_inertia(float i) const { auto s = *this; s.inertia_ = i; return s; }
  float current() const { return current_; }
  float target() const { return target_; }
  bool need_ramp() const {  // :198-206
    const float EPSILON = F32_EPSILON * 100.0f;
    float inertia_add = (target_ - current_) * inertia_ * sample_rate_comp;
    return std::fabs(inertia_add) > EPSILON;
  }
  void ramp() { current_ += (target_ - current_) * inertia_ * sample_rate_comp; }  // :208-214
  float next() { if (need_ramp()) { ramp(); return current_; } return target_; }    // :21-28
  void init(float a) { target_ = a; current_ = a; }
  void set_target(float t) { target_ = t; if (!need_ramp()) current_ = target_; }   // :221-226
  void set_sample_rate(uint32_t sr) { sample_rate_comp = 44100.0f / (float)sr; }
};

struct LinearSmoothedValue {  // :249-418
  float current_ = 0.0f, target_ = 0.0f, step_ = 0.01f, current_step = 0.0f;
  uint32_t num_pending_steps = 0;
  float sample_rate_comp = 1.0f;
  LinearSmoothedValue() : LinearSmoothedValue(0.0f, UNINITIALIZED_SAMPLE_RATE) {}
  LinearSmoothedValue(float value, uint32_t sample_rate)
      : current_(value), target_(value), step_(0.01f), current_step(0.0f), num_pending_steps(0),
        sample_rate_comp(44100.0f / (float)sample_rate) {}
  static LinearSmoothedValue from_f32(float v) { return LinearSmoothedValue(v, UNINITIALIZED_SAMPLE_RATE); }
  LinearSmoothedValue with_step(float s) const { auto v = *this; v.set_step(s); return v; }
  float step() const { return step_; }
  void set_step(float step) {  // :294-310
    step_ = step;
    current_step = (current_ > target_) ? -step_ * sample_rate_comp : step_ * sample_rate_comp;
    float pending_steps = (target_ - current_) / current_step;
    num_pending_steps = as_u32(rmaxf(std::round(pending_steps), 0.0f));
    if (num_pending_steps == 0) current_ = target_;
  }
  void set_target_with_duration(float target, bool has_duration, uint32_t dur_samples) {  // :312-341
    target_ = target;
    if (current_ == target_) {
      num_pending_steps = 0;
    } else {
      if (has_duration) {
        num_pending_steps = dur_samples;
        current_step = (target_ - current_) / (float)dur_samples;
        step_ = std::fabs(current_step);
      } else {
        current_step = (current_ > target_) ? -step_ * sample_rate_comp : step_ * sample_rate_comp;
        float pending_steps = (target_ - current_) / current_step;
        num_pending_steps = as_u32(rmaxf(std::round(pending_steps), 0.0f));
      }
      if (num_pending_steps == 0) current_ = target_;
    }
  }
  float current() const { return current_; }
  float target() const { return target_; }
  bool need_ramp() const { return num_pending_steps > 0; }
  void ramp() {  // :370-382
    if (num_pending_steps > 0) {
      current_ += current_step;
      num_pending_steps -= 1;
      if (num_pending_steps == 0) current_ = target_;
    }
  }
  float next() { if (need_ramp()) { ramp(); return current_; } return target_; }
  void init(float a) { target_ = a; current_ = a; num_pending_steps = 0; }
  void set_target(float t) { set_target_with_duration(t, false, 0); }
  void set_sample_rate(uint32_t sr) {  // :394-401
    sample_rate_comp = 44100.0f / (float)sr;
    current_step = (current_ > target_) ? -step_ * sample_rate_comp : step_ * sample_rate_comp;
  }
};

struct SpringSmoothedValue {  // :424-552
  float current_ = 0.0f, velocity_ = 0.0f, target_ = 0.0f, omega_ = 5.5f / 4410.0f, sample_rate_comp = 1.0f;
  SpringSmoothedValue() : SpringSmoothedValue(0.0f, UNINITIALIZED_SAMPLE_RATE) {}
  SpringSmoothedValue(float value, uint32_t sample_rate)
      : current_(value), velocity_(0.0f), target_(value), omega_(5.5f / (float)4410),
        sample_rate_comp(44100.0f / (float)sample_rate) {}
  static SpringSmoothedValue from_f32(float v) { SpringSmoothedValue s; s.init(v); return s; }  // :544-550
  SpringSmoothedValue with_duration(size_t d) const { auto s = *this; s.omega_ = 5.5f / (float)d; return s; }
  float current() const { return current_; }
  float target() const { return target_; }
  float velocity() const { return velocity_; }
  size_t duration() const { return as_usize(std::round(5.5f / omega_)); }
  bool need_ramp() const {  // :499-506
    const float EPSILON = F32_EPSILON * 100.0f;
    return std::fabs(velocity_) > EPSILON || std::fabs(target_ - current_) > EPSILON;
  }
  void ramp() {  // :508-518
    float omega = omega_ * sample_rate_comp;
    float k = omega * omega;
    float d = 2.0f * omega;
    velocity_ += (target_ - current_) * k - velocity_ * d;
    current_ += velocity_;
  }
  float next() { if (need_ramp()) { ramp(); return current_; } return target_; }
  void init(float v) { current_ = v; velocity_ = 0.0f; target_ = v; }
  void set_target(float v) { target_ = v; }
  void set_sample_rate(uint32_t sr) { sample_rate_comp = 44100.0f / (float)sr; }
};

// src/utils/smoothing.rs:60-71
template <class S>
inline void apply_smoothed_gain(float* buffer, size_t n, S& smoothed) {
  if (smoothed.need_ramp()) {
    for (size_t i = 0; i < n; ++i) buffer[i] *= smoothed.next();
  } else {
    float gain = smoothed.target();
    if (std::fabs(1.0f - gain) > 0.000001f) scale_buffer(buffer, n, gain);
  }
}
// src/utils/smoothing.rs:74-122
template <class S>
inline void apply_smoothed_panning(float* buffer, size_t n, size_t channel_count, S& smoothed) {
  if (channel_count >= 2) {
    if (smoothed.need_ramp()) {
      for (size_t f = 0; f + channel_count <= n; f += channel_count) {
        float l, r;
        panning_factors(smoothed.next(), l, r);
        buffer[f] *= l;
        buffer[f + 1] *= r;
      }
    } else {
      float pan = smoothed.target();
      if (std::fabs(pan) > 0.000001f) {
        float l, r;
        panning_factors(pan, l, r);
        for (size_t f = 0; f + channel_count <= n; f += channel_count) {
          buffer[f] *= l;
          buffer[f + 1] *= r;
        }
      }
    }
  }
}

// ---- src/utils/fader.rs ----------------------------------------------------------------------
enum class FaderState { Stopped, IsRunning, Finished };
struct VolumeFader {  // :27-122
  FaderState state = FaderState::Stopped;
  float current_volume = 1.0f, target_volume_ = 1.0f, inertia = 1.0f;
  size_t channel_count = 2;
  uint32_t sample_rate = 44100;
  VolumeFader() {}
  VolumeFader(size_t ch, uint32_t sr) : channel_count(ch), sample_rate(sr) {}
  float target_volume() const { return target_volume_; }
  void start_fade_in(float seconds) {  // :59-65
    if (state == FaderState::IsRunning) start(current_volume, 1.0f, seconds);
    else start(0.0f, 1.0f, seconds);
  }
  void start_fade_out(float seconds) {  // :67-73
    if (state == FaderState::IsRunning) start(current_volume, 0.0f, seconds);
    else start(1.0f, 0.0f, seconds);
  }
  void start(float from, float to, float duration_secs_f32) {  // :75-91 (Duration::as_secs_f32)
    if (duration_secs_f32 == 0.0f) {
      current_volume = to; target_volume_ = to; state = FaderState::Finished;
    } else {
      state = FaderState::IsRunning;
      current_volume = from;
      target_volume_ = to;
      const float LN100 = 4.605f;
      float samples_duration = (float)sample_rate * duration_secs_f32 / LN100;
      inertia = 1.0f - std::exp(-1.0f / samples_duration);
    }
  }
  void reset() { state = FaderState::Stopped; current_volume = 1.0f; target_volume_ = 1.0f; }
  void process(float* output, size_t n) {  // :103-122
    if (state != FaderState::IsRunning) {
      if (target_volume_ != 1.0f) scale_buffer(output, n, target_volume_);
    } else {
      for (size_t f = 0; f + channel_count <= n; f += channel_count) {
        current_volume += (target_volume_ - current_volume) * inertia;
        for (size_t c = 0; c < channel_count; ++c) output[f + c] *= current_volume;
      }
      if (std::fabs(current_volume - target_volume_) < 0.0001f) state = FaderState::Finished;
    }
  }
};

}  // namespace po
