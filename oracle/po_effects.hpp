// ORACLE — TEST INFRASTRUCTURE ONLY (see po_utils.hpp header). Parity unpinned: the reference has
// no tests for any effect, so these restatements are checked only by transcription
// review, analytic known answers (tests/test_oracle_analytic.py) and the reference KATs of the
// primitives they are built from.
//
// CPU restatement of src/effect/*.rs of emuell/phonic v0.16.0.
#pragma once
#include "po_dsp.hpp"
#include "po_params.hpp"

namespace po {

struct Effect {  // trait Effect, src/effect.rs:86-215
  virtual ~Effect() {}
  virtual const char* name() const = 0;
  virtual size_t weight() const = 0;
  virtual bool initialize(uint32_t sample_rate, size_t channel_count, size_t max_frames) = 0;
  virtual void process_started() {}
  virtual void process_stopped() {}
  virtual void process(float* output, size_t n_samples) = 0;
  // None -> {false, _}; Some(n) -> {true, n}; usize::MAX = infinite
  virtual bool process_tail(size_t& frames) const = 0;
  virtual bool process_parameter_update(uint32_t id, const ParamUpdate& u) = 0;
  virtual bool process_reset_message() { return false; }  // effects without messages: error
  // construction-time raw value ("with_parameters" constructors); false if unknown id
  virtual bool init_param(uint32_t id, float raw) = 0;
  virtual void finish_init_params() {}
};

constexpr size_t USIZE_MAX = std::numeric_limits<size_t>::max();

// ---- src/effect/gain.rs ----------------------------------------------------------------------
struct GainEffect : Effect {
  static constexpr float MIN_DB = -60.0f, MAX_DB = 24.0f;
  SmoothedParameterValue<> gain{FloatParameter{fourcc("gain"), 0.000001f, 15.848932f, 1.0f, {Scaling::Decibel, MIN_DB, MAX_DB}}};
  EnumParameterValue dc_filter_mode{EnumParameter{fourcc("dcfm"), 4, 0}};  // Off, Slow, Default, Fast
  std::vector<DcFilter> dc_filters;
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  const char* name() const override { return "Gain"; }
  size_t weight() const override { return 1; }
  static bool to_dc_mode(int v, DcMode& m) {  // :32-41
    if (v == 0) return false;
    m = v == 1 ? DcMode::Slow : (v == 2 ? DcMode::Default : DcMode::Fast);
    return true;
  }
  bool init_param(uint32_t id, float raw) override {  // with_parameters(gain_db, dc_mode) :96-102 takes dB; here raw = linear gain
    if (id == gain.description.id) { gain.init_value(raw); return true; }
    if (id == dc_filter_mode.description.id) { dc_filter_mode.set_value((int)raw); return true; }
    return false;
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :123-141
    sample_rate = sr; channel_count = ch;
    gain.set_sample_rate(sr);
    DcMode m = DcMode::Default;
    to_dc_mode(dc_filter_mode.value(), m);
    dc_filters.assign(ch, DcFilter(sr, m));
    return true;
  }
  void process(float* out, size_t n) override {  // :143-166
    if (dc_filter_mode.value() != 0) {
      for (size_t c = 0; c < channel_count; ++c)
        for (size_t i = c; i < n; i += channel_count) out[i] = (float)dc_filters[c].process_sample((double)out[i]);
    }
    if (gain.value_need_ramp()) {
      for (size_t f = 0; f + channel_count <= n; f += channel_count) {
        float g = gain.next_value();
        for (size_t c = 0; c < channel_count; ++c) out[f + c] *= g;
      }
    } else {
      scale_buffer(out, n, gain.target_value());
    }
  }
  bool process_tail(size_t& frames) const override {  // :168-175
    DcMode m;
    if (to_dc_mode(dc_filter_mode.value(), m)) frames = (size_t)sample_rate / (size_t)dc_mode_hz(m);
    else frames = 0;
    return true;
  }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {  // :177-205
    if (id == gain.description.id) { gain.apply_update(u); return true; }
    if (id == dc_filter_mode.description.id) {
      dc_filter_mode.apply_update(u);
      DcMode m;
      if (to_dc_mode(dc_filter_mode.value(), m)) for (auto& f : dc_filters) f.set_mode(m, sample_rate);
      else for (auto& f : dc_filters) f.reset();
      return true;
    }
    return false;
  }
};

// ---- src/effect/pan.rs -----------------------------------------------------------------------
struct PanningEffect : Effect {
  size_t channel_count = 0;
  SmoothedParameterValue<> pan{FloatParameter{fourcc("pan "), -1.0f, 1.0f, 0.0f, {}}};
  SmoothedParameterValue<> width{FloatParameter{fourcc("wdth"), 0.0f, 2.0f, 1.0f, {}}};
  BooleanParameterValue invert_l{BooleanParameter{fourcc("invl"), false}};
  BooleanParameterValue invert_r{BooleanParameter{fourcc("invr"), false}};
  const char* name() const override { return "Panning"; }
  size_t weight() const override { return 1; }
  bool init_param(uint32_t id, float raw) override {
    if (id == pan.description.id) { pan.init_value(raw); return true; }
    if (id == width.description.id) { width.init_value(raw); return true; }
    if (id == invert_l.description.id) { invert_l.value_ = raw != 0.0f; return true; }
    if (id == invert_r.description.id) { invert_r.value_ = raw != 0.0f; return true; }
    return false;
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :88-103
    if (ch != 2) return false;
    channel_count = ch;
    pan.set_sample_rate(sr); width.set_sample_rate(sr);
    return true;
  }
  void process(float* out, size_t n) override {  // :105-158
    float inv_l = invert_l.value() ? -1.0f : 1.0f;
    float inv_r = invert_r.value() ? -1.0f : 1.0f;
    bool has_invert = inv_l < 0.0f || inv_r < 0.0f;
    bool pan_ramping = pan.value_need_ramp();
    bool width_ramping = width.value_need_ramp();
    if (!has_invert && !pan_ramping && !width_ramping && std::fabs(pan.target_value()) < 1e-6f &&
        std::fabs(width.target_value() - 1.0f) < 1e-6f)
      return;
    for (size_t f = 0; f + 2 <= n; f += 2) {
      float l = out[f] * inv_l;
      float r = out[f + 1] * inv_r;
      float w = width_ramping ? width.next_value() : width.target_value();
      if (std::fabs(w - 1.0f) > 1e-6f) {
        float mid = (l + r) * 0.5f;
        float side = (l - r) * 0.5f;
        l = mid + side * w;
        r = mid - side * w;
      }
      float p = pan_ramping ? pan.next_value() : pan.target_value();
      if (std::fabs(p) > 1e-6f) {
        float pl, pr;
        panning_factors(p, pl, pr);
        l *= pl;
        r *= pr;
      }
      out[f] = l;
      out[f + 1] = r;
    }
  }
  bool process_tail(size_t& frames) const override { frames = 0; return true; }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {
    if (id == pan.description.id) { pan.apply_update(u); return true; }
    if (id == width.description.id) { width.apply_update(u); return true; }
    if (id == invert_l.description.id) { invert_l.apply_update(u); return true; }
    if (id == invert_r.description.id) { invert_r.apply_update(u); return true; }
    return false;
  }
};

// ---- src/effect/filter.rs --------------------------------------------------------------------
struct FilterEffect : Effect {
  size_t channel_count = 0;
  uint32_t sample_rate = 0;
  std::vector<BiquadFilter> filters;
  BiquadCoefficients filter_coefficients;
  EnumParameterValue filter_type{EnumParameter{fourcc("type"), 4, 0}};  // Lowpass, Bandpass, Bandstop, Highpass
  SmoothedParameterValue<> cutoff{FloatParameter{fourcc("cuto"), 20.0f, 20000.0f, 20000.0f, {Scaling::Exponential, 2.5f, 0}}};
  SmoothedParameterValue<LinearSmoothedValue> q{FloatParameter{fourcc("fltq"), 0.001f, 4.0f, 0.707f, {}}};
  bool with_params = false;
  FilterEffect() { filter_coefficients.set(BiquadType::Lowpass, 44100, 22050.0f, 0.707f, 0.0f); }  // new() :87-101
  static BiquadType to_biquad(int t) {  // :33-41
    switch (t) { case 0: return BiquadType::Lowpass; case 1: return BiquadType::Bandpass; case 2: return BiquadType::Notch; default: return BiquadType::Highpass; }
  }
  const char* name() const override { return "Filter"; }
  size_t weight() const override { return 2; }
  bool init_param(uint32_t id, float raw) override {
    with_params = true;
    if (id == filter_type.description.id) { filter_type.set_value((int)raw); return true; }
    if (id == cutoff.description.id) { cutoff.init_value(raw); return true; }
    if (id == q.description.id) { q.init_value(raw); return true; }
    return false;
  }
  void finish_init_params() override {  // with_parameters :104-115
    if (with_params) {
      float c = rclampf(cutoff.target_value(), 20.0f, 44100.0f / 2.0f);
      filter_coefficients.set(to_biquad(filter_type.value()), 44100, c, q.target_value(), 0.0f);
    }
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :141-164
    sample_rate = sr; channel_count = ch;
    float c = rclampf(filter_coefficients.cutoff, 20.0f, (float)sample_rate / 2.0f);
    filter_coefficients.set_cutoff(c);
    filters.resize(ch);
    cutoff.set_sample_rate(sr);
    q.set_sample_rate(sr);
    return true;
  }
  void process(float* buf, size_t n) override {  // :166-201
    if (cutoff.value_need_ramp() || q.value_need_ramp()) {
      for (size_t f = 0; f + channel_count <= n; f += channel_count) {
        float c = rclampf(cutoff.next_value(), 20.0f, (float)sample_rate / 2.0f);
        float qq = q.next_value();
        filter_coefficients.set(to_biquad(filter_type.value()), sample_rate, c, qq, 0.0f);
        for (size_t ch = 0; ch < channel_count; ++ch)
          buf[f + ch] = (float)filters[ch].process_sample(filter_coefficients, (double)buf[f + ch]);
      }
    } else {
      for (size_t ch = 0; ch < channel_count; ++ch)
        for (size_t i = ch; i < n; i += channel_count) buf[i] = (float)filters[ch].process_sample(filter_coefficients, (double)buf[i]);
    }
  }
  bool process_tail(size_t& frames) const override { frames = (size_t)sample_rate / 10; return true; }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {  // :209-237
    if (id == filter_type.description.id) { filter_type.apply_update(u); filter_coefficients.set_filter_type(to_biquad(filter_type.value())); return true; }
    if (id == cutoff.description.id) { cutoff.apply_update(u); return true; }
    if (id == q.description.id) { q.apply_update(u); return true; }
    return false;
  }
};

// ---- src/effect/eq5.rs -----------------------------------------------------------------------
struct Eq5Effect : Effect {
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  SmoothedParameterValue<> gains[5];
  SmoothedParameterValue<> frequencies[5];
  SmoothedParameterValue<LinearSmoothedValue> bandwidths[5];
  BiquadCoefficients filter_coeffs[5];
  std::vector<BiquadFilter> filters;  // [ch][5]
  Eq5Effect() {  // :152-170
    static const char gid[5][5] = {"gan1", "gan2", "gan3", "gan4", "gan5"};
    static const char fid[5][5] = {"frq1", "frq2", "frq3", "frq4", "frq5"};
    static const char bid[5][5] = {"bw_1", "bw_2", "bw_3", "bw_4", "bw_5"};
    static const float fdef[5] = {100.0f, 1000.0f, 4000.0f, 8000.0f, 12000.0f};
    for (int i = 0; i < 5; ++i) {
      gains[i] = SmoothedParameterValue<>(FloatParameter{fourcc(gid[i]), -20.0f, 20.0f, 0.0f, {}});
      frequencies[i] = SmoothedParameterValue<>(FloatParameter{fourcc(fid[i]), 20.0f, 20000.0f, fdef[i], {Scaling::Exponential, 2.5f, 0}});
      float bmax = (i == 0 || i == 4) ? 1.0f : 4.0f;
      bandwidths[i] = SmoothedParameterValue<LinearSmoothedValue>(FloatParameter{fourcc(bid[i]), 0.0001f, bmax, bmax, {}})
                          .with_smoother(LinearSmoothedValue());
    }
  }
  static BiquadType band_type(int i) { return i == 0 ? BiquadType::Lowshelf : (i == 4 ? BiquadType::Highshelf : BiquadType::Bell); }
  const char* name() const override { return "Eq5"; }
  size_t weight() const override { return 3; }
  bool init_param(uint32_t id, float raw) override {
    for (int i = 0; i < 5; ++i) {
      if (id == gains[i].description.id) { gains[i].init_value(raw); return true; }
      if (id == frequencies[i].description.id) { frequencies[i].init_value(raw); return true; }
      if (id == bandwidths[i].description.id) { bandwidths[i].init_value(raw); return true; }
    }
    return false;
  }
  bool update_filter_coefficients() {  // :172-188
    for (int i = 0; i < 5; ++i) {
      float c = rclampf(frequencies[i].current_value(), 20.0f, (float)sample_rate / 2.0f);
      float qq = bandwidths[i].current_value();
      float g = gains[i].current_value();
      if (!filter_coeffs[i].set(band_type(i), sample_rate, c, qq, g)) return false;
    }
    return true;
  }
  bool ramp_filter_coefficients() {  // :190-207
    for (int i = 0; i < 5; ++i) {
      float qq = (i == 0 || i == 4) ? bandwidths[i].next_value() : 1.0f / rmaxf(bandwidths[i].next_value(), 0.001f);
      float c = rclampf(frequencies[i].next_value(), 20.0f, (float)sample_rate / 2.0f);
      float g = gains[i].next_value();
      if (!filter_coeffs[i].set(band_type(i), sample_rate, c, qq, g)) return false;
    }
    return true;
  }
  void reset() {  // :209-225
    for (auto& f : filters) f.reset();
    for (int i = 0; i < 5; ++i) {
      gains[i].init_value(gains[i].target_value());
      frequencies[i].init_value(frequencies[i].target_value());
      bandwidths[i].init_value(bandwidths[i].target_value());
    }
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :268-294
    sample_rate = sr; channel_count = ch;
    for (int i = 0; i < 5; ++i) { gains[i].set_sample_rate(sr); frequencies[i].set_sample_rate(sr); bandwidths[i].set_sample_rate(sr); }
    if (!update_filter_coefficients()) return false;
    filters.assign(ch * 5, BiquadFilter());
    reset();
    return true;
  }
  void process(float* out, size_t n) override {  // :297-326
    bool need_ramp = false;
    for (int i = 0; i < 5; ++i) need_ramp = need_ramp || frequencies[i].value_need_ramp();
    for (int i = 0; i < 5; ++i) need_ramp = need_ramp || bandwidths[i].value_need_ramp();
    for (int i = 0; i < 5; ++i) need_ramp = need_ramp || gains[i].value_need_ramp();
    size_t frame_count = n / channel_count;
    for (size_t f = 0; f < frame_count; ++f) {
      if (need_ramp) ramp_filter_coefficients();
      for (size_t ch = 0; ch < channel_count; ++ch) {
        size_t idx = f * channel_count + ch;
        float sample = out[idx];
        for (int i = 0; i < 5; ++i) sample = (float)filters[ch * 5 + i].process_sample(filter_coeffs[i], (double)sample);
        out[idx] = sample;
      }
    }
  }
  bool process_tail(size_t& frames) const override { frames = (size_t)sample_rate / 5; return true; }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {  // :334-363
    bool found = false;
    for (int i = 0; i < 5 && !found; ++i) {
      if (id == gains[i].description.id) { gains[i].apply_update(u); found = true; }
      else if (id == frequencies[i].description.id) { frequencies[i].apply_update(u); found = true; }
      else if (id == bandwidths[i].description.id) { bandwidths[i].apply_update(u); found = true; }
    }
    if (!found) return false;
    update_filter_coefficients();
    return true;
  }
};

// ---- src/effect/delay.rs ---------------------------------------------------------------------
inline double delay_saturate(double input, float drive) {  // :70-79
  if (drive < 0.001f) return input;
  double gain = 1.0 + (double)drive * 4.0;
  double x = input * gain;
  double x2 = x * x;
  double output = x * (27.0 + x2) / (27.0 + 9.0 * x2);
  return output / std::sqrt(gain);
}

struct DelayEffect : Effect {
  static constexpr float MAX_DELAY_MS = 4000.0f, MAX_LFO_TIME_MOD_MS = 50.0f, FILTER_RESONANCE = 0.302f;
  uint32_t sample_rate = 0;
  EnumParameterValue mode{EnumParameter{fourcc("mode"), 2, 0}};  // Stereo, Ping Pong
  SmoothedParameterValue<SpringSmoothedValue> delay_time =
      SmoothedParameterValue<SpringSmoothedValue>(FloatParameter{fourcc("dlay"), 1.0f, MAX_DELAY_MS, 375.0f, {}})
          .with_smoother(SpringSmoothedValue().with_duration(20000));
  SmoothedParameterValue<> feedback{FloatParameter{fourcc("fdbk"), 0.0f, 1.0f, 0.5f, {}}};
  SmoothedParameterValue<> filter_cutoff{FloatParameter{fourcc("cuto"), 20.0f, 20000.0f, 6000.0f, {Scaling::Exponential, 2.5f, 0}}};
  EnumParameterValue filter_type{EnumParameter{fourcc("ftyp"), 3, 0}};  // Lowpass, Highpass, Bandpass
  SmoothedParameterValue<> drive{FloatParameter{fourcc("driv"), 0.0f, 1.0f, 0.0f, {}}};
  SmoothedParameterValue<> wet_mix{FloatParameter{fourcc("wet_"), 0.0f, 1.0f, 0.5f, {}}};
  SmoothedParameterValue<> stereo_width{FloatParameter{fourcc("wdth"), 0.0f, 1.0f, 0.5f, {}}};
  SmoothedParameterValue<> lfo_rate{FloatParameter{fourcc("lfor"), 0.01f, 10.0f, 1.0f, {Scaling::Exponential, 2.0f, 0}}};
  EnumParameterValue lfo_shape{EnumParameter{fourcc("lfos"), 7, 0}};
  SmoothedParameterValue<> lfo_depth_time{FloatParameter{fourcc("lfdt"), -1.0f, 1.0f, 0.0f, {}}};
  SmoothedParameterValue<> lfo_depth_feedback{FloatParameter{fourcc("ldfb"), -1.0f, 1.0f, 0.0f, {}}};
  SmoothedParameterValue<> lfo_depth_filter{FloatParameter{fourcc("lfdf"), -1.0f, 1.0f, 0.0f, {}}};
  InterpolatedDelayLine<1> delay_left, delay_right;
  Lfo lfo;
  SmallRng lfo_seed;  // state the LFO's generator starts from in initialize (pg_effect_init::lfo_rng_state; the reference: from_os_rng)
  SvfCoefficients filter_coefficients;
  SvfFilter filter_left, filter_right;
  DcFilter dc_left, dc_right;
  float feedback_left = 0.0f, feedback_right = 0.0f;

  const char* name() const override { return "Delay"; }
  size_t weight() const override { return 3; }
  static SvfType to_svf(int v) { return v == 0 ? SvfType::Lowpass : (v == 1 ? SvfType::Highpass : SvfType::Bandpass); }
  bool init_param(uint32_t id, float raw) override {
    if (id == mode.description.id) { mode.set_value((int)raw); return true; }
    if (id == delay_time.description.id) { delay_time.init_value(raw); return true; }
    if (id == feedback.description.id) { feedback.init_value(raw); return true; }
    if (id == filter_type.description.id) { filter_type.set_value((int)raw); return true; }
    if (id == filter_cutoff.description.id) { filter_cutoff.init_value(raw); return true; }
    if (id == drive.description.id) { drive.init_value(raw); return true; }
    if (id == wet_mix.description.id) { wet_mix.init_value(raw); return true; }
    if (id == stereo_width.description.id) { stereo_width.init_value(raw); return true; }
    if (id == lfo_rate.description.id) { lfo_rate.init_value(raw); return true; }
    if (id == lfo_shape.description.id) { lfo_shape.set_value((int)raw); return true; }
    if (id == lfo_depth_time.description.id) { lfo_depth_time.init_value(raw); return true; }
    if (id == lfo_depth_feedback.description.id) { lfo_depth_feedback.init_value(raw); return true; }
    if (id == lfo_depth_filter.description.id) { lfo_depth_filter.init_value(raw); return true; }
    return false;
  }
  void reset() {  // :213-223
    delay_left.flush(); delay_right.flush();
    filter_left.reset(); filter_right.reset();
    dc_left.reset(); dc_right.reset();
    lfo.reset();
    feedback_left = 0.0f; feedback_right = 0.0f;
  }
  static float process_feedback(SvfFilter& filter, const SvfCoefficients& co, DcFilter& dc, float delayed, float drv) {  // :226-237
    double filtered = filter.process_sample(co, (double)delayed);
    double saturated = delay_saturate(filtered, drv);
    float clean = (float)dc.process_sample(saturated);
    return rclampf(clean, -4.0f, 4.0f);
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :273-332
    sample_rate = sr;
    if (ch != 2) return false;
    delay_time.set_sample_rate(sr); feedback.set_sample_rate(sr); filter_cutoff.set_sample_rate(sr);
    drive.set_sample_rate(sr); wet_mix.set_sample_rate(sr); stereo_width.set_sample_rate(sr);
    lfo_rate.set_sample_rate(sr); lfo_depth_time.set_sample_rate(sr); lfo_depth_feedback.set_sample_rate(sr);
    lfo_depth_filter.set_sample_rate(sr);
    size_t max_delay_samples = as_usize(std::ceil((MAX_DELAY_MS + MAX_LFO_TIME_MOD_MS) * (float)sr / 1000.0f));
    delay_left = InterpolatedDelayLine<1>(max_delay_samples + 4);
    delay_right = InterpolatedDelayLine<1>(max_delay_samples + 4);
    float c = rclampf(filter_cutoff.target_value(), 20.0f, (float)sr / 2.0f);
    filter_coefficients = SvfCoefficients();
    if (!filter_coefficients.set(to_svf(filter_type.value()), sr, c, FILTER_RESONANCE)) return false;
    lfo = Lfo(sr, (double)lfo_rate.target_value(), (LfoWaveform)lfo_shape.value(), lfo_seed);  // `Lfo::new` :318-322 with the seeded generator
    dc_left = DcFilter(sr, DcMode::Default);
    dc_right = DcFilter(sr, DcMode::Default);
    feedback_left = 0.0f; feedback_right = 0.0f;
    return true;
  }
  void process(float* out, size_t n) override {  // :334-454
    float srf = (float)sample_rate;
    int md = mode.value();
    for (size_t f = 0; f + 2 <= n; f += 2) {
      float left_input = out[f], right_input = out[f + 1];
      float lfo_val = lfo.run();
      if (lfo_rate.value_need_ramp()) {
        float rate = lfo_rate.next_value();
        lfo.set_rate(sample_rate, (double)rate);
      }
      float base_delay_ms = delay_time.next_value();
      float time_mod_ms = lfo_val * lfo_depth_time.next_value() * MAX_LFO_TIME_MOD_MS;
      float delay_ms = rmaxf(base_delay_ms + time_mod_ms, 1.0f);
      float delay_samples = delay_ms * 0.001f * srf;
      float filter_depth = lfo_depth_filter.next_value();
      float filter_mod = std::pow(2.0f, lfo_val * filter_depth * 2.0f);
      float cutoff = rclampf(filter_cutoff.next_value() * filter_mod, 20.0f, (float)sample_rate / 2.0f);
      filter_coefficients.set(to_svf(filter_type.value()), sample_rate, cutoff, FILTER_RESONANCE);
      float base_feedback = feedback.next_value();
      float feedback_depth = lfo_depth_feedback.next_value();
      float fb = rclampf(base_feedback + lfo_val * feedback_depth * (1.0f - std::fabs(base_feedback)), 0.0f, 0.999f);
      float drv = drive.next_value();
      float wet = wet_mix.next_value();
      float width = stereo_width.next_value();
      float wet_l, wet_r;
      if (md == 0) {
        float l_in = left_input + feedback_left * fb;
        float delayed_l;
        delay_left.process(&l_in, 0.0f, delay_samples, &delayed_l);
        float clean_l = process_feedback(filter_left, filter_coefficients, dc_left, delayed_l, drv);
        feedback_left = clean_l;
        float r_in = right_input + feedback_right * fb;
        float delayed_r;
        delay_right.process(&r_in, 0.0f, delay_samples, &delayed_r);
        float clean_r = process_feedback(filter_right, filter_coefficients, dc_right, delayed_r, drv);
        feedback_right = clean_r;
        wet_l = clean_l; wet_r = clean_r;
      } else {
        float mono_in = (left_input + right_input) * 0.5f;
        float l_in = mono_in + feedback_right * fb;
        float delayed_l;
        delay_left.process(&l_in, 0.0f, delay_samples, &delayed_l);
        float clean_l = process_feedback(filter_left, filter_coefficients, dc_left, delayed_l, drv);
        float r_in = feedback_left * fb;
        float delayed_r;
        delay_right.process(&r_in, 0.0f, delay_samples, &delayed_r);
        float clean_r = process_feedback(filter_right, filter_coefficients, dc_right, delayed_r, drv);
        feedback_left = clean_l;
        feedback_right = clean_r;
        wet_l = clean_l; wet_r = clean_r;
      }
      float dry_gain = rminf((1.0f - wet) * 2.0f, 1.0f);
      float wet_gain = rminf(wet * 2.0f, 1.0f);
      float out_l = left_input * dry_gain + wet_l * wet_gain;
      float out_r = right_input * dry_gain + wet_r * wet_gain;
      float mid = (out_l + out_r) * 0.5f;
      float side = (out_l - out_r) * 0.5f;
      out[f] = mid + side * width;
      out[f + 1] = mid - side * width;
    }
  }
  bool process_tail(size_t& frames) const override {  // :456-475
    if (drive.target_value() > 0.0f) return false;
    double delay_ms = (double)(delay_time.target_value() + MAX_LFO_TIME_MOD_MS);
    double fb = (double)std::fabs(feedback.target_value());
    if (fb >= 0.9999) frames = USIZE_MAX;
    else if (fb < 0.001) frames = as_usize(std::ceil(delay_ms * (double)sample_rate / 1000.0));
    else {
      const double SILENCE = 0.001;
      double delay_samples = delay_ms * (double)sample_rate / 1000.0;
      double decay_samples = delay_samples + delay_samples * std::log10(SILENCE) / std::log10(fb);
      frames = std::max(as_usize(std::ceil(decay_samples)), (size_t)1);
    }
    return true;
  }
  bool process_reset_message() override { reset(); return true; }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {  // :490-520
    if (id == mode.description.id) mode.apply_update(u);
    else if (id == delay_time.description.id) delay_time.apply_update(u);
    else if (id == feedback.description.id) feedback.apply_update(u);
    else if (id == filter_type.description.id) filter_type.apply_update(u);
    else if (id == filter_cutoff.description.id) filter_cutoff.apply_update(u);
    else if (id == drive.description.id) drive.apply_update(u);
    else if (id == wet_mix.description.id) wet_mix.apply_update(u);
    else if (id == stereo_width.description.id) stereo_width.apply_update(u);
    else if (id == lfo_rate.description.id) lfo_rate.apply_update(u);
    else if (id == lfo_shape.description.id) { lfo_shape.apply_update(u); lfo.set_waveform((LfoWaveform)lfo_shape.value()); }
    else if (id == lfo_depth_time.description.id) lfo_depth_time.apply_update(u);
    else if (id == lfo_depth_feedback.description.id) lfo_depth_feedback.apply_update(u);
    else if (id == lfo_depth_filter.description.id) lfo_depth_filter.apply_update(u);
    else return false;
    return true;
  }
};

// ---- src/effect/reverb.rs --------------------------------------------------------------------
struct ReverbDelayLine {  // :518-615 (CHANNELS = 2)
  std::vector<double> buffer;  // [(size+1)][2]
  size_t count = 1, delay = 1;
  double feedback[2] = {0.0, 0.0};
  double depth = 0.0;
  double vib_phase[2] = {0.0, 0.0};
  ReverbDelayLine() {}
  ReverbDelayLine(size_t size, double depth_, double p0, double p1) : buffer((size + 1) * 2, 0.0), depth(depth_) { vib_phase[0] = p0; vib_phase[1] = p1; }
  size_t frames() const { return buffer.size() / 2; }
  void flush() { std::fill(buffer.begin(), buffer.end(), 0.0); }
  void get(double vib_depth, double blend, double* output) const {  // :554-586
    for (int ch = 0; ch < 2; ++ch) {
      double offset = (std::sin(vib_phase[ch]) + 1.0) * vib_depth;
      double working = (double)count + offset;
      double w_floor = std::floor(working);
      double w_frac = working - w_floor;
      size_t w_int = as_usize(w_floor);
      size_t read_1 = w_int;
      if (read_1 > delay) read_1 -= delay + 1;
      size_t read_2 = w_int + 1;
      if (read_2 > delay) read_2 -= delay + 1;
      log_index(read_1);
      double val1 = buffer[read_1 * 2 + ch];
      double val2 = buffer[read_2 * 2 + ch];
      double interpol = val1 * (1.0 - w_frac) + val2 * w_frac;
      interpol = (1.0 - blend) * interpol + (val1 * blend);
      output[ch] = interpol;
    }
  }
  void set(const double* values) { for (int ch = 0; ch < 2; ++ch) buffer[count * 2 + ch] = values[ch] + feedback[ch]; }  // :588-594
  void step(double speed) {  // :596-604
    count += 1;
    if (count > delay) count = 0;
    for (int ch = 0; ch < 2; ++ch) vib_phase[ch] += depth * speed;
  }
  void set_delay(size_t d) { delay = std::min(d, frames() - 1); }  // :606-614
};

struct ReverbEffect : Effect {
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  SmoothedParameterValue<LinearSmoothedValue> room_size =
      SmoothedParameterValue<LinearSmoothedValue>(FloatParameter{fourcc("room"), 0.0f, 1.0f, 0.6f, {}})
          .with_smoother(LinearSmoothedValue().with_step(0.01f));
  SmoothedParameterValue<> wet{FloatParameter{fourcc("wet "), 0.0f, 1.0f, 0.35f, {}}};
  BiquadCoefficients biquad_a_coefficients, biquad_b_coefficients, biquad_c_coefficients;
  BiquadFilter biquad_a_l, biquad_a_r, biquad_b_l, biquad_b_r, biquad_c_l, biquad_c_r;
  uint32_t fpd_l = 16386, fpd_r = 16386;
  ReverbDelayLine line[8];  // a..h
  AllpassDelayLine<2> ap[4];  // i..l
  DelayLine<2> m;
  // seeds: the reference draws them from rand::rng() (:95-103,532-538); explicit here
  ReverbEffect(uint32_t fl, uint32_t fr, const double* vib16) : fpd_l(fl), fpd_r(fr) {
    static const size_t sizes[8] = {8111, 7511, 7311, 6911, 6311, 6111, 5511, 4911};                      // :106-113
    static const double depths[8] = {0.003251, 0.002999, 0.002917, 0.002749, 0.002503, 0.002423, 0.002146, 0.002088};  // :137-144
    for (int i = 0; i < 8; ++i) line[i] = ReverbDelayLine(sizes[i], depths[i], vib16[i * 2], vib16[i * 2 + 1]);
    static const size_t apsizes[4] = {4511, 4311, 3911, 3311};  // :114-117
    for (int i = 0; i < 4; ++i) ap[i] = AllpassDelayLine<2>(apsizes[i]);
    m = DelayLine<2>(3111);  // :118
  }
  const char* name() const override { return "Reverb"; }
  size_t weight() const override { return 5; }
  bool init_param(uint32_t id, float raw) override {
    if (id == room_size.description.id) { room_size.init_value(raw); return true; }
    if (id == wet.description.id) { wet.init_value(raw); return true; }
    return false;
  }
  void update_filter_coefs(float cutoff) {  // :161-194
    cutoff = rclampf(cutoff, 20.0f, (float)sample_rate / 2.0f);
    if (biquad_a_coefficients.set(BiquadType::Lowpass, sample_rate, cutoff, 1.618034f, 0.0f))
      if (biquad_b_coefficients.set(BiquadType::Lowpass, sample_rate, cutoff, 0.618034f, 0.0f))
        biquad_c_coefficients.set(BiquadType::Lowpass, sample_rate, cutoff, 0.5f, 0.0f);
  }
  size_t update_delay_sizes(double size) {  // :196-213
    static const double k[8] = {79.0, 73.0, 71.0, 67.0, 61.0, 59.0, 53.0, 47.0};
    for (int i = 0; i < 8; ++i) line[i].set_delay(as_usize(k[i] * size));
    static const double ka[4] = {43.0, 41.0, 37.0, 31.0};
    for (int i = 0; i < 4; ++i) ap[i].set_delay(as_usize(ka[i] * size));
    return as_usize(29.0 * size);
  }
  inline void process_frame(float* frame, double blend, double regen, size_t predelay, double wetd) {  // :217-369
    const double vib_speed = 0.1, vib_depth = 7.0;
    double input_l = (double)frame[0], input_r = (double)frame[1];
    if (std::fabs(input_l) < 1.18e-23) input_l = (double)fpd_l * 1.18e-17;
    if (std::fabs(input_r) < 1.18e-23) input_r = (double)fpd_r * 1.18e-17;
    double dry_l = input_l, dry_r = input_r;
    double in2[2] = {input_l, input_r}, pd[2];
    m.process(predelay, in2, pd);
    input_l = pd[0]; input_r = pd[1];
    input_l = biquad_a_l.process_sample(biquad_a_coefficients, input_l);
    input_r = biquad_a_r.process_sample(biquad_a_coefficients, input_r);
    input_l *= wetd; input_r *= wetd;
    input_l = std::sin(input_l); input_r = std::sin(input_r);
    double in_ap[2] = {input_l, input_r}, out_i[2], out_j[2], out_k[2], out_l[2];
    ap[0].process(in_ap, out_i);
    ap[1].process(out_i, out_j);
    ap[2].process(out_j, out_k);
    ap[3].process(out_k, out_l);
    line[0].set(out_l); line[1].set(out_k); line[2].set(out_j); line[3].set(out_i);  // :275-282
    line[4].set(out_i); line[5].set(out_j); line[6].set(out_k); line[7].set(out_l);
    for (int i = 0; i < 8; ++i) line[i].step(vib_speed);
    double g[8][2];
    for (int i = 0; i < 8; ++i) line[i].get(vib_depth, blend, g[i]);
    for (int ch = 0; ch < 2; ++ch) {  // :303-319
      double a = g[0][ch], b = g[1][ch], c = g[2][ch], d = g[3][ch], e = g[4][ch], f = g[5][ch], gg = g[6][ch], h = g[7][ch];
      line[0].feedback[ch] = (a - (b + c + d)) * regen;
      line[1].feedback[ch] = (b - (a + c + d)) * regen;
      line[2].feedback[ch] = (c - (a + b + d)) * regen;
      line[3].feedback[ch] = (d - (a + b + c)) * regen;
      line[4].feedback[ch] = (e - (f + gg + h)) * regen;
      line[5].feedback[ch] = (f - (e + gg + h)) * regen;
      line[6].feedback[ch] = (gg - (e + f + h)) * regen;
      line[7].feedback[ch] = (h - (e + f + gg)) * regen;
    }
    input_l = (g[0][0] + g[1][0] + g[2][0] + g[3][0] + g[4][0] + g[5][0] + g[6][0] + g[7][0]) / 8.0;  // :321-338
    input_r = (g[0][1] + g[1][1] + g[2][1] + g[3][1] + g[4][1] + g[5][1] + g[6][1] + g[7][1]) / 8.0;
    input_l = biquad_b_l.process_sample(biquad_b_coefficients, input_l);
    input_r = biquad_b_r.process_sample(biquad_b_coefficients, input_r);
    input_l = rclamp(input_l, -1.0, 1.0); input_r = rclamp(input_r, -1.0, 1.0);
    input_l = std::asin(input_l); input_r = std::asin(input_r);
    input_l = biquad_c_l.process_sample(biquad_c_coefficients, input_l);
    input_r = biquad_c_r.process_sample(biquad_c_coefficients, input_r);
    if (wetd != 1.0) { input_l += dry_l * (1.0 - wetd); input_r += dry_r * (1.0 - wetd); }
    frame[0] = (float)input_l; frame[1] = (float)input_r;
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :391-407
    sample_rate = sr; channel_count = ch;
    if (ch != 2) return false;
    room_size.set_sample_rate(sr); wet.set_sample_rate(sr);
    return true;
  }
  void block_params(double rs, double w, float& cutoff, double& size, double& blend, double& regen) const {  // :413-420
    cutoff = (float)(10000.0 - (rs * w * 3000.0));
    size = (rs * rs * 75.0) + 25.0;
    double t = 1.0 - (0.82 - (((1.0 - rs) * 0.7) + (size * 0.002)));
    double depth_factor = 1.0 - (t * t) * (t * t);  // powi(4): x*x squared (LLVM powi expansion)
    blend = 0.955 - (size * 0.007);
    regen = depth_factor * 0.5;
  }
  void process(float* out, size_t n) override {  // :409-447
    if (room_size.value_need_ramp() || wet.value_need_ramp()) {
      for (size_t f = 0; f + 2 <= n; f += 2) {
        double rs = (double)room_size.next_value();
        double w = (double)wet.next_value();
        float cutoff; double size, blend, regen;
        block_params(rs, w, cutoff, size, blend, regen);
        size_t predelay = update_delay_sizes(size);
        update_filter_coefs(cutoff);
        process_frame(out + f, blend, regen, predelay, w);
      }
    } else {
      double rs = (double)room_size.target_value();
      double w = (double)wet.target_value();
      float cutoff; double size, blend, regen;
      block_params(rs, w, cutoff, size, blend, regen);
      size_t predelay = update_delay_sizes(size);
      update_filter_coefs(cutoff);
      for (size_t f = 0; f + 2 <= n; f += 2) process_frame(out + f, blend, regen, predelay, w);
    }
  }
  bool process_tail(size_t& frames) const override {  // :449-467
    double rs = (double)room_size.target_value();
    double size = (rs * rs * 75.0) + 25.0;
    size_t max_delay = as_usize(79.0 * size);
    double t = 1.0 - (0.82 - (((1.0 - rs) * 0.7) + (size * 0.002)));
    double fb = 1.0 - (t * t) * (t * t);
    if (fb >= 1.0) frames = USIZE_MAX;
    else if (fb == 0.0) frames = max_delay;
    else frames = max_delay + as_usize((double)max_delay * std::log10(0.001) / std::log10(fb));
    return true;
  }
  bool process_reset_message() override {  // :469-487
    for (auto& l : line) l.flush();
    for (auto& a : ap) a.flush();
    m.flush();
    return true;
  }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {
    if (id == room_size.description.id) room_size.apply_update(u);
    else if (id == wet.description.id) wet.apply_update(u);
    else return false;
    return true;
  }
};

// ---- src/effect/chorus.rs --------------------------------------------------------------------
struct ChorusEffect : Effect {
  static constexpr float MAX_APPLIED_RANGE_IN_SAMPLES = 256.0f, MAX_APPLIED_DELAY_IN_MS = 100.0f;
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  SmoothedParameterValue<LinearSmoothedValue> rate =
      SmoothedParameterValue<LinearSmoothedValue>(FloatParameter{fourcc("rate"), 0.01f, 10.0f, 1.0f, {Scaling::Exponential, 2.0f, 0}})
          .with_smoother(LinearSmoothedValue().with_step(0.005f));
  SmoothedParameterValue<LinearSmoothedValue> phase =
      SmoothedParameterValue<LinearSmoothedValue>(FloatParameter{fourcc("phas"), 0.0f, (float)F64_PI, (float)F64_PI / 2.0f, {}})
          .with_smoother(LinearSmoothedValue().with_step(0.001f));
  SmoothedParameterValue<> depth{FloatParameter{fourcc("dpth"), 0.0f, 1.0f, 0.25f, {}}};
  SmoothedParameterValue<> feedback{FloatParameter{fourcc("fdbk"), -1.0f, 1.0f, 0.5f, {}}};
  SmoothedParameterValue<SpringSmoothedValue> delay =
      SmoothedParameterValue<SpringSmoothedValue>(FloatParameter{fourcc("dlay"), 0.0f, 100.0f, 12.0f, {}})
          .with_smoother(SpringSmoothedValue().with_duration(1000));
  SmoothedParameterValue<> wet_mix{FloatParameter{fourcc("wet_"), 0.0f, 1.0f, 0.5f, {}}};
  EnumParameterValue filter_type{EnumParameter{fourcc("fltt"), 3, 0}};  // SvfFilterType: Lowpass, Highpass, Bandpass
  SmoothedParameterValue<> filter_freq{FloatParameter{fourcc("fltf"), 20.0f, 20000.0f, 20000.0f, {Scaling::Exponential, 2.5f, 0}}};
  SmoothedParameterValue<> filter_resonance{FloatParameter{fourcc("fltq"), 0.0f, 1.0f, 0.0f, {}}};
  float lfo_range = 0.0f;
  double current_phase = 0.0;
  Lfo left_osc, right_osc;
  InterpolatedDelayLine<1> delay_buffer_left, delay_buffer_right;
  SvfCoefficients filter_coefficients;
  SvfFilter filter_left, filter_right;

  const char* name() const override { return "Chorus"; }
  size_t weight() const override { return 3; }
  static SvfType to_svf(int v) { return v == 0 ? SvfType::Lowpass : (v == 1 ? SvfType::Highpass : SvfType::Bandpass); }
  bool init_param(uint32_t id, float raw) override {
    if (id == rate.description.id) { rate.init_value(raw); return true; }
    if (id == phase.description.id) { phase.init_value(raw); return true; }
    if (id == depth.description.id) { depth.init_value(raw); return true; }
    if (id == feedback.description.id) { feedback.init_value(raw); return true; }
    if (id == delay.description.id) { delay.init_value(raw); return true; }
    if (id == wet_mix.description.id) { wet_mix.init_value(raw); return true; }
    if (id == filter_type.description.id) { filter_type.set_value((int)raw); return true; }
    if (id == filter_freq.description.id) { filter_freq.init_value(raw); return true; }
    if (id == filter_resonance.description.id) { filter_resonance.init_value(raw); return true; }
    return false;
  }
  void reset_lfos() {  // :212-221
    double r = (double)rate.current_value();
    left_osc = Lfo(sample_rate, r, LfoWaveform::Sine);
    right_osc = Lfo(sample_rate, r, LfoWaveform::Sine);
    double phase_offset = (double)phase.current_value();
    left_osc.set_phase_degrees((float)current_phase);
    right_osc.set_phase_degrees((float)(current_phase + phase_offset));
  }
  void reset() {  // :201-210
    delay_buffer_left.flush(); delay_buffer_right.flush();
    filter_left.reset(); filter_right.reset();
    rate.init_value(rate.target_value());
    phase.init_value(phase.target_value());
    current_phase = 0.0;
    reset_lfos();
  }
  void update_lfos() {  // :223-231
    double r = (double)rate.next_value();
    left_osc.set_rate(sample_rate, r);
    right_osc.set_rate(sample_rate, r);
    double phase_offset = (double)phase.next_value();
    left_osc.set_phase_degrees((float)current_phase);
    right_osc.set_phase_degrees((float)(current_phase + phase_offset));
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :263-309
    sample_rate = sr; channel_count = ch;
    if (ch != 2) return false;
    rate.set_sample_rate(sr); phase.set_sample_rate(sr); depth.set_sample_rate(sr); feedback.set_sample_rate(sr);
    delay.set_sample_rate(sr); wet_mix.set_sample_rate(sr); filter_freq.set_sample_rate(sr); filter_resonance.set_sample_rate(sr);
    lfo_range = MAX_APPLIED_RANGE_IN_SAMPLES * ((float)sample_rate / 44100.0f);
    size_t max_depth_in_samples = as_usize(std::ceil(lfo_range));
    size_t max_delay_time_in_samples = as_usize(std::ceil(MAX_APPLIED_DELAY_IN_MS * (float)sample_rate / 1000.0f));
    size_t max_buffer_size = 2 + max_delay_time_in_samples + 2 * max_depth_in_samples + 1;
    delay_buffer_left = InterpolatedDelayLine<1>(max_buffer_size);
    delay_buffer_right = InterpolatedDelayLine<1>(max_buffer_size);
    float c = rclampf(filter_freq.target_value(), 20.0f, (float)sr / 2.0f);
    filter_coefficients = SvfCoefficients();
    if (!filter_coefficients.set(to_svf(filter_type.value()), sr, c, filter_resonance.target_value())) return false;
    reset();
    return true;
  }
  void process(float* out, size_t n) override {  // :311-394
    for (size_t f = 0; f + 2 <= n; f += 2) {
      float left_input = out[f], right_input = out[f + 1];
      float delay_ms = delay.next_value();
      float dpt = depth.next_value();
      float fb = rclampf(feedback.next_value(), -0.999f, 0.999f);
      float wet = wet_mix.next_value();
      float wet_amount = wet;
      float dry_amount = 1.0f - wet;
      if (rate.value_need_ramp() || phase.value_need_ramp()) update_lfos();
      double filtered_left, filtered_right;
      if (filter_freq.value_need_ramp() || filter_resonance.value_need_ramp()) {
        float c = rclampf(filter_freq.next_value(), 20.0f, (float)sample_rate / 2.0f);
        float res = filter_resonance.next_value();
        filter_coefficients.set(to_svf(filter_type.value()), sample_rate, c, res);
        filtered_left = filter_left.process_sample(filter_coefficients, (double)left_input);
        filtered_right = filter_right.process_sample(filter_coefficients, (double)right_input);
      } else {
        filtered_left = filter_left.process_sample(filter_coefficients, (double)left_input);
        filtered_right = filter_right.process_sample(filter_coefficients, (double)right_input);
      }
      float delay_in_samples = delay_ms * (float)sample_rate * 0.001f;
      float depth_in_samples = lfo_range * dpt;
      float left_lfo = left_osc.run();
      float right_lfo = right_osc.run();
      float left_delay_pos = 2.0f + delay_in_samples + (1.0f + left_lfo) * depth_in_samples;
      float right_delay_pos = 2.0f + delay_in_samples + (1.0f + right_lfo) * depth_in_samples;
      float fl = (float)filtered_left, fr = (float)filtered_right;
      float left_output, right_output;
      delay_buffer_left.process(&fl, fb, left_delay_pos, &left_output);
      delay_buffer_right.process(&fr, fb, right_delay_pos, &right_output);
      out[f] = left_input * dry_amount + left_output * wet_amount;
      out[f + 1] = right_input * dry_amount + right_output * wet_amount;
    }
    double phase_inc = 2.0 * F64_PI * (double)rate.current_value() / (double)sample_rate;  // :388-393
    current_phase += (double)n / (double)channel_count * phase_inc;
    while (current_phase >= 2.0 * F64_PI) current_phase -= 2.0 * F64_PI;
  }
  bool process_tail(size_t& frames) const override {  // :396-416
    float delay_ms = delay.target_value();
    float depth_ms = MAX_APPLIED_RANGE_IN_SAMPLES * 1000.0f / (float)sample_rate;
    float total_delay_ms = delay_ms + depth_ms;
    float fb = std::fabs(feedback.target_value());
    if (fb >= 1.0f) frames = USIZE_MAX;
    else if (fb < 0.001f) frames = as_usize(std::ceil(total_delay_ms * (float)sample_rate / 1000.0f));
    else {
      float total_delay_samples = total_delay_ms * (float)sample_rate / 1000.0f;
      float decay = total_delay_samples + (float)((double)total_delay_samples * std::log10(0.001) / std::log10((double)fb));
      frames = as_usize(std::ceil(decay));
    }
    return true;
  }
  bool process_reset_message() override { reset(); return true; }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {  // :433-459
    if (id == rate.description.id) rate.apply_update(u);
    else if (id == phase.description.id) phase.apply_update(u);
    else if (id == depth.description.id) depth.apply_update(u);
    else if (id == feedback.description.id) feedback.apply_update(u);
    else if (id == delay.description.id) delay.apply_update(u);
    else if (id == wet_mix.description.id) wet_mix.apply_update(u);
    else if (id == filter_type.description.id) { filter_type.apply_update(u); filter_coefficients.set_filter_type(to_svf(filter_type.value())); }
    else if (id == filter_freq.description.id) filter_freq.apply_update(u);
    else if (id == filter_resonance.description.id) filter_resonance.apply_update(u);
    else return false;
    return true;
  }
};

// ---- src/effect/compressor.rs ----------------------------------------------------------------
struct CompressorEffect : Effect {
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  FloatParameterValue threshold{FloatParameter{fourcc("thrs"), -60.0f, 0.0f, -12.0f, {}}};
  FloatParameterValue ratio{FloatParameter{fourcc("rato"), 1.0f, 20.0f, 8.0f, {}}};
  FloatParameterValue knee_width{FloatParameter{fourcc("knee"), 0.0f, 12.0f, 3.0f, {}}};
  FloatParameterValue attack_time{FloatParameter{fourcc("attk"), 0.001f, 0.5f, 0.02f, {}}};
  FloatParameterValue release_time{FloatParameter{fourcc("rels"), 0.1f, 2.0f, 2.0f, {}}};
  SmoothedParameterValue<> makeup_gain{FloatParameter{fourcc("gain"), -24.0f, 24.0f, 6.0f, {}}};
  FloatParameterValue lookahead_time{FloatParameter{fourcc("look"), 0.001f, 0.2f, 0.04f, {}}};
  EnvelopeFollower envelope_follower;
  std::vector<float> input_buffer;
  LookupDelayLine<2> delay_line;
  const char* name() const override { return "Compressor"; }
  size_t weight() const override { return 4; }
  bool init_param(uint32_t id, float raw) override {
    if (id == threshold.description.id) { threshold.set_value(raw); return true; }
    if (id == ratio.description.id) { ratio.set_value(raw); return true; }
    if (id == knee_width.description.id) { knee_width.set_value(raw); return true; }
    if (id == attack_time.description.id) { attack_time.set_value(raw); return true; }
    if (id == release_time.description.id) { release_time.set_value(raw); return true; }
    if (id == makeup_gain.description.id) { makeup_gain.init_value(raw); return true; }
    if (id == lookahead_time.description.id) { lookahead_time.set_value(raw); return true; }
    return false;
  }
  bool initialize(uint32_t sr, size_t ch, size_t max_frames) override {  // :196-228
    sample_rate = sr; channel_count = ch;
    if (ch != 2) return false;
    makeup_gain.set_sample_rate(sr);
    input_buffer.assign(max_frames * ch, 0.0f);
    delay_line = LookupDelayLine<2>(sr, lookahead_time.value());
    envelope_follower = EnvelopeFollower(sr, attack_time.value(), release_time.value());
    envelope_follower.reset(ratio.value() >= 20.0f ? -120.0f : 0.0f);
    return true;
  }
  void process(float* out, size_t n) override {  // :230-294
    copy_buffers(input_buffer.data(), out, n);
    for (size_t f = 0; f + 2 <= n; f += 2) {
      const float* in_frame = &input_buffer[f];
      float delayed[2];
      delay_line.process(in_frame, delayed);
      float input_db;
      if (ratio.value() >= 20.0f) {
        float lookahead_peak = delay_line.peak_value();
        input_db = (lookahead_peak > 1e-6f) ? 20.0f * std::log10(lookahead_peak) : -120.0f;
      } else {
        float frame_peak = rmaxf(std::fabs(in_frame[0]), std::fabs(in_frame[1]));
        input_db = (frame_peak > 1e-6f) ? 20.0f * std::log10(frame_peak) : -120.0f;
      }
      float envelope = envelope_follower.run(input_db);
      float t = threshold.value();
      float w = knee_width.value();
      float slope = (ratio.value() >= 20.0f) ? 1.0f : 1.0f - 1.0f / ratio.value();
      if (w > 0.0f) log_knee_edge(envelope, t + w / 2.0f, chunk_time_now() + (uint64_t)(f / 2));   // (test hook, po_utils.hpp)
      float gr_db;
      if (w > 0.0f && envelope > (t - w / 2.0f) && envelope < (t + w / 2.0f)) {
        float knee_lower = t - w / 2.0f;
        float x = (envelope - knee_lower) / w;
        gr_db = x * x * slope * w / 2.0f;
      } else if (envelope > (t + w / 2.0f)) {
        gr_db = (envelope - t) * slope;
      } else {
        gr_db = 0.0f;
      }
      float mg = makeup_gain.next_value();
      float total_gain_db = mg - gr_db;
      float total_gain = db_to_linear(total_gain_db);
      out[f] = delayed[0] * total_gain;
      out[f + 1] = delayed[1] * total_gain;
    }
  }
  bool process_tail(size_t& frames) const override {  // :296-302
    size_t la = as_usize(std::ceil(lookahead_time.value() * (float)sample_rate));
    size_t rel = as_usize(std::ceil(release_time.value() * (float)sample_rate));
    frames = la + rel;
    return true;
  }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {  // :304-330
    float old_lookahead = lookahead_time.value();
    if (id == threshold.description.id) threshold.apply_update(u);
    else if (id == ratio.description.id) ratio.apply_update(u);
    else if (id == knee_width.description.id) knee_width.apply_update(u);
    else if (id == attack_time.description.id) attack_time.apply_update(u);
    else if (id == release_time.description.id) release_time.apply_update(u);
    else if (id == makeup_gain.description.id) makeup_gain.apply_update(u);
    else if (id == lookahead_time.description.id) lookahead_time.apply_update(u);
    else return false;
    if (sample_rate > 0) {  // update_envelope_follower :159-166
      envelope_follower.set_attack_time(attack_time.value());
      envelope_follower.set_release_time(release_time.value());
    }
    if (lookahead_time.value() != old_lookahead && sample_rate > 0) delay_line = LookupDelayLine<2>(sample_rate, lookahead_time.value());
    return true;
  }
};

// ---- src/effect/gate.rs ----------------------------------------------------------------------
struct GateEffect : Effect {
  FloatParameterValue threshold{FloatParameter{fourcc("thrs"), -60.0f, 0.0f, -30.0f, {}}};
  FloatParameterValue attack_time{FloatParameter{fourcc("attk"), 0.001f, 0.5f, 0.005f, {}}};
  FloatParameterValue hold_time{FloatParameter{fourcc("hold"), 0.0f, 2.0f, 0.1f, {}}};
  FloatParameterValue release_time{FloatParameter{fourcc("rels"), 0.01f, 2.0f, 0.2f, {}}};
  FloatParameterValue range{FloatParameter{fourcc("rnge"), -60.0f, 0.0f, -60.0f, {}}};
  EnvelopeFollower envelope_follower;
  uint32_t hold_counter = 0;
  float gate_gain_db = -60.0f, attack_coeff = 0.0f, release_coeff = 0.0f;
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  const char* name() const override { return "Gate"; }
  size_t weight() const override { return 2; }
  bool init_param(uint32_t id, float raw) override {
    if (id == threshold.description.id) { threshold.set_value(raw); return true; }
    if (id == attack_time.description.id) { attack_time.set_value(raw); return true; }
    if (id == hold_time.description.id) { hold_time.set_value(raw); return true; }
    if (id == release_time.description.id) { release_time.set_value(raw); return true; }
    if (id == range.description.id) { range.set_value(raw); return true; }
    return false;
  }
  void update_coefficients() {  // :80-90
    if (sample_rate > 0) {
      envelope_follower.set_attack_time(attack_time.value());
      envelope_follower.set_release_time(release_time.value());
      float sr = (float)sample_rate;
      attack_coeff = std::exp(-1.0f / (attack_time.value() * sr));
      release_coeff = std::exp(-1.0f / (release_time.value() * sr));
    }
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :122-145
    if (ch != 2) return false;
    sample_rate = sr; channel_count = ch;
    envelope_follower = EnvelopeFollower(sr, attack_time.value(), release_time.value());
    envelope_follower.reset(-120.0f);
    hold_counter = 0;
    gate_gain_db = range.value();
    update_coefficients();
    return true;
  }
  void process(float* out, size_t n) override {  // :147-195
    float thr = threshold.value();
    float range_db = range.value();
    uint32_t hold_samples = as_u32(hold_time.value() * (float)sample_rate);
    for (size_t f = 0; f + 2 <= n; f += 2) {
      float frame_peak = rmaxf(std::fabs(out[f]), std::fabs(out[f + 1]));
      float input_db = (frame_peak > 1e-6f) ? 20.0f * std::log10(frame_peak) : -120.0f;
      float envelope = envelope_follower.run(input_db);
      float target_gain_db;
      if (envelope >= thr) { hold_counter = hold_samples; target_gain_db = 0.0f; }
      else if (hold_counter > 0) { hold_counter -= 1; target_gain_db = 0.0f; }
      else target_gain_db = range_db;
      if (target_gain_db > gate_gain_db) gate_gain_db = attack_coeff * gate_gain_db + (1.0f - attack_coeff) * target_gain_db;
      else gate_gain_db = release_coeff * gate_gain_db + (1.0f - release_coeff) * target_gain_db;
      float gain = (gate_gain_db <= -60.0f) ? 0.0f : db_to_linear(gate_gain_db);
      out[f] *= gain;
      out[f + 1] *= gain;
    }
  }
  bool process_tail(size_t& frames) const override {  // :197-201
    frames = as_usize(std::ceil(hold_time.value() * (float)sample_rate)) + as_usize(std::ceil(release_time.value() * (float)sample_rate));
    return true;
  }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {
    if (id == threshold.description.id) threshold.apply_update(u);
    else if (id == attack_time.description.id) attack_time.apply_update(u);
    else if (id == hold_time.description.id) hold_time.apply_update(u);
    else if (id == release_time.description.id) release_time.apply_update(u);
    else if (id == range.description.id) range.apply_update(u);
    else return false;
    update_coefficients();
    return true;
  }
};

// ---- src/effect/distortion.rs ----------------------------------------------------------------
namespace dist {
constexpr float MAX_DRIVE = 4.0f;
inline float soft_clip(float sample, float drive) {  // :124-141
  const float BOOST = 15.0f;
  float t = drive / MAX_DRIVE;
  float gain = 1.0f + (t * t) * (BOOST - 1.0f);
  float x = sample * gain;
  if (x >= 1.0f) return 1.0f;
  else if (x > -1.0f) {
    if (gain <= 1.0f) return sample;
    return (3.0f / 2.0f) * (x - (x * x * x) / 3.0f);
  }
  return -1.0f;
}
inline float hard_clip(float sample, float drive) {  // :143-150
  const float BOOST = 25.0f;
  float t = drive / MAX_DRIVE;
  float gain = 1.0f + (t * t) * (BOOST - 1.0f);
  float threshold = 1.0f / gain;
  return rclampf(sample, -threshold, threshold) * gain;
}
inline float diode(float sample, float drive) {  // :152-160
  const float BOOST = 20.0f;
  float t = drive / MAX_DRIVE;
  float curve = 0.6f * (t * t) + 0.4f * t;
  float gain = 1.0f + curve * (BOOST - 1.0f);
  float diode_clipping = std::exp((0.1f * sample) / (0.0253f * 1.68f)) - 1.0f;
  return 2.0f / F32_PI * std::atan(diode_clipping * gain);
}
inline float fuzz(float sample, float drive) {  // :162-175
  const float BOOST = 30.0f;
  float t = drive / MAX_DRIVE;
  float gain = 1.0f + (1.0f - std::exp(-3.0f * t)) * (BOOST - 1.0f);
  float amplified = sample * gain;
  float saturated = (amplified < 0.0f) ? -1.0f * (1.0f - std::exp(-std::fabs(amplified))) : 1.0f * (1.0f - std::exp(-std::fabs(amplified)));
  return 1.5f * (saturated + std::fabs(saturated));
}
inline float fold(float sample, float drive) {  // :177-188
  const float BOOST = 4.0f;
  float t = drive / MAX_DRIVE;
  float gain = 1.0f + (t * t) * (BOOST - 1.0f);
  float x = sample * gain;
  float threshold = 1.0f / gain;
  if (x > threshold || x < -threshold) return std::fabs(std::fmod(std::fabs(x - threshold), threshold * 4.0f) - threshold * 2.0f) - threshold;
  return x;
}
inline float shape(int type, float s, float d) {
  switch (type) { case 0: return soft_clip(s, d); case 1: return hard_clip(s, d); case 2: return diode(s, d); case 3: return fuzz(s, d); default: return fold(s, d); }
}
inline float rms_compensation(int type, float drive) {  // :88-122
  const int N = 256;
  static const float PARTIALS[5][2] = {{1.0f, 0.60f}, {2.7f, 0.25f}, {5.3f, 0.10f}, {9.1f, 0.03f}, {14.6f, 0.02f}};
  float partials_peak = 0.0f;
  for (int p = 0; p < 5; ++p) partials_peak += PARTIALS[p][1];
  float input_sum_sq = 0.0f, output_sum_sq = 0.0f;
  for (int i = 0; i < N; ++i) {
    float t = F32_TAU * ((float)i + 0.5f) / (float)N;
    float s = 0.0f;
    for (int p = 0; p < 5; ++p) s += PARTIALS[p][1] * std::sin(PARTIALS[p][0] * t);
    float sample = s / partials_peak;
    input_sum_sq += sample * sample;
    float o = shape(type, sample, drive);
    output_sum_sq += o * o;
  }
  float input_rms = std::sqrt(input_sum_sq / (float)N);
  float output_rms = std::sqrt(output_sum_sq / (float)N);
  return (output_rms > 1e-10f) ? input_rms / output_rms : 1.0f;
}
}  // namespace dist

struct DistortionEffect : Effect {
  static constexpr int LUT_SIZE = 256;
  EnumParameterValue distortion_type{EnumParameter{fourcc("type"), 5, 2}};  // SoftClip, HardClip, Diode, Fuzz, Fold; default Diode
  SmoothedParameterValue<LinearSmoothedValue> drive =
      SmoothedParameterValue<LinearSmoothedValue>(FloatParameter{fourcc("driv"), 0.0f, dist::MAX_DRIVE, 0.0f, {}})
          .with_smoother(LinearSmoothedValue().with_step(0.01f));
  SmoothedParameterValue<> mix = SmoothedParameterValue<>(FloatParameter{fourcc("mix "), 0.0f, 1.0f, 1.0f, {}})
                                     .with_smoother(ExponentialSmoothedValue().with_inertia(0.1f));
  float compensation_luts[5][LUT_SIZE];
  size_t channel_count = 0;
  DistortionEffect() {  // build_gain_compensation_table :265-277
    for (int s = 0; s < 5; ++s)
      for (int i = 0; i < LUT_SIZE; ++i) compensation_luts[s][i] = dist::rms_compensation(s, (float)i / (float)(LUT_SIZE - 1) * dist::MAX_DRIVE);
  }
  float lookup_gain_compensation(int lut_index, float drv) const {  // :280-288
    const float* lut = compensation_luts[lut_index];
    float pos = rclampf(drv / dist::MAX_DRIVE, 0.0f, 1.0f) * (float)(LUT_SIZE - 1);
    size_t lo = as_usize(pos);
    size_t hi = std::min(lo + 1, (size_t)(LUT_SIZE - 1));
    float frac = pos - (float)lo;
    return lut[lo] + (lut[hi] - lut[lo]) * frac;
  }
  const char* name() const override { return "Distortion"; }
  size_t weight() const override { return 1; }
  bool init_param(uint32_t id, float raw) override {
    if (id == distortion_type.description.id) { distortion_type.set_value((int)raw); return true; }
    if (id == drive.description.id) { drive.init_value(raw); return true; }
    if (id == mix.description.id) { mix.init_value(raw); return true; }
    return false;
  }
  bool initialize(uint32_t sr, size_t ch, size_t) override {  // :314-324
    channel_count = ch;
    mix.set_sample_rate(sr); drive.set_sample_rate(sr);
    return true;
  }
  void process(float* out, size_t n) override {  // :326-361
    int ty = distortion_type.value();
    if (!mix.value_need_ramp() && mix.target_value() == 0.0f) {
    } else if (!mix.value_need_ramp() && mix.target_value() >= 1.0f) {
      if (!drive.value_need_ramp()) {
        float d = drive.target_value();
        float comp = lookup_gain_compensation(ty, d);
        for (size_t i = 0; i < n; ++i) out[i] = dist::shape(ty, out[i], d) * comp;
      } else {
        for (size_t f = 0; f + channel_count <= n; f += channel_count) {
          float d = drive.next_value();
          float comp = lookup_gain_compensation(ty, d);
          for (size_t c = 0; c < channel_count; ++c) out[f + c] = dist::shape(ty, out[f + c], d) * comp;
        }
      }
    } else {
      for (size_t f = 0; f + channel_count <= n; f += channel_count) {
        float d = drive.next_value();
        float comp = lookup_gain_compensation(ty, d);
        float mx = mix.next_value();
        for (size_t c = 0; c < channel_count; ++c) {
          float dry = out[f + c];
          float wetv = dist::shape(ty, dry, d) * comp;
          out[f + c] = (1.0f - mx) * dry + mx * wetv;
        }
      }
    }
  }
  bool process_tail(size_t& frames) const override { frames = 0; return true; }
  bool process_parameter_update(uint32_t id, const ParamUpdate& u) override {
    if (id == distortion_type.description.id) distortion_type.apply_update(u);
    else if (id == drive.description.id) drive.apply_update(u);
    else if (id == mix.description.id) mix.apply_update(u);
    else return false;
    return true;
  }
};

}  // namespace po
