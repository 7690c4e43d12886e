// ORACLE — TEST INFRASTRUCTURE ONLY (see po_utils.hpp header).
//
// CPU restatement of src/parameter/{float,enum,boolean,scaling,smoothed}.rs of emuell/phonic.
#pragma once
#include "po_utils.hpp"

namespace po {

constexpr uint32_t fourcc(const char (&s)[5]) {
  return (uint32_t)(uint8_t)s[0] << 24 | (uint32_t)(uint8_t)s[1] << 16 | (uint32_t)(uint8_t)s[2] << 8 | (uint32_t)(uint8_t)s[3];
}

// src/parameter/scaling.rs:10-108
struct Scaling {
  enum Kind { Linear, Exponential, Decibel } kind = Linear;
  float a = 0.0f, b = 0.0f;
  float scale(float value) const {  // :45-74
    switch (kind) {
      case Linear: return value;
      case Exponential: return std::pow(value, a);
      case Decibel: {
        float db_value = a + value * (b - a);
        float linear_gain = db_to_linear(db_value);
        float min_linear = db_to_linear(a), max_linear = db_to_linear(b);
        return (linear_gain - min_linear) / (max_linear - min_linear);
      }
    }
    return value;
  }
};

// src/parameter.rs:104-110
struct ParamUpdate {
  bool normalized;
  float value;  // Raw: f32 value / enum index / bool; Normalized: 0..1
};

// src/parameter/float.rs:16-141
struct FloatParameter {
  uint32_t id;
  float min, max, def;
  Scaling scaling;
  float clamp_value(float v) const { return rclampf(v, min, max); }
  float denormalize_value(float normalized) const { return min + scaling.scale(normalized) * (max - min); }  // :137-141
};

// src/parameter/float.rs:210-280
struct FloatParameterValue {
  FloatParameter description;
  float value_;
  FloatParameterValue() {}
  explicit FloatParameterValue(const FloatParameter& d) : description(d), value_(d.def) {}
  float value() const { return value_; }
  void set_value(float v) { value_ = v; }
  void apply_update(const ParamUpdate& u) {  // :263-285
    if (!u.normalized) value_ = description.clamp_value(u.value);
    else value_ = description.denormalize_value(rclampf(u.value, 0.0f, 1.0f));
  }
};

// src/parameter/enum.rs (value kept as variant index)
struct EnumParameter {
  uint32_t id;
  int n_values;
  int default_index;
  int denormalize_index(float normalized) const {  // :151-155
    return (int)as_usize(std::round(normalized * (float)(n_values - 1)));
  }
};
struct EnumParameterValue {
  EnumParameter description;
  int value_;
  EnumParameterValue() {}
  explicit EnumParameterValue(const EnumParameter& d) : description(d), value_(d.default_index) {}
  int value() const { return value_; }
  void set_value(int v) { value_ = v; }
  void apply_update(const ParamUpdate& u) {  // :256-290
    if (!u.normalized) {
      int idx = (int)u.value;
      if (idx >= 0 && idx < description.n_values) value_ = idx;  // invalid raw values are ignored with a warning
    } else {
      value_ = description.denormalize_index(rclampf(u.value, 0.0f, 1.0f));
    }
  }
};

// src/parameter/boolean.rs
struct BooleanParameter { uint32_t id; bool def; };
struct BooleanParameterValue {
  BooleanParameter description;
  bool value_;
  BooleanParameterValue() {}
  explicit BooleanParameterValue(const BooleanParameter& d) : description(d), value_(d.def) {}
  bool value() const { return value_; }
  void apply_update(const ParamUpdate& u) {  // :195-214
    if (!u.normalized) value_ = u.value != 0.0f;
    else value_ = rclampf(u.value, 0.0f, 1.0f) >= 0.5f;  // :83-86
  }
};

// src/parameter/smoothed.rs:17-157
template <class S = ExponentialSmoothedValue>
struct SmoothedParameterValue {
  FloatParameter description;
  S value;
  SmoothedParameterValue() {}
  explicit SmoothedParameterValue(const FloatParameter& d) : description(d), value(S::from_f32(d.def)) {}  // from_description :30-36
  SmoothedParameterValue with_smoother(const S& s) const {  // :41-45
    SmoothedParameterValue r = *this;
    r.value = s;
    r.value.init(description.def);
    return r;
  }
  void set_sample_rate(uint32_t sr) { value.set_sample_rate(sr); }
  bool value_need_ramp() const { return value.need_ramp(); }
  float next_value() { return value.next(); }
  float current_value() const { return value.current(); }
  float target_value() const { return value.target(); }
  void init_value(float v) { value.init(v); }
  void apply_update(const ParamUpdate& u) {  // :136-157
    if (!u.normalized) value.set_target(description.clamp_value(u.value));
    else value.set_target(description.denormalize_value(rclampf(u.value, 0.0f, 1.0f)));
  }
};

}  // namespace po
