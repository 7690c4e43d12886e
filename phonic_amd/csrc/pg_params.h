// Parameter descriptors of the ten stock effects: ids (FourCC), ranges, defaults, scalings and smoother
// set-ups, exactly as the `pub const` descriptors of the reference (file:line per table), plus the
// host-side mapping of ParameterValueUpdate::{Raw, Normalized} to a raw target value
// (src/parameter/{float,enum,boolean,scaling}.rs).
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/phonic_gpu.h"
#include "pg_dev.h"

namespace pgh {

enum SmoothSpec { S_NONE = 0, S_EXP, S_LIN, S_SPRING };

struct ParamSpec {
  uint32_t fourcc;
  int type;  // pg_param_type
  float min, max, def;
  int scaling;
  float sa, sb;
  int n_values;
  const char* name;
  int smooth;       // SmoothSpec
  float smooth_arg; // exp: inertia (0 = default 1/256), lin: step, spring: duration
};

#define FCC(a, b, c, d) PG_FOURCC(a, b, c, d)

// src/effect/gain.rs:62-81
static const ParamSpec GAIN_PARAMS[] = {
    {FCC('g', 'a', 'i', 'n'), PG_PARAM_FLOAT, 0.000001f, 15.848932f, 1.0f, PG_SCALE_DECIBEL, -60.0f, 24.0f, 0, "Gain", S_EXP, 0},
    {FCC('d', 'c', 'f', 'm'), PG_PARAM_ENUM, 0, 3, 0, 0, 0, 0, 4, "DC Filter", S_NONE, 0},
};
// src/effect/pan.rs:28-50
static const ParamSpec PAN_PARAMS[] = {
    {FCC('p', 'a', 'n', ' '), PG_PARAM_FLOAT, -1.0f, 1.0f, 0.0f, 0, 0, 0, 0, "Pan", S_EXP, 0},
    {FCC('w', 'd', 't', 'h'), PG_PARAM_FLOAT, 0.0f, 2.0f, 1.0f, 0, 0, 0, 0, "Width", S_EXP, 0},
    {FCC('i', 'n', 'v', 'l'), PG_PARAM_BOOL, 0, 1, 0, 0, 0, 0, 0, "Invert L", S_NONE, 0},
    {FCC('i', 'n', 'v', 'r'), PG_PARAM_BOOL, 0, 1, 0, 0, 0, 0, 0, "Invert R", S_NONE, 0},
};
// src/effect/filter.rs:61-81
static const ParamSpec FILTER_PARAMS[] = {
    {FCC('t', 'y', 'p', 'e'), PG_PARAM_ENUM, 0, 3, 0, 0, 0, 0, 4, "Type", S_NONE, 0},
    {FCC('c', 'u', 't', 'o'), PG_PARAM_FLOAT, 20.0f, 20000.0f, 20000.0f, PG_SCALE_EXPONENTIAL, 2.5f, 0, 0, "Cutoff", S_EXP, 0},
    {FCC('f', 'l', 't', 'q'), PG_PARAM_FLOAT, 0.001f, 4.0f, 0.707f, 0, 0, 0, 0, "Resonance", S_LIN, 0.01f},
};
// src/effect/eq5.rs:38-150 (order of parameters(): gain, frequency, bandwidth per band :246-264)
#define EQ_BAND(n, g, f, b, fdef, bmax)                                                                                  \
  {g, PG_PARAM_FLOAT, -20.0f, 20.0f, 0.0f, 0, 0, 0, 0, "Gain " #n, S_EXP, 0},                                            \
      {f, PG_PARAM_FLOAT, 20.0f, 20000.0f, fdef, PG_SCALE_EXPONENTIAL, 2.5f, 0, 0, "Frequency " #n, S_EXP, 0},           \
      {b, PG_PARAM_FLOAT, 0.0001f, bmax, bmax, 0, 0, 0, 0, "Bandwidth " #n, S_LIN, 0.01f}
static const ParamSpec EQ5_PARAMS[] = {
    EQ_BAND(1, FCC('g', 'a', 'n', '1'), FCC('f', 'r', 'q', '1'), FCC('b', 'w', '_', '1'), 100.0f, 1.0f),
    EQ_BAND(2, FCC('g', 'a', 'n', '2'), FCC('f', 'r', 'q', '2'), FCC('b', 'w', '_', '2'), 1000.0f, 4.0f),
    EQ_BAND(3, FCC('g', 'a', 'n', '3'), FCC('f', 'r', 'q', '3'), FCC('b', 'w', '_', '3'), 4000.0f, 4.0f),
    EQ_BAND(4, FCC('g', 'a', 'n', '4'), FCC('f', 'r', 'q', '4'), FCC('b', 'w', '_', '4'), 8000.0f, 4.0f),
    EQ_BAND(5, FCC('g', 'a', 'n', '5'), FCC('f', 'r', 'q', '5'), FCC('b', 'w', '_', '5'), 12000.0f, 1.0f),
};
// src/effect/delay.rs:124-177 (order of parameters() :255-271)
static const ParamSpec DELAY_PARAMS[] = {
    {FCC('m', 'o', 'd', 'e'), PG_PARAM_ENUM, 0, 1, 0, 0, 0, 0, 2, "Mode", S_NONE, 0},
    {FCC('d', 'l', 'a', 'y'), PG_PARAM_FLOAT, 1.0f, 4000.0f, 375.0f, 0, 0, 0, 0, "Delay", S_SPRING, 20000},
    {FCC('f', 'd', 'b', 'k'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.5f, 0, 0, 0, 0, "Feedback", S_EXP, 0},
    {FCC('f', 't', 'y', 'p'), PG_PARAM_ENUM, 0, 2, 0, 0, 0, 0, 3, "Filter Type", S_NONE, 0},
    {FCC('c', 'u', 't', 'o'), PG_PARAM_FLOAT, 20.0f, 20000.0f, 6000.0f, PG_SCALE_EXPONENTIAL, 2.5f, 0, 0, "Filter Cutoff", S_EXP, 0},
    {FCC('d', 'r', 'i', 'v'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.0f, 0, 0, 0, 0, "Drive", S_EXP, 0},
    {FCC('w', 'e', 't', '_'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.5f, 0, 0, 0, 0, "Wet", S_EXP, 0},
    {FCC('w', 'd', 't', 'h'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.5f, 0, 0, 0, 0, "Width", S_EXP, 0},
    {FCC('l', 'f', 'o', 'r'), PG_PARAM_FLOAT, 0.01f, 10.0f, 1.0f, PG_SCALE_EXPONENTIAL, 2.0f, 0, 0, "LFO Rate", S_EXP, 0},
    {FCC('l', 'f', 'o', 's'), PG_PARAM_ENUM, 0, 6, 0, 0, 0, 0, 7, "LFO Shape", S_NONE, 0},
    {FCC('l', 'f', 'd', 't'), PG_PARAM_FLOAT, -1.0f, 1.0f, 0.0f, 0, 0, 0, 0, "LFO -> Time", S_EXP, 0},
    {FCC('l', 'd', 'f', 'b'), PG_PARAM_FLOAT, -1.0f, 1.0f, 0.0f, 0, 0, 0, 0, "LFO -> Feedback", S_EXP, 0},
    {FCC('l', 'f', 'd', 'f'), PG_PARAM_FLOAT, -1.0f, 1.0f, 0.0f, 0, 0, 0, 0, "LFO -> Filter", S_EXP, 0},
};
// src/effect/reverb.rs:78-91
static const ParamSpec REVERB_PARAMS[] = {
    {FCC('r', 'o', 'o', 'm'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.6f, 0, 0, 0, 0, "Room Size", S_LIN, 0.01f},
    {FCC('w', 'e', 't', ' '), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.35f, 0, 0, 0, 0, "Wet", S_EXP, 0},
};
// src/effect/chorus.rs:79-137 (order of parameters() :249-261)
static const ParamSpec CHORUS_PARAMS[] = {
    {FCC('r', 'a', 't', 'e'), PG_PARAM_FLOAT, 0.01f, 10.0f, 1.0f, PG_SCALE_EXPONENTIAL, 2.0f, 0, 0, "Rate", S_LIN, 0.005f},
    {FCC('d', 'p', 't', 'h'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.25f, 0, 0, 0, 0, "Depth", S_EXP, 0},
    {FCC('f', 'd', 'b', 'k'), PG_PARAM_FLOAT, -1.0f, 1.0f, 0.5f, 0, 0, 0, 0, "Feedback", S_EXP, 0},
    {FCC('d', 'l', 'a', 'y'), PG_PARAM_FLOAT, 0.0f, 100.0f, 12.0f, 0, 0, 0, 0, "Delay", S_SPRING, 1000},
    {FCC('w', 'e', 't', '_'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.5f, 0, 0, 0, 0, "Wet", S_EXP, 0},
    {FCC('p', 'h', 'a', 's'), PG_PARAM_FLOAT, 0.0f, 3.14159274101257324f, 3.14159274101257324f / 2.0f, 0, 0, 0, 0, "Phase", S_LIN, 0.001f},
    {FCC('f', 'l', 't', 't'), PG_PARAM_ENUM, 0, 2, 0, 0, 0, 0, 3, "Filter Type", S_NONE, 0},
    {FCC('f', 'l', 't', 'f'), PG_PARAM_FLOAT, 20.0f, 20000.0f, 20000.0f, PG_SCALE_EXPONENTIAL, 2.5f, 0, 0, "Filter Freq", S_EXP, 0},
    {FCC('f', 'l', 't', 'q'), PG_PARAM_FLOAT, 0.0f, 1.0f, 0.0f, 0, 0, 0, 0, "Filter Resonance", S_EXP, 0},
};
// src/effect/compressor.rs:44-91
static const ParamSpec COMP_PARAMS[] = {
    {FCC('t', 'h', 'r', 's'), PG_PARAM_FLOAT, -60.0f, 0.0f, -12.0f, 0, 0, 0, 0, "Threshold", S_NONE, 0},
    {FCC('r', 'a', 't', 'o'), PG_PARAM_FLOAT, 1.0f, 20.0f, 8.0f, 0, 0, 0, 0, "Ratio", S_NONE, 0},
    {FCC('k', 'n', 'e', 'e'), PG_PARAM_FLOAT, 0.0f, 12.0f, 3.0f, 0, 0, 0, 0, "Knee", S_NONE, 0},
    {FCC('a', 't', 't', 'k'), PG_PARAM_FLOAT, 0.001f, 0.5f, 0.02f, 0, 0, 0, 0, "Attack", S_NONE, 0},
    {FCC('r', 'e', 'l', 's'), PG_PARAM_FLOAT, 0.1f, 2.0f, 2.0f, 0, 0, 0, 0, "Release", S_NONE, 0},
    {FCC('g', 'a', 'i', 'n'), PG_PARAM_FLOAT, -24.0f, 24.0f, 6.0f, 0, 0, 0, 0, "Makeup Gain", S_EXP, 0},
    {FCC('l', 'o', 'o', 'k'), PG_PARAM_FLOAT, 0.001f, 0.2f, 0.04f, 0, 0, 0, 0, "Lookahead", S_NONE, 0},
};
// src/effect/gate.rs:33-47
static const ParamSpec GATE_PARAMS[] = {
    {FCC('t', 'h', 'r', 's'), PG_PARAM_FLOAT, -60.0f, 0.0f, -30.0f, 0, 0, 0, 0, "Threshold", S_NONE, 0},
    {FCC('a', 't', 't', 'k'), PG_PARAM_FLOAT, 0.001f, 0.5f, 0.005f, 0, 0, 0, 0, "Attack", S_NONE, 0},
    {FCC('h', 'o', 'l', 'd'), PG_PARAM_FLOAT, 0.0f, 2.0f, 0.1f, 0, 0, 0, 0, "Hold", S_NONE, 0},
    {FCC('r', 'e', 'l', 's'), PG_PARAM_FLOAT, 0.01f, 2.0f, 0.2f, 0, 0, 0, 0, "Release", S_NONE, 0},
    {FCC('r', 'n', 'g', 'e'), PG_PARAM_FLOAT, -60.0f, 0.0f, -60.0f, 0, 0, 0, 0, "Range", S_NONE, 0},
};
// src/effect/distortion.rs:209-228
static const ParamSpec DIST_PARAMS[] = {
    {FCC('t', 'y', 'p', 'e'), PG_PARAM_ENUM, 0, 4, 2, 0, 0, 0, 5, "Type", S_NONE, 0},
    {FCC('d', 'r', 'i', 'v'), PG_PARAM_FLOAT, 0.0f, 4.0f, 0.0f, 0, 0, 0, 0, "Drive", S_LIN, 0.01f},
    {FCC('m', 'i', 'x', ' '), PG_PARAM_FLOAT, 0.0f, 1.0f, 1.0f, 0, 0, 0, 0, "Mix", S_EXP, 0.1f},
};

struct KindInfo { const char* name; int weight; const ParamSpec* params; int n_params; };
// names: EFFECT_NAME consts; weights: `fn weight` of each effect
static const KindInfo KINDS[PG_FX_KIND_COUNT] = {
    {"Gain", 1, GAIN_PARAMS, 2},     {"Panning", 1, PAN_PARAMS, 4},   {"Filter", 2, FILTER_PARAMS, 3}, {"Eq5", 3, EQ5_PARAMS, 15},
    {"Delay", 3, DELAY_PARAMS, 13},  {"Reverb", 5, REVERB_PARAMS, 2}, {"Chorus", 3, CHORUS_PARAMS, 9}, {"Compressor", 4, COMP_PARAMS, 7},
    {"Gate", 2, GATE_PARAMS, 5},     {"Distortion", 1, DIST_PARAMS, 3},
};

inline int find_param(int kind, uint32_t fourcc) {
  const KindInfo& k = KINDS[kind];
  for (int i = 0; i < k.n_params; ++i) if (k.params[i].fourcc == fourcc) return i;
  return -1;
}

// src/utils.rs:41-51 (host copy for the Decibel scaling)
inline float h_db_to_linear(float value) {
  const float F = 2.30258509299404568402f / 20.0f;
  if (value != value) return value;
  if (value == 0.0f) return 1.0f;
  if (value > -200.0f) return std::exp(value * F);
  return 0.0f;
}
inline float h_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ParameterScaling::scale  src/parameter/scaling.rs:45-74
inline float scale_value(const ParamSpec& p, float v) {
  switch (p.scaling) {
    case PG_SCALE_EXPONENTIAL: return std::pow(v, p.sa);
    case PG_SCALE_DECIBEL: {
      float db_value = p.sa + v * (p.sb - p.sa);
      float lin = h_db_to_linear(db_value);
      float mn = h_db_to_linear(p.sa), mx = h_db_to_linear(p.sb);
      return (lin - mn) / (mx - mn);
    }
    default: return v;
  }
}
// The raw value an update resolves to: FloatParameterValue/SmoothedParameterValue::apply_update
// (float.rs:263-285, smoothed.rs:136-157), EnumParameterValue (enum.rs:256-290), BooleanParameterValue
// (boolean.rs:195-214). Returns false when a raw enum index is invalid (the reference logs and ignores it).
inline bool resolve_update(const ParamSpec& p, float value, bool normalized, float& out) {
  if (p.type == PG_PARAM_FLOAT) {
    if (!normalized) out = h_clamp(value, p.min, p.max);
    else out = p.min + scale_value(p, h_clamp(value, 0.0f, 1.0f)) * (p.max - p.min);
    return true;
  } else if (p.type == PG_PARAM_ENUM) {
    if (!normalized) {
      int idx = (int)value;
      if (idx < 0 || idx >= p.n_values) return false;
      out = (float)idx;
    } else {
      float n = h_clamp(value, 0.0f, 1.0f);
      out = (float)(int)std::round(n * (float)(p.n_values - 1));
    }
    return true;
  }
  if (!normalized) out = value != 0.0f ? 1.0f : 0.0f;
  else out = h_clamp(value, 0.0f, 1.0f) >= 0.5f ? 1.0f : 0.0f;
  return true;
}

}  // namespace pgh
