// Exact per-frame evaluation of the ten stock effects by ONE lane of the unit's workgroup (signal in LDS,
// effect state in HBM). This is the path taken while parameters ramp (per-frame coefficient updates:
// genuinely serial), for short feedback lags, and when the time-parallel paths are disabled
// (pg_graph_set_fast_math(g, 0)). The time-parallel steady-state paths live in pg_fx_fast.h.
//
// Every function follows the reference `Effect::process` of its effect statement by statement.
#pragma once
#include "pg_dsp_dev.h"

namespace pgd {

// ---- GainEffect::process  src/effect/gain.rs:143-166 ---------------------------------------------
DEVN void gain_serial(PgFx& fx, float* sig, int n) {
  PgGain& g = fx.u.gain;
  if (g.dc_mode != 0) {
    for (int c = 0; c < 2; ++c) {
      PgDc d = g.dc[c];
      for (int i = c; i < n; i += 2) sig[i] = (float)dc_tick(d, (double)sig[i]);
      g.dc[c] = d;
    }
  }
  if (sm_need_ramp(g.gain)) {
    PgSmooth s = g.gain;
    for (int f = 0; f + 2 <= n; f += 2) {
      float v = sm_next(s);
      sig[f] *= v; sig[f + 1] *= v;
    }
    g.gain = s;
  } else {
    float v = g.gain.target;
    for (int i = 0; i < n; ++i) sig[i] = sig[i] * v;
  }
}

// ---- PanningEffect::process  src/effect/pan.rs:105-158 -------------------------------------------
DEVN void pan_serial(PgFx& fx, float* sig, int n) {
  PgPan& p = fx.u.pan;
  float inv_l = p.invert_l ? -1.0f : 1.0f;
  float inv_r = p.invert_r ? -1.0f : 1.0f;
  bool has_invert = inv_l < 0.0f || inv_r < 0.0f;
  bool pan_ramping = sm_need_ramp(p.pan);
  bool width_ramping = sm_need_ramp(p.width);
  if (!has_invert && !pan_ramping && !width_ramping && fabsf(p.pan.target) < 1e-6f && fabsf(p.width.target - 1.0f) < 1e-6f) return;
  PgSmooth sp = p.pan, sw = p.width;
  for (int f = 0; f + 2 <= n; f += 2) {
    float l = sig[f] * inv_l;
    float r = sig[f + 1] * inv_r;
    float w = width_ramping ? sm_next(sw) : sw.target;
    if (fabsf(w - 1.0f) > 1e-6f) {
      float mid = (l + r) * 0.5f;
      float side = (l - r) * 0.5f;
      l = mid + side * w;
      r = mid - side * w;
    }
    float pv = pan_ramping ? sm_next(sp) : sp.target;
    if (fabsf(pv) > 1e-6f) {
      float pl, pr;
      panning_factors(pv, pl, pr);
      l *= pl; r *= pr;
    }
    sig[f] = l; sig[f + 1] = r;
  }
  p.pan = sp; p.width = sw;
}


// ---- FilterEffect::process  src/effect/filter.rs:166-201 -----------------------------------------
DEVN void filter_serial(PgFx& fx, float* sig, int n) {
  PgFilter& f = fx.u.filter;
  if (fx.call_ramp) {  // (value_need_ramp where the call began: fx_call_begin)
    PgSmooth sc = f.cutoff, sq = f.q;
    PgBiquadCoef c = f.coef;
    double s0a = f.st[0].ic1eq, s0b = f.st[0].ic2eq, s1a = f.st[1].ic1eq, s1b = f.st[1].ic2eq;
    for (int i = 0; i + 2 <= n; i += 2) {
      float cutoff = clampf(sm_next(sc), 20.0f, (float)fx.sample_rate / 2.0f);
      float q = sm_next(sq);
      biquad_set(c, filter_to_biquad(f.type), fx.sample_rate, cutoff, q, 0.0f);
      sig[i] = (float)biquad_tick(c, s0a, s0b, (double)sig[i]);
      sig[i + 1] = (float)biquad_tick(c, s1a, s1b, (double)sig[i + 1]);
    }
    f.cutoff = sc; f.q = sq; f.coef = c;
    f.st[0].ic1eq = s0a; f.st[0].ic2eq = s0b; f.st[1].ic1eq = s1a; f.st[1].ic2eq = s1b;
  } else {
    PgBiquadCoef c = f.coef;
    for (int ch = 0; ch < 2; ++ch) {
      double a = f.st[ch].ic1eq, b = f.st[ch].ic2eq;
      for (int i = ch; i < n; i += 2) sig[i] = (float)biquad_tick(c, a, b, (double)sig[i]);
      f.st[ch].ic1eq = a; f.st[ch].ic2eq = b;
    }
  }
}

// ---- Eq5Effect  src/effect/eq5.rs ------------------------------------------------------------------
DEV int eq5_band_type(int i) { return i == 0 ? 7 : (i == 4 ? 8 : 6); }
DEV void eq5_update_filter_coefficients(PgFx& fx) {  // :172-188
  PgEq5& e = fx.u.eq5;
  for (int i = 0; i < 5; ++i) {
    float cutoff = clampf(e.freqs[i].current, 20.0f, (float)fx.sample_rate / 2.0f);
    if (!biquad_set(e.coef[i], eq5_band_type(i), fx.sample_rate, cutoff, e.bws[i].current, e.gains[i].current)) return;
  }
}
DEVN void eq5_serial(PgFx& fx, float* sig, int n) {  // :297-326
  PgEq5& e = fx.u.eq5;
  const bool need_ramp = fx.call_ramp != 0;  // (any of the fifteen smoothers needed a ramp where the call began: fx_call_begin)
  int frames = n / 2;
  for (int f = 0; f < frames; ++f) {
    if (need_ramp) {  // ramp_filter_coefficients :190-207
      for (int i = 0; i < 5; ++i) {
        float q = (i == 0 || i == 4) ? sm_next(e.bws[i]) : 1.0f / fmaxf(sm_next(e.bws[i]), 0.001f);
        float cutoff = clampf(sm_next(e.freqs[i]), 20.0f, (float)fx.sample_rate / 2.0f);
        float gain = sm_next(e.gains[i]);
        if (!biquad_set(e.coef[i], eq5_band_type(i), fx.sample_rate, cutoff, q, gain)) break;
      }
    }
    for (int ch = 0; ch < 2; ++ch) {
      float sample = sig[f * 2 + ch];
      for (int i = 0; i < 5; ++i) sample = (float)biquad_tick(e.coef[i], e.st[ch][i].ic1eq, e.st[ch][i].ic2eq, (double)sample);
      sig[f * 2 + ch] = sample;
    }
  }
}

// ---- InterpolatedDelayLine<1>::process  src/utils/dsp/delay.rs:107-155 -----------------------------
DEV float interp_delay_process(double* buf, uint32_t mask, uint32_t& write_pos, float input, float feedback, float delay) {
  double read_pos = (double)write_pos - (double)delay;
  double read_pos_floor = floor(read_pos);
  double fraction = read_pos - read_pos_floor;
  long long index1 = (long long)read_pos_floor;
  long long index2 = index1 + 1;
  uint32_t i1 = (uint32_t)((unsigned long long)index1 & (unsigned long long)mask);
  uint32_t i2 = (uint32_t)((unsigned long long)index2 & (unsigned long long)mask);
  double v1 = buf[i1], v2 = buf[i2];
  float out = (float)(v1 + (v2 - v1) * fraction);
  buf[write_pos & mask] = (double)input + (double)out * (double)feedback;
  write_pos = (write_pos + 1) & mask;
  return out;
}

// saturate  src/effect/delay.rs:70-79
DEV double delay_saturate(double input, float drive) {
  if (drive < 0.001f) return input;
  double gain = 1.0 + (double)drive * 4.0;
  double x = input * gain;
  double x2 = x * x;
  double output = x * (27.0 + x2) / (27.0 + 9.0 * x2);
  return output / sqrt(gain);
}
// process_feedback  src/effect/delay.rs:226-237
DEV float delay_process_feedback(const PgSvfCoef& c, PgState2& st, PgDc& dc, float delayed, float drive) {
  double filtered = svf_tick(c, st.ic1eq, st.ic2eq, (double)delayed);
  double saturated = delay_saturate(filtered, drive);
  float clean = (float)dc_tick(dc, saturated);
  return clampf(clean, -4.0f, 4.0f);
}

// ---- DelayEffect::process  src/effect/delay.rs:334-454 ---------------------------------------------
DEVN void delay_serial(PgFx& fx, float* sig, int n) {
  PgDelay& d = fx.u.delay;
  const float MAX_LFO_TIME_MOD_MS = 50.0f, FILTER_RESONANCE = 0.302f;
  float srf = (float)fx.sample_rate;
  for (int f = 0; f + 2 <= n; f += 2) {
    float left_input = sig[f], right_input = sig[f + 1];
    float lfo_val = delay_lfo_run(d);
    if (sm_need_ramp(d.lfo_rate)) {
      float rate = sm_next(d.lfo_rate);
      lfo_set_rate(d.lfo, fx.sample_rate, (double)rate);
    }
    float base_delay_ms = sm_next(d.delay_time);
    float time_mod_ms = lfo_val * sm_next(d.d_time) * MAX_LFO_TIME_MOD_MS;
    float delay_ms = fmaxf(base_delay_ms + time_mod_ms, 1.0f);
    float delay_samples = delay_ms * 0.001f * srf;
    float filter_depth = sm_next(d.d_filter);
    float filter_mod = powf(2.0f, lfo_val * filter_depth * 2.0f);
    float cutoff = clampf(sm_next(d.cutoff) * filter_mod, 20.0f, (float)fx.sample_rate / 2.0f);
    svf_set(d.coef, delay_to_svf(d.filter_type), fx.sample_rate, cutoff, FILTER_RESONANCE);
    float base_feedback = sm_next(d.feedback);
    float feedback_depth = sm_next(d.d_feedback);
    float fb = clampf(base_feedback + lfo_val * feedback_depth * (1.0f - fabsf(base_feedback)), 0.0f, 0.999f);
    float drive = sm_next(d.drive);
    float wet = sm_next(d.wet);
    float width = sm_next(d.width);
    float wet_l, wet_r;
    if (d.mode == 0) {
      float l_in = left_input + d.fb[0] * fb;
      float delayed_l = interp_delay_process(d.line[0], d.mask, d.write_pos[0], l_in, 0.0f, delay_samples);
      float clean_l = delay_process_feedback(d.coef, d.flt[0], d.dc[0], delayed_l, drive);
      d.fb[0] = clean_l;
      float r_in = right_input + d.fb[1] * fb;
      float delayed_r = interp_delay_process(d.line[1], d.mask, d.write_pos[1], r_in, 0.0f, delay_samples);
      float clean_r = delay_process_feedback(d.coef, d.flt[1], d.dc[1], delayed_r, drive);
      d.fb[1] = clean_r;
      wet_l = clean_l; wet_r = clean_r;
    } else {
      float mono_in = (left_input + right_input) * 0.5f;
      float l_in = mono_in + d.fb[1] * fb;
      float delayed_l = interp_delay_process(d.line[0], d.mask, d.write_pos[0], l_in, 0.0f, delay_samples);
      float clean_l = delay_process_feedback(d.coef, d.flt[0], d.dc[0], delayed_l, drive);
      float r_in = d.fb[0] * fb;
      float delayed_r = interp_delay_process(d.line[1], d.mask, d.write_pos[1], r_in, 0.0f, delay_samples);
      float clean_r = delay_process_feedback(d.coef, d.flt[1], d.dc[1], delayed_r, drive);
      d.fb[0] = clean_l; d.fb[1] = clean_r;
      wet_l = clean_l; wet_r = clean_r;
    }
    float dry_gain = fminf((1.0f - wet) * 2.0f, 1.0f);
    float wet_gain = fminf(wet * 2.0f, 1.0f);
    float out_l = left_input * dry_gain + wet_l * wet_gain;
    float out_r = right_input * dry_gain + wet_r * wet_gain;
    float mid = (out_l + out_r) * 0.5f;
    float side = (out_l - out_r) * 0.5f;
    sig[f] = mid + side * width;
    sig[f + 1] = mid - side * width;
  }
}

// ---- ReverbEffect  src/effect/reverb.rs ------------------------------------------------------------
struct ReverbBlock { double blend, regen, wet; uint32_t predelay; };

// per-block / per-frame parameter law  :413-424 / :429-440
DEV void reverb_params(PgFx& fx, double rs, double w, ReverbBlock& rb) {
  PgReverb& r = fx.u.reverb;
  float cutoff = (float)(10000.0 - (rs * w * 3000.0));
  double size = (rs * rs * 75.0) + 25.0;
  double t = 1.0 - (0.82 - (((1.0 - rs) * 0.7) + (size * 0.002)));
  double depth_factor = 1.0 - (t * t) * (t * t);
  rb.blend = 0.955 - (size * 0.007);
  rb.regen = depth_factor * 0.5;
  rb.wet = w;
  // update_delay_sizes :196-213
  const double k[8] = {79.0, 73.0, 71.0, 67.0, 61.0, 59.0, 53.0, 47.0};
  for (int i = 0; i < 8; ++i) { uint32_t dl = (uint32_t)d2u64(k[i] * size); uint32_t mx = r.line[i].frames - 1; r.line[i].delay = dl < mx ? dl : mx; }
  const double ka[4] = {43.0, 41.0, 37.0, 31.0};
  for (int i = 0; i < 4; ++i) { uint32_t dl = (uint32_t)d2u64(ka[i] * size); uint32_t mx = r.ap[i].frames - 1; r.ap[i].delay = dl < mx ? dl : mx; }
  rb.predelay = (uint32_t)d2u64(29.0 * size);
  // update_filter_coefs :161-194
  cutoff = clampf(cutoff, 20.0f, (float)fx.sample_rate / 2.0f);
  if (biquad_set(r.ca, 0, fx.sample_rate, cutoff, 1.618034f, 0.0f))
    if (biquad_set(r.cb, 0, fx.sample_rate, cutoff, 0.618034f, 0.0f)) biquad_set(r.cc, 0, fx.sample_rate, cutoff, 0.5f, 0.0f);
}

// AllpassDelayLine<2>::process, one channel  src/utils/dsp/delay.rs:314-350
// (both channels share write_pos; the caller advances it once per frame)
DEV double allpass_ch(PgAllpass& a, uint32_t write_pos, uint32_t next_pos, int ch, double input) {
  uint32_t read_pos = write_pos + 1;
  if (read_pos > a.delay) read_pos = 0;
  double delayed = a.buf[read_pos * 2 + ch];
  double buf = input - (delayed * 0.5);
  double out = buf * 0.5;
  a.buf[write_pos * 2 + ch] = buf;
  out += a.buf[next_pos * 2 + ch];
  return out;
}

// ReverbEffect::process_frame  :217-369
DEV void reverb_frame(PgFx& fx, float* frame, const ReverbBlock& rb) {
  PgReverb& r = fx.u.reverb;
  const double vib_speed = 0.1, vib_depth = 7.0;
  double in[2] = {(double)frame[0], (double)frame[1]};
  if (fabs(in[0]) < 1.18e-23) in[0] = (double)r.fpd_l * 1.18e-17;
  if (fabs(in[1]) < 1.18e-23) in[1] = (double)r.fpd_r * 1.18e-17;
  double dry[2] = {in[0], in[1]};
  // predelay: DelayLine<2>::process  src/utils/dsp/delay.rs:47-66
  uint32_t wp = r.pre_write_pos & r.pre_mask;
  r.pre[wp * 2] = in[0]; r.pre[wp * 2 + 1] = in[1];
  wp = (wp + 1) & r.pre_mask;
  if (wp > rb.predelay) wp = 0;
  r.pre_write_pos = wp;
  in[0] = r.pre[wp * 2]; in[1] = r.pre[wp * 2 + 1];
  double apo[4][2];
  uint32_t apw[4], apn[4];
  for (int i = 0; i < 4; ++i) { apw[i] = r.ap[i].write_pos; uint32_t nx = apw[i] + 1; if (nx > r.ap[i].delay) nx = 0; apn[i] = nx; }
  double g[8][2];
  for (int ch = 0; ch < 2; ++ch) {
    double x = biquad_tick(r.ca, r.sa[ch].ic1eq, r.sa[ch].ic2eq, in[ch]);
    x *= rb.wet;
    x = sin(x);
    x = allpass_ch(r.ap[0], apw[0], apn[0], ch, x); apo[0][ch] = x;
    x = allpass_ch(r.ap[1], apw[1], apn[1], ch, x); apo[1][ch] = x;
    x = allpass_ch(r.ap[2], apw[2], apn[2], ch, x); apo[2][ch] = x;
    x = allpass_ch(r.ap[3], apw[3], apn[3], ch, x); apo[3][ch] = x;
  }
  for (int i = 0; i < 4; ++i) r.ap[i].write_pos = apn[i];
  // set :275-282 (a<-l, b<-k, c<-j, d<-i, e<-i, f<-j, g<-k, h<-l), step :284-291
  const int src[8] = {3, 2, 1, 0, 0, 1, 2, 3};
  for (int i = 0; i < 8; ++i) {
    PgReverbLine& l = r.line[i];
    l.buf[l.count * 2] = apo[src[i]][0] + l.feedback[0];
    l.buf[l.count * 2 + 1] = apo[src[i]][1] + l.feedback[1];
    l.count += 1;
    if (l.count > l.delay) l.count = 0;
    l.vib_phase[0] += l.depth * vib_speed;
    l.vib_phase[1] += l.depth * vib_speed;
  }
  // get :554-586
  for (int i = 0; i < 8; ++i) {
    PgReverbLine& l = r.line[i];
    for (int ch = 0; ch < 2; ++ch) {
      double offset = (sin(l.vib_phase[ch]) + 1.0) * vib_depth;
      double working = (double)l.count + offset;
      double w_floor = floor(working);
      double w_frac = working - w_floor;
      uint32_t w_int = (uint32_t)d2u64(w_floor);
      uint32_t read_1 = w_int;
      if (read_1 > l.delay) read_1 -= l.delay + 1;
      uint32_t read_2 = w_int + 1;
      if (read_2 > l.delay) read_2 -= l.delay + 1;
      double val1 = l.buf[read_1 * 2 + ch];
      double val2 = l.buf[read_2 * 2 + ch];
      double interpol = val1 * (1.0 - w_frac) + val2 * w_frac;
      interpol = (1.0 - rb.blend) * interpol + (val1 * rb.blend);
      g[i][ch] = interpol;
    }
  }
  for (int ch = 0; ch < 2; ++ch) {  // :303-319
    double a = g[0][ch], b = g[1][ch], c = g[2][ch], d = g[3][ch], e = g[4][ch], f = g[5][ch], gg = g[6][ch], h = g[7][ch];
    r.line[0].feedback[ch] = (a - (b + c + d)) * rb.regen;
    r.line[1].feedback[ch] = (b - (a + c + d)) * rb.regen;
    r.line[2].feedback[ch] = (c - (a + b + d)) * rb.regen;
    r.line[3].feedback[ch] = (d - (a + b + c)) * rb.regen;
    r.line[4].feedback[ch] = (e - (f + gg + h)) * rb.regen;
    r.line[5].feedback[ch] = (f - (e + gg + h)) * rb.regen;
    r.line[6].feedback[ch] = (gg - (e + f + h)) * rb.regen;
    r.line[7].feedback[ch] = (h - (e + f + gg)) * rb.regen;
    double x = (a + b + c + d + e + f + gg + h) / 8.0;
    x = biquad_tick(r.cb, r.sb[ch].ic1eq, r.sb[ch].ic2eq, x);
    x = clampd(x, -1.0, 1.0);
    x = asin(x);
    x = biquad_tick(r.cc, r.sc[ch].ic1eq, r.sc[ch].ic2eq, x);
    if (rb.wet != 1.0) x += dry[ch] * (1.0 - rb.wet);
    frame[ch] = (float)x;
  }
}

// ReverbEffect::process  :409-447
DEVN void reverb_serial(PgFx& fx, float* sig, int n) {
  PgReverb& r = fx.u.reverb;
  r.cache_valid = 0;
  ReverbBlock rb;
  if (sm_need_ramp(r.room) || sm_need_ramp(r.wet)) {
    for (int f = 0; f + 2 <= n; f += 2) {
      double rs = (double)sm_next(r.room);
      double w = (double)sm_next(r.wet);
      reverb_params(fx, rs, w, rb);
      reverb_frame(fx, sig + f, rb);
    }
  } else {
    reverb_params(fx, (double)r.room.target, (double)r.wet.target, rb);
    for (int f = 0; f + 2 <= n; f += 2) reverb_frame(fx, sig + f, rb);
  }
}

// ---- ChorusEffect::process  src/effect/chorus.rs:311-394 -------------------------------------------
DEV void chorus_reset_lfos(PgFx& fx) {  // :212-221
  PgChorus& c = fx.u.chorus;
  double rate = (double)c.rate.current;
  for (int i = 0; i < 2; ++i) { c.osc[i].phase = 0.0f; c.osc[i].waveform = 0; lfo_set_rate(c.osc[i], fx.sample_rate, rate); }
  double phase_offset = (double)c.phase.current;
  lfo_set_phase_degrees(c.osc[0], (float)c.current_phase);
  lfo_set_phase_degrees(c.osc[1], (float)(c.current_phase + phase_offset));
}
DEVN void chorus_serial(PgFx& fx, float* sig, int n) {
  PgChorus& c = fx.u.chorus;
  for (int f = 0; f + 2 <= n; f += 2) {
    float left_input = sig[f], right_input = sig[f + 1];
    float delay_ms = sm_next(c.delay);
    float depth = sm_next(c.depth);
    float feedback = clampf(sm_next(c.feedback), -0.999f, 0.999f);
    float wet_mix = sm_next(c.wet);
    float wet_amount = wet_mix;
    float dry_amount = 1.0f - wet_mix;
    if (sm_need_ramp(c.rate) || sm_need_ramp(c.phase)) {  // update_lfos :223-231
      double rate = (double)sm_next(c.rate);
      lfo_set_rate(c.osc[0], fx.sample_rate, rate);
      lfo_set_rate(c.osc[1], fx.sample_rate, rate);
      double phase_offset = (double)sm_next(c.phase);
      lfo_set_phase_degrees(c.osc[0], (float)c.current_phase);
      lfo_set_phase_degrees(c.osc[1], (float)(c.current_phase + phase_offset));
    }
    if (sm_need_ramp(c.freq) || sm_need_ramp(c.res)) {
      float cutoff = clampf(sm_next(c.freq), 20.0f, (float)fx.sample_rate / 2.0f);
      float res = sm_next(c.res);
      svf_set(c.coef, delay_to_svf(c.filter_type), fx.sample_rate, cutoff, res);
    }
    double filtered_left = svf_tick(c.coef, c.flt[0].ic1eq, c.flt[0].ic2eq, (double)left_input);
    double filtered_right = svf_tick(c.coef, c.flt[1].ic1eq, c.flt[1].ic2eq, (double)right_input);
    float delay_in_samples = delay_ms * (float)fx.sample_rate * 0.001f;
    float depth_in_samples = c.lfo_range * depth;
    float left_lfo = lfo_run(c.osc[0]);
    float right_lfo = lfo_run(c.osc[1]);
    float left_delay_pos = 2.0f + delay_in_samples + (1.0f + left_lfo) * depth_in_samples;
    float right_delay_pos = 2.0f + delay_in_samples + (1.0f + right_lfo) * depth_in_samples;
    float left_output = interp_delay_process(c.line[0], c.mask, c.write_pos[0], (float)filtered_left, feedback, left_delay_pos);
    float right_output = interp_delay_process(c.line[1], c.mask, c.write_pos[1], (float)filtered_right, feedback, right_delay_pos);
    sig[f] = left_input * dry_amount + left_output * wet_amount;
    sig[f + 1] = right_input * dry_amount + right_output * wet_amount;
  }
}
// "Move our LFO offset to keep our oscillators updated when changing the rate or phase" (chorus.rs:388-393): once per process call, behind its
// last frame, with the call's whole length and the rate the call ended on. The callers run it when a call's last piece has been rendered
// (fx_process_wg): update_lfos re-seats the oscillators on current_phase every frame while rate or phase ramp, so the value must not move
// between the pieces of a call.
DEV void chorus_call_end(PgFx& fx, uint64_t call_frames) {
  PgChorus& c = fx.u.chorus;
  double phase_inc = 2.0 * F64_PI * (double)c.rate.current / (double)fx.sample_rate;
  c.current_phase += (double)(call_frames * 2ull) / 2.0 * phase_inc;
  while (c.current_phase >= 2.0 * F64_PI) c.current_phase -= 2.0 * F64_PI;
}

// ---- CompressorEffect::process  src/effect/compressor.rs:230-294 -----------------------------------
// LookupDelayLine<2>::process  src/utils/dsp/delay.rs:206-265
DEV void lookup_process(PgComp& c, const float* in, float* delayed) {
  if (c.delay_frames == 0) { delayed[0] = in[0]; delayed[1] = in[1]; return; }
  uint32_t buffer_frames = c.mask + 1;
  uint32_t read_frame_index = (c.write_pos + buffer_frames - c.delay_frames) & c.mask;
  delayed[0] = (float)c.line[read_frame_index * 2];
  delayed[1] = (float)c.line[read_frame_index * 2 + 1];
  uint32_t write_frame_index = c.write_pos & c.mask;
  c.line[write_frame_index * 2] = (double)in[0];
  c.line[write_frame_index * 2 + 1] = (double)in[1];
  bool peak_expired = c.peak_pos == read_frame_index;
  double new_peak = fmax(fmax(0.0, (double)fabsf(in[0])), (double)fabsf(in[1]));
  if (new_peak >= c.peak_value) {
    c.peak_value = new_peak;
    c.peak_pos = c.write_pos;
  } else if (peak_expired) {
    c.peak_value = 0.0;
    for (uint32_t i = 0; i < c.delay_frames; ++i) {
      uint32_t frame_index = (c.write_pos + buffer_frames - i) & c.mask;
      double frame_peak = fmax(fmax(0.0, fabs(c.line[frame_index * 2])), fabs(c.line[frame_index * 2 + 1]));
      if (frame_peak >= c.peak_value) { c.peak_value = frame_peak; c.peak_pos = frame_index; }
    }
  }
  c.write_pos = (c.write_pos + 1) & c.mask;
}
DEVN void comp_serial(PgFx& fx, float* sig, int n) {
  PgComp& c = fx.u.comp;
  for (int f = 0; f + 2 <= n; f += 2) {
    float in_frame[2] = {sig[f], sig[f + 1]};
    float delayed[2];
    lookup_process(c, in_frame, delayed);
    float input_db;
    if (c.ratio >= 20.0f) {
      float lookahead_peak = (float)c.peak_value;
      input_db = (lookahead_peak > 1e-6f) ? 20.0f * pg_log10f(lookahead_peak) : -120.0f;
    } else {
      float frame_peak = fmaxf(fabsf(in_frame[0]), fabsf(in_frame[1]));
      input_db = (frame_peak > 1e-6f) ? 20.0f * pg_log10f(frame_peak) : -120.0f;
    }
    float envelope = env_run(c.env_current, c.env_attack, c.env_release, input_db);
    float t = c.threshold, w = c.knee;
    float slope = (c.ratio >= 20.0f) ? 1.0f : 1.0f - 1.0f / c.ratio;
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
    float makeup = sm_next(c.makeup);
    float total_gain = db_to_linear(makeup - gr_db);
    sig[f] = delayed[0] * total_gain;
    sig[f + 1] = delayed[1] * total_gain;
  }
}

// ---- GateEffect::process  src/effect/gate.rs:147-195 -----------------------------------------------
DEVN void gate_serial(PgFx& fx, float* sig, int n) {
  PgGate& g = fx.u.gate;
  float threshold = g.threshold, range_db = g.range;
  uint32_t hold_samples = f2u32(g.hold * (float)fx.sample_rate);
  float env = g.env_current, gain_db = g.gate_gain_db;
  uint32_t hold_counter = g.hold_counter;
  for (int f = 0; f + 2 <= n; f += 2) {
    float frame_peak = fmaxf(fabsf(sig[f]), fabsf(sig[f + 1]));
    float input_db = (frame_peak > 1e-6f) ? 20.0f * pg_log10f(frame_peak) : -120.0f;
    float envelope = env_run(env, g.env_attack, g.env_release, input_db);
    float target_gain_db;
    if (envelope >= threshold) { hold_counter = hold_samples; target_gain_db = 0.0f; }
    else if (hold_counter > 0) { hold_counter -= 1; target_gain_db = 0.0f; }
    else target_gain_db = range_db;
    if (target_gain_db > gain_db) gain_db = g.attack_coeff * gain_db + (1.0f - g.attack_coeff) * target_gain_db;
    else gain_db = g.release_coeff * gain_db + (1.0f - g.release_coeff) * target_gain_db;
    float gain = (gain_db <= -60.0f) ? 0.0f : db_to_linear(gain_db);
    sig[f] *= gain;
    sig[f + 1] *= gain;
  }
  g.env_current = env; g.gate_gain_db = gain_db; g.hold_counter = hold_counter;
}

// ---- DistortionEffect  src/effect/distortion.rs ----------------------------------------------------
DEV float dist_shape(int type, float sample, float drive) {  // :124-189
  const float MAX_DRIVE = 4.0f;
  float t = drive / MAX_DRIVE;
  switch (type) {
    case 0: {  // soft_clip
      float gain = 1.0f + (t * t) * (15.0f - 1.0f);
      float x = sample * gain;
      if (x >= 1.0f) return 1.0f;
      if (x > -1.0f) { if (gain <= 1.0f) return sample; return (3.0f / 2.0f) * (x - (x * x * x) / 3.0f); }
      return -1.0f;
    }
    case 1: {  // hard_clip
      float gain = 1.0f + (t * t) * (25.0f - 1.0f);
      float threshold = 1.0f / gain;
      return clampf(sample, -threshold, threshold) * gain;
    }
    case 2: {  // diode
      float curve = 0.6f * (t * t) + 0.4f * t;
      float gain = 1.0f + curve * (20.0f - 1.0f);
      float diode_clipping = expf((0.1f * sample) / (0.0253f * 1.68f)) - 1.0f;
      return 2.0f / F32_PI * atanf(diode_clipping * gain);
    }
    case 3: {  // fuzz
      float gain = 1.0f + (1.0f - expf(-3.0f * t)) * (30.0f - 1.0f);
      float amplified = sample * gain;
      float saturated = (amplified < 0.0f) ? -1.0f * (1.0f - expf(-fabsf(amplified))) : 1.0f * (1.0f - expf(-fabsf(amplified)));
      return 1.5f * (saturated + fabsf(saturated));
    }
    default: {  // fold
      float gain = 1.0f + (t * t) * (4.0f - 1.0f);
      float x = sample * gain;
      float threshold = 1.0f / gain;
      if (x > threshold || x < -threshold) return fabsf(fmodf(fabsf(x - threshold), threshold * 4.0f) - threshold * 2.0f) - threshold;
      return x;
    }
  }
}
DEV float dist_compensation(const float* luts, int lut_index, float drive) {  // :280-288
  const float* lut = luts + lut_index * 256;
  float pos = clampf(drive / 4.0f, 0.0f, 1.0f) * 255.0f;
  uint32_t lo = f2u32(pos);
  uint32_t hi = (lo + 1 < 255u) ? lo + 1 : 255u;
  float frac = pos - (float)lo;
  return lut[lo] + (lut[hi] - lut[lo]) * frac;
}
DEVN void dist_serial(PgFx& fx, float* sig, int n) {  // :326-361 (branches with a ramp; the memoryless branch is parallel)
  PgDist& d = fx.u.dist;
  int ty = d.type;
  if (!sm_need_ramp(d.mix) && d.mix.target == 0.0f) return;
  if (!sm_need_ramp(d.mix) && d.mix.target >= 1.0f) {
    if (!sm_need_ramp(d.drive)) {
      float drive = d.drive.target;
      float comp = dist_compensation(d.luts, ty, drive);
      for (int i = 0; i < n; ++i) sig[i] = dist_shape(ty, sig[i], drive) * comp;
    } else {
      for (int f = 0; f + 2 <= n; f += 2) {
        float drive = sm_next(d.drive);
        float comp = dist_compensation(d.luts, ty, drive);
        sig[f] = dist_shape(ty, sig[f], drive) * comp;
        sig[f + 1] = dist_shape(ty, sig[f + 1], drive) * comp;
      }
    }
  } else {
    for (int f = 0; f + 2 <= n; f += 2) {
      float drive = sm_next(d.drive);
      float comp = dist_compensation(d.luts, ty, drive);
      float mix = sm_next(d.mix);
      for (int c = 0; c < 2; ++c) {
        float dry = sig[f + c];
        float wet = dist_shape(ty, dry, drive) * comp;
        sig[f + c] = (1.0f - mix) * dry + mix * wet;
      }
    }
  }
}

// ---- process_tail of every effect (thread 0) --------------------------------------------------------
// returns false for None; frames == PG_USIZE_MAX for an infinite tail
DEVN bool fx_process_tail(const PgFx& fx, uint64_t& frames) {
  uint32_t sr = fx.sample_rate;
  switch (fx.kind) {
    case 0: {  // gain.rs:168-175
      int m = fx.u.gain.dc_mode;
      frames = (m == 0) ? 0 : (uint64_t)sr / (uint64_t)(m == 1 ? 1 : (m == 2 ? 5 : 20));
      return true;
    }
    case 1: frames = 0; return true;
    case 2: frames = sr / 10; return true;  // filter.rs:203-207
    case 3: frames = sr / 5; return true;   // eq5.rs:328-332
    case 4: {  // delay.rs:456-475
      const PgDelay& d = fx.u.delay;
      if (d.drive.target > 0.0f) return false;
      double delay_ms = (double)(d.delay_time.target + 50.0f);
      double fb = (double)fabsf(d.feedback.target);
      if (fb >= 0.9999) frames = PG_USIZE_MAX;
      else if (fb < 0.001) frames = d2u64(ceil(delay_ms * (double)sr / 1000.0));
      else {
        double delay_samples = delay_ms * (double)sr / 1000.0;
        double decay = delay_samples + delay_samples * log10(0.001) / log10(fb);
        uint64_t v = d2u64(ceil(decay));
        frames = v > 1 ? v : 1;
      }
      return true;
    }
    case 5: {  // reverb.rs:449-467
      double rs = (double)fx.u.reverb.room.target;
      double size = (rs * rs * 75.0) + 25.0;
      uint64_t max_delay = d2u64(79.0 * size);
      double t = 1.0 - (0.82 - (((1.0 - rs) * 0.7) + (size * 0.002)));
      double fb = 1.0 - (t * t) * (t * t);
      if (fb >= 1.0) frames = PG_USIZE_MAX;
      else if (fb == 0.0) frames = max_delay;
      else frames = max_delay + d2u64((double)max_delay * log10(0.001) / log10(fb));
      return true;
    }
    case 6: {  // chorus.rs:396-416
      const PgChorus& c = fx.u.chorus;
      float total_delay_ms = c.delay.target + 256.0f * 1000.0f / (float)sr;
      float fb = fabsf(c.feedback.target);
      if (fb >= 1.0f) frames = PG_USIZE_MAX;
      else if (fb < 0.001f) frames = f2u64(ceilf(total_delay_ms * (float)sr / 1000.0f));
      else {
        float total = total_delay_ms * (float)sr / 1000.0f;
        float decay = total + (float)((double)total * log10(0.001) / log10((double)fb));
        frames = f2u64(ceilf(decay));
      }
      return true;
    }
    case 7: frames = f2u64(ceilf(fx.u.comp.lookahead * (float)sr)) + f2u64(ceilf(fx.u.comp.release * (float)sr)); return true;  // compressor.rs:296-302
    case 8: frames = f2u64(ceilf(fx.u.gate.hold * (float)sr)) + f2u64(ceilf(fx.u.gate.release * (float)sr)); return true;       // gate.rs:197-201
    default: frames = 0; return true;
  }
}

}  // namespace pgd
