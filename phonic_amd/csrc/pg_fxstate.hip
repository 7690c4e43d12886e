// Effect construction: the state of each stock effect right after `new()/with_parameters()` + `initialize()`, the shared read-only
// tables (distortion LUTs, vibrato rotation table) and the parameter descriptors of `Effect::parameters()` behind the C ABI.
#include "pg_host_internal.h"

// ---- smoother construction (constructors of src/utils/smoothing.rs + SmoothedParameterValue) --------------
PgSmooth make_smooth(const ParamSpec& p, float value, uint32_t sr) {
  PgSmooth s;
  memset(&s, 0, sizeof s);
  s.comp = 44100.0f / (float)sr;  // set_sample_rate
  s.current = s.target = value;
  switch (p.smooth) {
    case S_LIN:  // LinearSmoothedValue::default().with_step(step); init(v); set_sample_rate(sr)  :257-310,394-401
      s.kind = SM_LIN; s.a = p.smooth_arg; s.b = s.a * s.comp; s.pending = 0; break;
    case S_SPRING:  // SpringSmoothedValue::default().with_duration(d)  :434-474
      s.kind = SM_SPRING; s.a = 5.5f / (float)(size_t)p.smooth_arg; s.b = 0.0f; break;
    default:  // ExponentialSmoothedValue (inertia 1/256 unless stated)  :139,156-170
      s.kind = SM_EXP; s.a = p.smooth_arg > 0.0f ? p.smooth_arg : 1.0f / 256.0f; break;
  }
  return s;
}

// distortion LUT (DistortionType::rms_compensation, src/effect/distortion.rs:88-122): pure function of the
// shaper, built once per device on the host with the same f32 arithmetic as the reference's `new()`.
static float h_dist_shape(int type, float sample, float drive) {
  const float MAX_DRIVE = 4.0f, PI32 = 3.14159274101257324f;
  float t = drive / MAX_DRIVE;
  switch (type) {
    case 0: { float gain = 1.0f + (t * t) * 14.0f; float x = sample * gain; if (x >= 1.0f) return 1.0f; if (x > -1.0f) { if (gain <= 1.0f) return sample; return (3.0f / 2.0f) * (x - (x * x * x) / 3.0f); } return -1.0f; }
    case 1: { float gain = 1.0f + (t * t) * 24.0f; float th = 1.0f / gain; return h_clamp(sample, -th, th) * gain; }
    case 2: { float curve = 0.6f * (t * t) + 0.4f * t; float gain = 1.0f + curve * 19.0f; float dc = std::exp((0.1f * sample) / (0.0253f * 1.68f)) - 1.0f; return 2.0f / PI32 * std::atan(dc * gain); }
    case 3: { float gain = 1.0f + (1.0f - std::exp(-3.0f * t)) * 29.0f; float a = sample * gain; float s = (a < 0.0f) ? -1.0f * (1.0f - std::exp(-std::fabs(a))) : 1.0f * (1.0f - std::exp(-std::fabs(a))); return 1.5f * (s + std::fabs(s)); }
    default: { float gain = 1.0f + (t * t) * 3.0f; float x = sample * gain; float th = 1.0f / gain; if (x > th || x < -th) return std::fabs(std::fmod(std::fabs(x - th), th * 4.0f) - th * 2.0f) - th; return x; }
  }
}
static void build_dist_luts(float* luts /*[5][256]*/) {
  const int N = 256;
  static const float PARTIALS[5][2] = {{1.0f, 0.60f}, {2.7f, 0.25f}, {5.3f, 0.10f}, {9.1f, 0.03f}, {14.6f, 0.02f}};
  float partials_peak = 0.0f;
  for (int p = 0; p < 5; ++p) partials_peak += PARTIALS[p][1];
  for (int type = 0; type < 5; ++type)
    for (int li = 0; li < 256; ++li) {
      float drive = (float)li / 255.0f * 4.0f;
      float in_sq = 0.0f, out_sq = 0.0f;
      for (int i = 0; i < N; ++i) {
        float t = 6.28318548202514648f * ((float)i + 0.5f) / (float)N;
        float s = 0.0f;
        for (int p = 0; p < 5; ++p) s += PARTIALS[p][1] * std::sin(PARTIALS[p][0] * t);
        float sample = s / partials_peak;
        in_sq += sample * sample;
        float o = h_dist_shape(type, sample, drive);
        out_sq += o * o;
      }
      float in_rms = std::sqrt(in_sq / (float)N), out_rms = std::sqrt(out_sq / (float)N);
      luts[type * 256 + li] = (out_rms > 1e-10f) ? in_rms / out_rms : 1.0f;
    }
}
// vibrato rotation table of the reverb fast path: cos/sin(j * depth_i * vib_speed), j = 0..128, for the eight lines
// (depths: src/effect/reverb.rs:137-144; increment depth*speed: reverb.rs:601-603). Read-only, shared by all instances.
static std::mutex g_tables_mutex;  // the shared read-only tables are built once per device, from whichever thread gets there first;
static std::map<int, double*>& g_vib_tabs = *new std::map<int, double*>();  // they live until the process ends (a few KB per device; the map is never destroyed: the tables stay reachable)
static int get_vib_tab(int device, const double** out) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  auto it = g_vib_tabs.find(device);
  if (it == g_vib_tabs.end()) {
    static const double depths[8] = {0.003251, 0.002999, 0.002917, 0.002749, 0.002503, 0.002423, 0.002146, 0.002088};
    std::vector<double> h(8 * 129 * 2);
    for (int i = 0; i < 8; ++i) {
      const double d = depths[i] * 0.1;
      for (int j = 0; j <= 128; ++j) { h[(i * 129 + j) * 2] = std::cos((double)j * d); h[(i * 129 + j) * 2 + 1] = std::sin((double)j * d); }
    }
    double* dp = nullptr;
    HIP_TRY(pg_malloc((void**)&dp, h.size() * 8));
    HIP_TRY(pg_memcpy(dp, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    it = g_vib_tabs.emplace(device, dp).first;
  }
  *out = it->second;
  return PG_OK;
}
static std::map<int, float*>& g_dist_luts = *new std::map<int, float*>();  // per device
static int get_dist_luts(int device, const float** out) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  auto it = g_dist_luts.find(device);
  if (it == g_dist_luts.end()) {
    std::vector<float> h(5 * 256);
    build_dist_luts(h.data());
    float* d = nullptr;
    HIP_TRY(pg_malloc((void**)&d, h.size() * 4));
    HIP_TRY(pg_memcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    it = g_dist_luts.emplace(device, d).first;
  }
  *out = it->second;
  return PG_OK;
}

int host_fx_from_init(int kind, const pg_effect_init* init, HostFx& h) {
  if (kind < 0 || kind >= PG_FX_KIND_COUNT) return set_error(PG_ERR_PARAMETER, "unknown effect kind %d", kind);
  const KindInfo& k = KINDS[kind];
  h.kind = kind;
  h.init_raw.resize(k.n_params);
  for (int i = 0; i < k.n_params; ++i) h.init_raw[i] = k.params[i].def;
  if (init) {
    if (init->n_params > PG_MAX_INIT_PARAMS) return set_error(PG_ERR_PARAMETER, "too many init parameters");
    for (uint32_t i = 0; i < init->n_params; ++i) {
      int pi = find_param(kind, init->fourcc[i]);
      if (pi < 0) return set_error(PG_ERR_PARAMETER, "Unknown parameter: 0x%08x for effect '%s'", init->fourcc[i], k.name);
      const ParamSpec& p = k.params[pi];
      float v = init->value[i];
      if (p.type == PG_PARAM_FLOAT && !(v >= p.min && v <= p.max)) return set_error(PG_ERR_PARAMETER, "Value out of bounds for '%s'", p.name);
      if (p.type == PG_PARAM_ENUM && !((int)v >= 0 && (int)v < p.n_values)) return set_error(PG_ERR_PARAMETER, "Invalid enum index for '%s'", p.name);
      h.init_raw[pi] = v;
      h.with_params = true;
    }
    if (init->has_lfo_seed && (init->lfo_rng_state[0] | init->lfo_rng_state[1] | init->lfo_rng_state[2] | init->lfo_rng_state[3]) != 0) {
      h.has_lfo_seed = true;
      memcpy(h.lfo_rng, init->lfo_rng_state, sizeof h.lfo_rng);
    }
    if (init->has_reverb_seeds) {
      h.has_seeds = true;
      h.fpd_l = init->reverb_fpd_l; h.fpd_r = init->reverb_fpd_r;
      memcpy(h.vib, init->reverb_vib_phase, sizeof h.vib);
    }
  }
  h.target = h.init_raw;
  return PG_OK;
}

// State of the effect right after `Effect::initialize(sample_rate, 2, max_frames)`.
int build_fx_device_state(HostFx& h, uint32_t sr, int device, bool standalone, PgFx& fx) {
  memset(&fx, 0, sizeof fx);
  const KindInfo& k = KINDS[h.kind];
  const ParamSpec* P = k.params;
  const std::vector<float>& v = h.init_raw;
  fx.kind = h.kind;
  fx.sample_rate = sr;
  fx.bypassed = 1;                         // EffectProcessor::new  effect.rs:26-33
  fx.tail_counter = 0;
  fx.silence_counter = PG_USIZE_MAX;
  fx.standalone = standalone ? 1 : 0;
  auto alloc = [&](size_t bytes) -> int {
    h.d_mem_bytes = bytes;
    HIP_TRY(pg_malloc(&h.d_mem, bytes));
    HIP_TRY(pg_memset(h.d_mem, 0, bytes));
    return PG_OK;
  };
  switch (h.kind) {
    case PG_FX_GAIN: {  // gain.rs:123-141
      PgGain& g = fx.u.gain;
      g.gain = make_smooth(P[0], v[0], sr);
      g.dc_mode = (int)v[1];
      double hz = g.dc_mode == 1 ? 1.0 : (g.dc_mode == 3 ? 20.0 : 5.0);  // unwrap_or(Default)
      for (int c = 0; c < 2; ++c) { g.dc[c].x1 = g.dc[c].y1 = 0.0; g.dc[c].r = dc_r(hz, sr); }
    } break;
    case PG_FX_PANNING: {
      PgPan& p = fx.u.pan;
      p.pan = make_smooth(P[0], v[0], sr);
      p.width = make_smooth(P[1], v[1], sr);
      p.invert_l = v[2] != 0.0f; p.invert_r = v[3] != 0.0f;
    } break;
    case PG_FX_FILTER: {  // filter.rs:87-115,141-164
      PgFilter& f = fx.u.filter;
      f.type = (int)v[0];
      f.cutoff = make_smooth(P[1], v[1], sr);
      f.q = make_smooth(P[2], v[2], sr);
      memset(&f.coef, 0, sizeof f.coef);
      biquad_set(f.coef, 0, 44100, 22050.0f, 0.707f, 0.0f);  // new(): coefficients for 44100 Hz (!)
      if (h.with_params) {
        float c = clampf(v[1], 20.0f, 44100.0f / 2.0f);
        if (!biquad_set(f.coef, filter_to_biquad(f.type), 44100, c, v[2], 0.0f)) return set_error(PG_ERR_PARAMETER, "Invalid filter parameters");
      }
      float c = clampf(f.coef.cutoff, 20.0f, (float)sr / 2.0f);  // initialize(): set_cutoff only
      if (f.coef.cutoff != c) { f.coef.cutoff = c; biquad_apply(f.coef); }
    } break;
    case PG_FX_EQ5: {  // eq5.rs:152-170,268-294
      PgEq5& e = fx.u.eq5;
      for (int i = 0; i < 5; ++i) {
        e.gains[i] = make_smooth(P[i * 3], v[i * 3], sr);
        e.freqs[i] = make_smooth(P[i * 3 + 1], v[i * 3 + 1], sr);
        e.bws[i] = make_smooth(P[i * 3 + 2], v[i * 3 + 2], sr);
        memset(&e.coef[i], 0, sizeof e.coef[i]);
        float c = clampf(e.freqs[i].current, 20.0f, (float)sr / 2.0f);
        int bt = i == 0 ? 7 : (i == 4 ? 8 : 6);
        if (!biquad_set(e.coef[i], bt, sr, c, e.bws[i].current, e.gains[i].current)) return set_error(PG_ERR_PARAMETER, "Invalid EQ parameters");
      }
    } break;
    case PG_FX_DELAY: {  // delay.rs:273-332
      PgDelay& d = fx.u.delay;
      d.mode = (int)v[P_DELAY_MODE]; d.filter_type = (int)v[P_DELAY_FTYPE]; d.lfo_shape = (int)v[P_DELAY_LFO_SHAPE];
      d.delay_time = make_smooth(P[P_DELAY_TIME], v[P_DELAY_TIME], sr);
      d.feedback = make_smooth(P[P_DELAY_FEEDBACK], v[P_DELAY_FEEDBACK], sr);
      d.cutoff = make_smooth(P[P_DELAY_CUTOFF], v[P_DELAY_CUTOFF], sr);
      d.drive = make_smooth(P[P_DELAY_DRIVE], v[P_DELAY_DRIVE], sr);
      d.wet = make_smooth(P[P_DELAY_WET], v[P_DELAY_WET], sr);
      d.width = make_smooth(P[P_DELAY_WIDTH], v[P_DELAY_WIDTH], sr);
      d.lfo_rate = make_smooth(P[P_DELAY_LFO_RATE], v[P_DELAY_LFO_RATE], sr);
      d.d_time = make_smooth(P[P_DELAY_D_TIME], v[P_DELAY_D_TIME], sr);
      d.d_feedback = make_smooth(P[P_DELAY_D_FEEDBACK], v[P_DELAY_D_FEEDBACK], sr);
      d.d_filter = make_smooth(P[P_DELAY_D_FILTER], v[P_DELAY_D_FILTER], sr);
      size_t max_delay_samples = (size_t)std::ceil((4000.0f + 50.0f) * (float)sr / 1000.0f);
      size_t frames = next_pow2(max_delay_samples + 4);
      int rc = alloc(frames * 8 * 2);
      if (rc) return rc;
      d.line[0] = (double*)h.d_mem; d.line[1] = d.line[0] + frames;
      d.mask = (uint32_t)(frames - 1);
      memset(&d.coef, 0, sizeof d.coef);
      if (!svf_set(d.coef, d.filter_type, sr, clampf(d.cutoff.target, 20.0f, (float)sr / 2.0f), 0.302f)) return set_error(PG_ERR_PARAMETER, "Invalid delay filter");
      d.lfo.phase = 0.0f; d.lfo.phase_inc = (float)((double)d.lfo_rate.target / (double)sr); d.lfo.waveform = d.lfo_shape;
      // Lfo::new (lfo.rs:70-86): a fresh SmallRng — here the explicit state, else SplitMix64(0x5EED0000) x 4 — and three draws
      if (h.has_lfo_seed) memcpy(d.lfo_rng, h.lfo_rng, sizeof d.lfo_rng);
      else { uint64_t z = 0x5EED0000ull; for (int i = 0; i < 4; ++i) { z += 0x9E3779B97F4A7C15ull; uint64_t x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; d.lfo_rng[i] = x ^ (x >> 31); } }
      d.lfo_sample_hold = lfo_random_bipolar(d.lfo_rng);
      d.lfo_jitter_current = lfo_random_bipolar(d.lfo_rng);
      d.lfo_jitter_target = lfo_random_bipolar(d.lfo_rng);
      for (int c = 0; c < 2; ++c) { d.dc[c].x1 = d.dc[c].y1 = 0.0; d.dc[c].r = dc_r(5.0, sr); }
    } break;
    case PG_FX_REVERB: {  // reverb.rs:94-151,391-407
      PgReverb& r = fx.u.reverb;
      r.room = make_smooth(P[0], v[0], sr);
      r.wet = make_smooth(P[1], v[1], sr);
      r.fpd_l = h.fpd_l; r.fpd_r = h.fpd_r;
      static const size_t sizes[8] = {8111, 7511, 7311, 6911, 6311, 6111, 5511, 4911};
      static const double depths[8] = {0.003251, 0.002999, 0.002917, 0.002749, 0.002503, 0.002423, 0.002146, 0.002088};
      static const size_t apsizes[4] = {4511, 4311, 3911, 3311};
      size_t total = 0;
      for (int i = 0; i < 8; ++i) total += (sizes[i] + 1) * 2;
      for (int i = 0; i < 4; ++i) total += apsizes[i] * 2;
      total += 4096 * 2;  // DelayLine::new(3111) -> next_power_of_two
      int rc = alloc(total * 8);
      if (rc) return rc;
      double* p = (double*)h.d_mem;
      for (int i = 0; i < 8; ++i) {
        PgReverbLine& l = r.line[i];
        l.buf = p; p += (sizes[i] + 1) * 2;
        l.frames = (uint32_t)(sizes[i] + 1); l.count = 1; l.delay = 1;
        l.depth = depths[i];
        l.vib_phase[0] = h.vib[i * 2]; l.vib_phase[1] = h.vib[i * 2 + 1];
      }
      for (int i = 0; i < 4; ++i) { r.ap[i].buf = p; p += apsizes[i] * 2; r.ap[i].frames = (uint32_t)apsizes[i]; r.ap[i].delay = 0; r.ap[i].write_pos = 0; }
      r.pre = p; r.pre_mask = 4095; r.pre_write_pos = 0;
      rc = get_vib_tab(device, &r.vib_tab);
      if (rc) return rc;
    } break;
    case PG_FX_CHORUS: {  // chorus.rs:263-309
      PgChorus& c = fx.u.chorus;
      c.rate = make_smooth(P[P_CHORUS_RATE], v[P_CHORUS_RATE], sr);
      c.depth = make_smooth(P[P_CHORUS_DEPTH], v[P_CHORUS_DEPTH], sr);
      c.feedback = make_smooth(P[P_CHORUS_FEEDBACK], v[P_CHORUS_FEEDBACK], sr);
      c.delay = make_smooth(P[P_CHORUS_DELAY], v[P_CHORUS_DELAY], sr);
      c.wet = make_smooth(P[P_CHORUS_WET], v[P_CHORUS_WET], sr);
      c.phase = make_smooth(P[P_CHORUS_PHASE], v[P_CHORUS_PHASE], sr);
      c.filter_type = (int)v[P_CHORUS_FTYPE];
      c.freq = make_smooth(P[P_CHORUS_FREQ], v[P_CHORUS_FREQ], sr);
      c.res = make_smooth(P[P_CHORUS_RES], v[P_CHORUS_RES], sr);
      c.lfo_range = 256.0f * ((float)sr / 44100.0f);
      size_t max_depth = (size_t)std::ceil(c.lfo_range);
      size_t max_delay = (size_t)std::ceil(100.0f * (float)sr / 1000.0f);
      size_t frames = next_pow2(2 + max_delay + 2 * max_depth + 1);
      int rc = alloc(frames * 8 * 2);
      if (rc) return rc;
      c.line[0] = (double*)h.d_mem; c.line[1] = c.line[0] + frames;
      c.mask = (uint32_t)(frames - 1);
      memset(&c.coef, 0, sizeof c.coef);
      if (!svf_set(c.coef, c.filter_type, sr, clampf(c.freq.target, 20.0f, (float)sr / 2.0f), c.res.target)) return set_error(PG_ERR_PARAMETER, "Invalid chorus filter");
      c.current_phase = 0.0;  // reset() :201-221
      for (int i = 0; i < 2; ++i) { c.osc[i].phase = 0.0f; c.osc[i].waveform = 0; lfo_set_rate(c.osc[i], sr, (double)c.rate.current); }
      lfo_set_phase_degrees(c.osc[0], (float)c.current_phase);
      lfo_set_phase_degrees(c.osc[1], (float)(c.current_phase + (double)c.phase.current));
    } break;
    case PG_FX_COMPRESSOR: {  // compressor.rs:196-228
      PgComp& c = fx.u.comp;
      c.threshold = v[0]; c.ratio = v[1]; c.knee = v[2]; c.attack = v[3]; c.release = v[4];
      c.makeup = make_smooth(P[5], v[5], sr);
      c.lookahead = v[6];
      c.env_attack = env_coeff(c.attack, sr); c.env_release = env_coeff(c.release, sr);
      c.env_current = c.ratio >= 20.0f ? -120.0f : 0.0f;
      size_t maxf = next_pow2((size_t)std::ceil(0.2f * (float)sr) + 1);
      int rc = alloc(maxf * 2 * 8);
      if (rc) return rc;
      c.line = (double*)h.d_mem; c.line_frames = (uint32_t)maxf;
      c.delay_frames = (uint32_t)f2u64(std::ceil(c.lookahead * (float)sr));
      c.mask = c.delay_frames > 0 ? (uint32_t)(next_pow2(c.delay_frames) - 1) : 0;
      c.write_pos = 0; c.peak_pos = 0; c.peak_value = 0.0;
    } break;
    case PG_FX_GATE: {  // gate.rs:122-145
      PgGate& g = fx.u.gate;
      g.threshold = v[0]; g.attack = v[1]; g.hold = v[2]; g.release = v[3]; g.range = v[4];
      g.env_attack = env_coeff(g.attack, sr); g.env_release = env_coeff(g.release, sr);
      g.env_current = -120.0f; g.hold_counter = 0; g.gate_gain_db = g.range;
      g.attack_coeff = std::exp(-1.0f / (g.attack * (float)sr));
      g.release_coeff = std::exp(-1.0f / (g.release * (float)sr));
    } break;
    default: {  // distortion.rs:232-256,314-324
      PgDist& d = fx.u.dist;
      d.type = (int)v[0];
      d.drive = make_smooth(P[1], v[1], sr);
      d.mix = make_smooth(P[2], v[2], sr);
      int rc = get_dist_luts(device, &d.luts);
      if (rc) return rc;
    } break;
  }
  return PG_OK;
}

// insert_event (src/utils/event.rs:31-38): sorted by sample time, after the events of the same time
// PgCmd::value64 of a parameter update: the coefficients that follow from a new attack / release time of the Compressor's and the Gate's envelope
// follower (EnvelopeFollower::set_attack_time / set_release_time, envelope.rs:27-42) and of the Gate's gain smoothing (gate.rs:80-90), computed here
// with the host's expf — the same call the effect's initial state was built with, and the one the reference makes.
uint64_t fx_param_aux(int kind, int param, float raw, uint32_t sr) {
  float lo = 0.0f, hi = 0.0f;
  if (kind == PG_FX_COMPRESSOR && (param == P_COMP_ATTACK || param == P_COMP_RELEASE)) lo = env_coeff(raw, sr);
  else if (kind == PG_FX_GATE && (param == P_GATE_ATTACK || param == P_GATE_RELEASE)) { lo = env_coeff(raw, sr); hi = std::exp(-1.0f / (raw * (float)sr)); }
  uint32_t l, h;
  memcpy(&l, &lo, 4); memcpy(&h, &hi, 4);
  return (uint64_t)l | ((uint64_t)h << 32);
}

extern "C" {

const char* pg_effect_kind_name(int kind) { return (kind >= 0 && kind < PG_FX_KIND_COUNT) ? KINDS[kind].name : nullptr; }
int pg_effect_kind_weight(int kind) { return (kind >= 0 && kind < PG_FX_KIND_COUNT) ? KINDS[kind].weight : -1; }
int pg_effect_kind_param_count(int kind) { return (kind >= 0 && kind < PG_FX_KIND_COUNT) ? KINDS[kind].n_params : -1; }
int pg_effect_kind_param(int kind, int index, pg_param_desc* out) {
  if (kind < 0 || kind >= PG_FX_KIND_COUNT || index < 0 || index >= KINDS[kind].n_params) return set_error(PG_ERR_NOT_FOUND, "no such parameter");
  const ParamSpec& p = KINDS[kind].params[index];
  out->fourcc = p.fourcc; out->type = p.type; out->min = p.min; out->max = p.max; out->default_value = p.def;
  out->scaling = p.scaling; out->scaling_arg0 = p.sa; out->scaling_arg1 = p.sb; out->n_values = p.n_values; out->name = p.name;
  return PG_OK;
}
void pg_voice_options_default(pg_voice_options* o) {  // FilePlaybackOptions::default()  file.rs:94-112
  memset(o, 0, sizeof *o);
  o->volume = 1.0f; o->panning = 0.0f; o->speed = 1.0;
  o->fade_in_seconds = 0.0f; o->fade_out_seconds = 0.05f;
}

}  // extern "C"
