#include "pg_host_internal.h"

// ---- standalone effect: a graph of UNIT_EFFECT units without mixer logic -----------------------------------------
// One stereo unit for the stereo-only effects. Filter, Eq5, Gain and Distortion take ANY channel count in the reference
// (src/effect/filter.rs:144-201, eq5.rs:297-326, gain.rs:143-166, distortion.rs:326-361): their channels are processed independently —
// own filter states, the SAME per-frame parameter sequence (the smoothers step once per frame, whatever the channel count) — so a
// C-channel instance is ceil(C / 2) stereo units with identical parameter state, one per channel pair (a last odd channel rides in both
// lanes of its unit), all rendered by one launch; the host buffer is de-interleaved into the pairs' buffers and back. Bit-identical to the
// N-channel loop of the reference per channel.
struct pg_effect {
  int kind = 0, device = 0;
  HostFx host;
  bool initialized = false;
  uint32_t sample_rate = 0;
  size_t max_frames = 0;
  size_t channels = 2, n_pairs = 1;
  hipStream_t stream = nullptr;
  PgUnit* d_unit = nullptr;      // [n_pairs]
  PgFx* d_fx = nullptr;          // [n_pairs]
  int32_t* d_fx_index = nullptr; // [n_pairs] identity
  PgCmd* d_cmds = nullptr;
  float* d_buf = nullptr;        // [n_pairs][2 * max_frames]
  std::vector<float> h_pairs;    // host staging of the pair buffers (channel counts other than 2)
  int32_t* d_idx_log = nullptr;  // test hook (pg_effect_debug_index_log)
  size_t idx_log_words = 0;
  std::vector<PgCmd> pending;
  size_t cmd_cap = 64;           // commands d_cmds holds
};
static bool kind_takes_any_channel_count(int kind) { return kind == PG_FX_GAIN || kind == PG_FX_FILTER || kind == PG_FX_EQ5 || kind == PG_FX_DISTORTION; }

extern "C" {

pg_effect* pg_effect_create(int kind, const pg_effect_init* init, int device) {
  std::unique_ptr<pg_effect> e(new pg_effect());
  e->kind = kind; e->device = device;
  if (host_fx_from_init(kind, init, e->host)) return nullptr;
  return e.release();
}
void pg_effect_destroy(pg_effect* e) {
  if (!e) return;
  if (e->initialized) {
    (void)hipSetDevice(e->device);
    (void)pg_stream_sync(e->stream);
    (void)pg_free(e->d_unit); (void)pg_free(e->d_fx); (void)pg_free(e->d_fx_index); (void)pg_free(e->d_cmds); (void)pg_free(e->d_buf);
    if (e->host.d_mem) (void)pg_free(e->host.d_mem);
    if (e->d_idx_log) (void)pg_free(e->d_idx_log);
    (void)hipStreamDestroy(e->stream);
  }
  delete e;
}
int pg_effect_initialize(pg_effect* e, uint32_t sample_rate, size_t channel_count, size_t max_frames) {
  if (e->initialized) return set_error(PG_ERR_STATE, "effect is already initialized");
  if (channel_count != 2 && !kind_takes_any_channel_count(e->kind)) return set_error(PG_ERR_PARAMETER, "%sEffect only supports stereo I/O", KINDS[e->kind].name);
  if (channel_count == 0 || channel_count > 64) return set_error(PG_ERR_PARAMETER, "channel count must be in 1..=64");
  if (sample_rate == 0 || max_frames == 0 || max_frames > PG_MAX_FRAMES) return set_error(PG_ERR_PARAMETER, "max_frames must be in 1..=%d", PG_MAX_FRAMES);
  HIP_TRY(hipSetDevice(e->device));
  PgFx fx;
  int rc = build_fx_device_state(e->host, sample_rate, e->device, true, fx);
  if (rc) return rc;
  HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  const size_t P = (channel_count + 1) / 2;
  std::vector<PgUnit> units(P);
  std::vector<PgFx> fxs(P, fx);  // (the kinds that come in several pairs own no device memory: the state block is the whole instance)
  std::vector<int32_t> idx(P);
  for (size_t p = 0; p < P; ++p) {
    memset(&units[p], 0, sizeof(PgUnit));
    units[p].kind = UNIT_EFFECT; units[p].n_fx = 1; units[p].fx_off = (int)p; units[p].fx0 = (int)p; units[p].effects_bypassed = 0;
    idx[p] = (int32_t)p;
  }
  HIP_TRY(pg_malloc((void**)&e->d_unit, P * sizeof(PgUnit)));
  HIP_TRY(pg_malloc((void**)&e->d_fx, P * sizeof(PgFx)));
  HIP_TRY(pg_malloc((void**)&e->d_fx_index, P * 4));
  HIP_TRY(pg_malloc((void**)&e->d_cmds, sizeof(PgCmd) * 64));
  e->cmd_cap = 64;
  HIP_TRY(pg_malloc((void**)&e->d_buf, P * max_frames * 2 * sizeof(float)));
  HIP_TRY(pg_memcpy(e->d_unit, units.data(), P * sizeof(PgUnit), hipMemcpyHostToDevice));
  HIP_TRY(pg_memcpy(e->d_fx, fxs.data(), P * sizeof(PgFx), hipMemcpyHostToDevice));
  HIP_TRY(pg_memcpy(e->d_fx_index, idx.data(), P * 4, hipMemcpyHostToDevice));
  if (channel_count != 2) e->h_pairs.assign(P * max_frames * 2, 0.0f);
  e->sample_rate = sample_rate; e->max_frames = max_frames; e->channels = channel_count; e->n_pairs = P; e->initialized = true;
  return PG_OK;
}
// Test hook: collect the floor()-derived read indices of the effect's delay lines during the following process calls (time-parallel
// paths of Reverb: ((frame * 8 + line) * 2 + channel) -> read_1 of ReverbDelayLine::get; Delay / Chorus: (frame * 2 + channel) ->
// read_idx1 of InterpolatedDelayLine::process), `words` slots, -1 = not written. out == nullptr: (re)arm the log; else copy it out.
int pg_effect_debug_index_log(pg_effect* e, int32_t* out, size_t words) {
  if (!e->initialized) return set_error(PG_ERR_STATE, "effect is not initialized");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(pg_stream_sync(e->stream));
  if (!out) {
    if (words > e->idx_log_words) {
      if (e->d_idx_log) (void)pg_free(e->d_idx_log);
      e->d_idx_log = nullptr;
      HIP_TRY(pg_malloc((void**)&e->d_idx_log, words * sizeof(int32_t)));
      e->idx_log_words = words;
    }
    if (e->d_idx_log) HIP_TRY(pg_memset(e->d_idx_log, 0xff, e->idx_log_words * sizeof(int32_t)));
    return PG_OK;
  }
  if (!e->d_idx_log || words > e->idx_log_words) return set_error(PG_ERR_PARAMETER, "index log is not armed for %zu words", words);
  HIP_TRY(pg_memcpy(out, e->d_idx_log, words * sizeof(int32_t), hipMemcpyDeviceToHost));
  return PG_OK;
}
int pg_effect_process_started(pg_effect*) { return PG_OK; }  // no-ops for all stock effects (src/effect.rs:127-139)
int pg_effect_process_stopped(pg_effect*) { return PG_OK; }

// room for `n` more queued commands (a launch applies commands at the head of the frames it renders, so a launch of no frames cannot flush them:
// the queue grows instead)
static int effect_reserve_cmd(pg_effect* e, size_t n) {
  if (e->pending.size() + n <= e->cmd_cap) return PG_OK;
  HIP_TRY(hipSetDevice(e->device));
  size_t cap = e->cmd_cap;
  while (e->pending.size() + n > cap) cap *= 2;
  PgCmd* bigger = nullptr;
  HIP_TRY(pg_malloc((void**)&bigger, sizeof(PgCmd) * cap));
  HIP_TRY(pg_stream_sync(e->stream));
  (void)pg_free(e->d_cmds);
  e->d_cmds = bigger;
  e->cmd_cap = cap;
  return PG_OK;
}
// every pair unit gets the command: the pairs share one parameter state by construction (queued unit-major: the kernel walks its commands by unit)
static int effect_queue(pg_effect* e, PgCmd c) {
  int rc = effect_reserve_cmd(e, e->n_pairs);
  if (rc) return rc;
  for (size_t p = 0; p < e->n_pairs; ++p) { c.unit = (int)p; c.target = (int)p; e->pending.push_back(c); }
  return PG_OK;
}
static int effect_run(pg_effect* e, float* host_buf, size_t n_samples, uint64_t pos) {
  HIP_TRY(hipSetDevice(e->device));
  const size_t C = e->channels, P = e->n_pairs, frames = n_samples / C, pair_stride = e->max_frames * 2;
  float* src = host_buf;
  if (C != 2) {  // de-interleave into the pairs' stereo buffers (a last odd channel fills both lanes of its pair)
    for (size_t p = 0; p < P; ++p) {
      const size_t c0 = 2 * p, c1 = 2 * p + 1 < C ? 2 * p + 1 : 2 * p;
      float* d = e->h_pairs.data() + p * pair_stride;
      for (size_t f = 0; f < frames; ++f) { d[2 * f] = host_buf[f * C + c0]; d[2 * f + 1] = host_buf[f * C + c1]; }
    }
    src = e->h_pairs.data();
  }
  if (frames) {
    if (C == 2) HIP_TRY(hipMemcpyAsync(e->d_buf, src, frames * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    else HIP_TRY(hipMemcpy2DAsync(e->d_buf, pair_stride * sizeof(float), src, pair_stride * sizeof(float), frames * 2 * sizeof(float), P, hipMemcpyHostToDevice, e->stream));
  }
  if (!e->pending.empty()) {
    std::stable_sort(e->pending.begin(), e->pending.end(), [](const PgCmd& a, const PgCmd& b) { return a.unit < b.unit; });
    HIP_TRY(hipMemcpyAsync(e->d_cmds, e->pending.data(), e->pending.size() * sizeof(PgCmd), hipMemcpyHostToDevice, e->stream));
  }
  PgLaunch L;
  memset(&L, 0, sizeof L);
  L.units = e->d_unit; L.fx = e->d_fx; L.fx_index = e->d_fx_index;
  L.cmds = e->d_cmds; L.n_cmds = (int)e->pending.size();
  L.n_units = (int)P; L.unit_base = 0; L.n_frames = (uint32_t)frames; L.pos = pos; L.sample_rate = e->sample_rate; L.fast = 1;
  L.bus = e->d_buf; L.bus_unit_stride = pair_stride;
  L.index_log = e->d_idx_log;
  HIP_TRY(pg_launch_units(L, e->stream));
  if (frames) {
    if (C == 2) HIP_TRY(hipMemcpyAsync(host_buf, e->d_buf, frames * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    else HIP_TRY(hipMemcpy2DAsync(e->h_pairs.data(), pair_stride * sizeof(float), e->d_buf, pair_stride * sizeof(float), frames * 2 * sizeof(float), P, hipMemcpyDeviceToHost, e->stream));
  }
  HIP_TRY(pg_stream_sync(e->stream));
  e->pending.clear();
  if (C != 2) {
    for (size_t p = 0; p < P; ++p) {
      const float* d = e->h_pairs.data() + p * pair_stride;
      for (size_t f = 0; f < frames; ++f) { host_buf[f * C + 2 * p] = d[2 * f]; if (2 * p + 1 < C) host_buf[f * C + 2 * p + 1] = d[2 * f + 1]; }
    }
  }
  return PG_OK;
}
int pg_effect_process(pg_effect* e, float* interleaved, size_t n_samples, uint64_t pos_in_frames) {
  if (!e->initialized) return set_error(PG_ERR_STATE, "effect is not initialized");
  if (n_samples % e->channels != 0 || n_samples / e->channels > e->max_frames) return set_error(PG_ERR_PARAMETER, "buffer must hold <= max_frames frames of %zu channel(s)", e->channels);
  if (n_samples == 0) return PG_OK;  // nothing to render: parameter updates and messages received so far stay queued, in order, for the next call that does
  return effect_run(e, interleaved, n_samples, pos_in_frames);
}
int pg_effect_set_parameter(pg_effect* e, uint32_t fourcc, float value, int is_normalized) {
  int pi = find_param(e->kind, fourcc);
  if (pi < 0) return set_error(PG_ERR_PARAMETER, "Unknown parameter: 0x%08x for effect '%s'", fourcc, KINDS[e->kind].name);
  float raw;
  if (!resolve_update(KINDS[e->kind].params[pi], value, is_normalized != 0, raw)) return PG_OK;
  e->host.target[pi] = raw;
  if (!e->initialized) { e->host.init_raw[pi] = raw; return PG_OK; }  // before initialize: plain value update
  PgCmd c;
  memset(&c, 0, sizeof c);
  c.type = CMD_FX_PARAM; c.param = pi; c.value = raw; c.frame = 0; c.value64 = fx_param_aux(e->kind, pi, raw, e->sample_rate);
  return effect_queue(e, c);
}
int pg_effect_message_reset(pg_effect* e) {
  if (e->kind != PG_FX_DELAY && e->kind != PG_FX_REVERB && e->kind != PG_FX_CHORUS)
    return set_error(PG_ERR_PARAMETER, "%sEffect: Invalid/unknown message payload", KINDS[e->kind].name);
  if (!e->initialized) return PG_OK;
  PgCmd c;
  memset(&c, 0, sizeof c);
  c.type = CMD_FX_RESET;
  return effect_queue(e, c);
}
int64_t pg_effect_tail(pg_effect* e) {  // Effect::process_tail from the target values (host shadow)
  const std::vector<float>& t = e->host.target;
  double sr = (double)e->sample_rate;
  switch (e->kind) {
    case PG_FX_GAIN: { int m = (int)t[1]; return m == 0 ? 0 : (int64_t)((uint64_t)e->sample_rate / (uint64_t)(m == 1 ? 1 : (m == 2 ? 5 : 20))); }
    case PG_FX_PANNING: return 0;
    case PG_FX_FILTER: return e->sample_rate / 10;
    case PG_FX_EQ5: return e->sample_rate / 5;
    case PG_FX_DELAY: {
      if (t[P_DELAY_DRIVE] > 0.0f) return -1;
      double delay_ms = (double)(t[P_DELAY_TIME] + 50.0f);
      double fb = (double)std::fabs(t[P_DELAY_FEEDBACK]);
      if (fb >= 0.9999) return INT64_MAX;
      if (fb < 0.001) return (int64_t)d2u64(std::ceil(delay_ms * sr / 1000.0));
      double ds = delay_ms * sr / 1000.0;
      return (int64_t)std::max<uint64_t>(d2u64(std::ceil(ds + ds * std::log10(0.001) / std::log10(fb))), 1);
    }
    case PG_FX_REVERB: {
      double rs = (double)t[0];
      double size = (rs * rs * 75.0) + 25.0;
      uint64_t max_delay = d2u64(79.0 * size);
      double tt = 1.0 - (0.82 - (((1.0 - rs) * 0.7) + (size * 0.002)));
      double fb = 1.0 - (tt * tt) * (tt * tt);
      if (fb >= 1.0) return INT64_MAX;
      if (fb == 0.0) return (int64_t)max_delay;
      return (int64_t)(max_delay + d2u64((double)max_delay * std::log10(0.001) / std::log10(fb)));
    }
    case PG_FX_CHORUS: {
      float srf = (float)e->sample_rate;
      float total_ms = t[P_CHORUS_DELAY] + 256.0f * 1000.0f / srf;
      float fb = std::fabs(t[P_CHORUS_FEEDBACK]);
      if (fb >= 1.0f) return INT64_MAX;
      if (fb < 0.001f) return (int64_t)f2u64(std::ceil(total_ms * srf / 1000.0f));
      float total = total_ms * srf / 1000.0f;
      float decay = total + (float)((double)total * std::log10(0.001) / std::log10((double)fb));
      return (int64_t)f2u64(std::ceil(decay));
    }
    case PG_FX_COMPRESSOR: return (int64_t)(f2u64(std::ceil(t[P_COMP_LOOKAHEAD] * (float)e->sample_rate)) + f2u64(std::ceil(t[P_COMP_RELEASE] * (float)e->sample_rate)));
    case PG_FX_GATE: return (int64_t)(f2u64(std::ceil(t[P_GATE_HOLD] * (float)e->sample_rate)) + f2u64(std::ceil(t[P_GATE_RELEASE] * (float)e->sample_rate)));
    default: return 0;
  }
}

}  // extern "C"
