// ORACLE — TEST INFRASTRUCTURE ONLY. C API over the CPU restatement so tests/, smoke() and the
// cpu_baseline leg of bench.py can drive it through ctypes with the same call sequence as the
// product's C ABI (include/phonic_gpu.h). Nothing in phonic_amd/ may link or load this library.
//
// Parity pinning: see po_utils.hpp. The reference (Rust) cannot be built here: "parity unpinned"
// except for the reference's own KATs restated in tests/test_oracle_kats.py.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <thread>

#include "../include/phonic_gpu.h"
#include "po_sources.hpp"

using namespace po;

namespace {
std::unique_ptr<Effect> make_effect(int kind, const pg_effect_init* init) {
  std::unique_ptr<Effect> e;
  switch (kind) {
    case PG_FX_GAIN: e.reset(new GainEffect()); break;
    case PG_FX_PANNING: e.reset(new PanningEffect()); break;
    case PG_FX_FILTER: e.reset(new FilterEffect()); break;
    case PG_FX_EQ5: e.reset(new Eq5Effect()); break;
    case PG_FX_DELAY: {
      DelayEffect* d = new DelayEffect();
      if (init && init->has_lfo_seed && (init->lfo_rng_state[0] | init->lfo_rng_state[1] | init->lfo_rng_state[2] | init->lfo_rng_state[3]) != 0)
        d->lfo_seed = SmallRng(init->lfo_rng_state);
      e.reset(d);
    } break;
    case PG_FX_REVERB: {
      double zero[16] = {0};
      if (init && init->has_reverb_seeds) e.reset(new ReverbEffect(init->reverb_fpd_l, init->reverb_fpd_r, init->reverb_vib_phase));
      else e.reset(new ReverbEffect(16386, 16386, zero));
    } break;
    case PG_FX_CHORUS: e.reset(new ChorusEffect()); break;
    case PG_FX_COMPRESSOR: e.reset(new CompressorEffect()); break;
    case PG_FX_GATE: e.reset(new GateEffect()); break;
    case PG_FX_DISTORTION: e.reset(new DistortionEffect()); break;
    default: return nullptr;
  }
  if (init) {
    for (uint32_t i = 0; i < init->n_params && i < PG_MAX_INIT_PARAMS; ++i)
      if (!e->init_param(init->fourcc[i], init->value[i])) return nullptr;
    e->finish_init_params();
  }
  return e;
}
}  // namespace

struct po_effect {
  std::unique_ptr<Effect> fx;
  bool initialized = false;
};

struct po_graph {
  std::unique_ptr<MixedSource> main;
  std::map<int, MixedSource*> mixers;            // id -> mixer (0 = main)
  std::map<int, MixedSource*> effect_mixer;      // effect id -> owning mixer
  std::map<int, MixedSource*> voice_mixer;       // voice id -> owning mixer
  std::map<int, bool> voice_transient;           // voice id -> PlayingSource::is_transient
  std::map<int, int> mixer_parent;               // mixer id -> parent mixer id
  int next_mixer = 1, next_effect = 0, next_voice = 0;
  uint32_t sample_rate;
  size_t channels;
};

extern "C" {

po_effect* po_effect_create(int kind, const pg_effect_init* init) {
  auto fx = make_effect(kind, init);
  if (!fx) return nullptr;
  po_effect* e = new po_effect();
  e->fx = std::move(fx);
  return e;
}
int po_effect_initialize(po_effect* e, uint32_t sr, size_t ch, size_t max_frames) {
  if (!e->fx->initialize(sr, ch, max_frames)) return PG_ERR_PARAMETER;
  e->initialized = true;
  return PG_OK;
}
int po_effect_process(po_effect* e, float* buf, size_t n, uint64_t) { e->fx->process(buf, n); return PG_OK; }
int64_t po_effect_tail(po_effect* e) {
  size_t f;
  if (!e->fx->process_tail(f)) return -1;
  return f == USIZE_MAX ? INT64_MAX : (int64_t)f;
}
int po_effect_set_parameter(po_effect* e, uint32_t id, float value, int normalized) {
  return e->fx->process_parameter_update(id, ParamUpdate{normalized != 0, value}) ? PG_OK : PG_ERR_PARAMETER;
}
int po_effect_message_reset(po_effect* e) { return e->fx->process_reset_message() ? PG_OK : PG_ERR_PARAMETER; }
void po_effect_destroy(po_effect* e) { delete e; }

// reverb state probes for bit-exact index/phase checks in tests
int po_effect_reverb_state(po_effect* e, double* vib_phase16, uint64_t* counts8) {
  auto* r = dynamic_cast<ReverbEffect*>(e->fx.get());
  if (!r) return PG_ERR_PARAMETER;
  for (int i = 0; i < 8; ++i) { vib_phase16[2 * i] = r->line[i].vib_phase[0]; vib_phase16[2 * i + 1] = r->line[i].vib_phase[1]; counts8[i] = r->line[i].count; }
  return PG_OK;
}

// ---- graph -----------------------------------------------------------------------------------
po_graph* po_graph_create(uint32_t sample_rate, uint32_t channels, size_t, int) {
  po_graph* g = new po_graph();
  g->main.reset(new MixedSource(channels, sample_rate));
  g->mixers[0] = g->main.get();
  g->sample_rate = sample_rate;
  g->channels = channels;
  return g;
}
void po_graph_destroy(po_graph* g) { delete g; }

int po_graph_add_mixer_to(po_graph* g, int parent_mixer_id) {  // Player::add_mixer(parent)  src/player.rs:773-822
  auto parent = g->mixers.find(parent_mixer_id);
  if (parent == g->mixers.end()) return -PG_ERR_NOT_FOUND;
  int id = g->next_mixer++;
  std::unique_ptr<SubMixerProcessor> p(new SubMixerProcessor());
  p->mixer.reset(new MixedSource(g->channels, g->sample_rate));
  MixedSource* child = p->mixer.get();
  MixedSource::Message m;
  m.kind = MixedSource::Message::AddMixer;
  m.id = id;
  m.mixer = std::move(p);
  parent->second->message_queue.push_back(std::move(m));
  g->mixers[id] = child;
  g->mixer_parent[id] = parent_mixer_id;
  return id;
}
int po_graph_add_mixer(po_graph* g) { return po_graph_add_mixer_to(g, 0); }  // Player::add_mixer(None): child of the main mixer
int po_graph_add_effect(po_graph* g, int mixer_id, int kind, const pg_effect_init* init) {  // Player::add_effect
  auto it = g->mixers.find(mixer_id);
  if (it == g->mixers.end()) return -PG_ERR_NOT_FOUND;
  auto fx = make_effect(kind, init);
  if (!fx) return -PG_ERR_PARAMETER;
  if (!fx->initialize(g->sample_rate, g->channels, MAX_MIX_BUFFER_SAMPLES / g->channels)) return -PG_ERR_PARAMETER;
  int id = g->next_effect++;
  MixedSource::Message m;
  m.kind = MixedSource::Message::AddEffect;
  m.id = id;
  m.effect.reset(new EffectProcessor(std::move(fx)));
  it->second->message_queue.push_back(std::move(m));
  g->effect_mixer[id] = it->second;
  return id;
}
int po_graph_add_voice(po_graph* g, int mixer_id, const float* pcm, size_t n_frames, uint32_t src_channels, uint32_t src_rate,
                       const pg_voice_options* opt) {  // Player::play_file_source_with_context, src/player.rs:519-602
  auto it = g->mixers.find(mixer_id);
  if (it == g->mixers.end()) return -PG_ERR_NOT_FOUND;
  auto fb = std::make_shared<AudioFileBuffer>();
  fb->buffer.assign(pcm, pcm + n_frames * src_channels);
  fb->sample_rate = src_rate;
  fb->channel_count = src_channels;
  FileOptions fo;
  fo.volume = opt->volume; fo.panning = opt->panning; fo.speed = opt->speed;
  fo.has_repeat = opt->has_repeat != 0;
  fo.repeat = opt->repeat == PG_REPEAT_FOREVER ? USIZE_MAX : (size_t)opt->repeat;
  fo.has_loop_range = opt->has_loop_range != 0;
  fo.loop_start = opt->loop_start; fo.loop_end = opt->loop_end;
  fo.fade_in_seconds = opt->fade_in_seconds;
  fo.fade_out_seconds = opt->fade_out_seconds;
  // PreloadedFileSource::from_shared_buffer(.., sample_rate): Player passes the mixer's rate; pg_voice_options::source_rate asks for another
  auto* file = new PreloadedFileSource(fb, fo, opt->source_rate ? opt->source_rate : g->sample_rate);
  std::unique_ptr<Source> src(file);
  // ConvertedSource::new (converted.rs:15-45): resample to the mixer's rate first if the source runs at another one, then map channels
  if (src->sample_rate() != g->sample_rate) src.reset(new ResampledSource(std::move(src), g->sample_rate));
  if (src->channel_count() != g->channels) src.reset(new ChannelMappedSource(std::move(src), g->channels));
  auto* amp = new AmplifiedSource(std::move(src), fo.volume);
  std::unique_ptr<Source> s2(amp);
  auto* pan = new PannedSource(std::move(s2), fo.panning);
  std::unique_ptr<MixedSource::PlayingSource> ps(new MixedSource::PlayingSource());
  int id = g->next_voice++;
  ps->playback_id = id;
  ps->is_transient = opt->non_transient == 0;   // PlayingSource::is_transient (mixed.rs:34-42): a source the mixer keeps when it is exhausted
  g->voice_transient[id] = ps->is_transient;
  ps->queues.file = file; ps->queues.amplified = amp; ps->queues.panned = pan;
  ps->source.reset(pan);
  ps->start_time = opt->start_time;
  MixedSource::Message m;
  m.kind = MixedSource::Message::AddSource;
  m.source = std::move(ps);
  it->second->message_queue.push_back(std::move(m));
  g->voice_mixer[id] = it->second;
  return id;
}
static int push_event(MixedSource* mx, const MixedSource::MixerEvent& ev) {
  MixedSource::Message m;
  m.kind = MixedSource::Message::Event;
  m.event = ev;
  mx->message_queue.push_back(std::move(m));
  return PG_OK;
}
int po_graph_stop_all_voices(po_graph* g) {  // Player::stop_all_sources (src/player.rs:1012-1045)
  for (auto& kv : g->voice_mixer) {  // send_stop() to every transient source: it stops when it is asked for output next
    if (!g->voice_transient[kv.first]) continue;
    MixedSource::Message m;
    m.kind = MixedSource::Message::StopSource;
    m.id = kv.first; m.sample_time = 0;
    kv.second->message_queue.push_back(std::move(m));
  }
  for (auto& kv : g->mixers) {  // scheduled sources and events of every mixer
    MixedSource::Message m;
    m.kind = MixedSource::Message::RemoveAllPendingEvents;
    kv.second->message_queue.push_back(std::move(m));
  }
  return PG_OK;
}
int po_graph_remove_mixer(po_graph* g, int mixer_id) {  // Player::remove_mixer (src/player.rs:825-867)
  if (mixer_id == 0) return PG_ERR_PARAMETER;
  auto it = g->mixers.find(mixer_id);
  if (it == g->mixers.end()) return PG_ERR_NOT_FOUND;
  MixedSource::Message m;
  m.kind = MixedSource::Message::RemoveMixer;
  m.id = mixer_id;
  g->mixers[g->mixer_parent[mixer_id]]->message_queue.push_back(std::move(m));
  // tracking maps: the mixer, everything nested under it, their effects and sources (the objects die with the SubMixerProcessor)
  std::vector<int> gone(1, mixer_id);
  for (size_t i = 0; i < gone.size(); ++i) for (auto& kv : g->mixer_parent) if (kv.second == gone[i] && g->mixers.count(kv.first)) gone.push_back(kv.first);
  for (int id : gone) {
    MixedSource* ms = g->mixers[id];
    for (auto e = g->effect_mixer.begin(); e != g->effect_mixer.end();) { if (e->second == ms) e = g->effect_mixer.erase(e); else ++e; }
    for (auto v = g->voice_mixer.begin(); v != g->voice_mixer.end();) { if (v->second == ms) v = g->voice_mixer.erase(v); else ++v; }
    g->mixers.erase(id);
  }
  return PG_OK;
}
int po_graph_remove_effect(po_graph* g, int effect_id) {  // Player::remove_effect -> MixerMessage::RemoveEffect
  auto it = g->effect_mixer.find(effect_id);
  if (it == g->effect_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::Message m;
  m.kind = MixedSource::Message::RemoveEffect;
  m.id = effect_id;
  it->second->message_queue.push_back(std::move(m));
  g->effect_mixer.erase(it);
  return PG_OK;
}
int po_graph_move_effect(po_graph* g, int effect_id, int mixer_id, int movement, int offset) {  // Player::move_effect -> MixerMessage::MoveEffect
  auto it = g->effect_mixer.find(effect_id);
  if (it == g->effect_mixer.end()) return PG_ERR_NOT_FOUND;
  auto mx = g->mixers.find(mixer_id);
  if (mx == g->mixers.end() || mx->second != it->second) return PG_ERR_PARAMETER;  // "Effect does not belong to mixer" (player.rs:953-958)
  if (movement < 0 || movement > 2) return PG_ERR_PARAMETER;
  MixedSource::Message m;
  m.kind = MixedSource::Message::MoveEffect;
  m.id = effect_id; m.movement = movement; m.offset = offset;
  it->second->message_queue.push_back(std::move(m));
  return PG_OK;
}
int po_graph_schedule_param(po_graph* g, int effect_id, uint32_t id, float value, int normalized, uint64_t sample_time) {
  auto it = g->effect_mixer.find(effect_id);
  if (it == g->effect_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::MixerEvent ev;
  ev.kind = MixedSource::MixerEvent::EffectParam; ev.id = effect_id; ev.sample_time = sample_time;
  ev.param_id = id; ev.update = ParamUpdate{normalized != 0, value};
  return push_event(it->second, ev);
}
int po_graph_schedule_reset(po_graph* g, int effect_id, uint64_t sample_time) {
  auto it = g->effect_mixer.find(effect_id);
  if (it == g->effect_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::MixerEvent ev;
  ev.kind = MixedSource::MixerEvent::EffectReset; ev.id = effect_id; ev.sample_time = sample_time;
  return push_event(it->second, ev);
}
int po_graph_set_voice_volume(po_graph* g, int voice, float v, uint64_t sample_time) {
  auto it = g->voice_mixer.find(voice);
  if (it == g->voice_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::MixerEvent ev;
  ev.kind = MixedSource::MixerEvent::SetSourceVolume; ev.id = voice; ev.sample_time = sample_time; ev.f = v;
  return push_event(it->second, ev);
}
int po_graph_set_voice_panning(po_graph* g, int voice, float v, uint64_t sample_time) {
  auto it = g->voice_mixer.find(voice);
  if (it == g->voice_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::MixerEvent ev;
  ev.kind = MixedSource::MixerEvent::SetSourcePanning; ev.id = voice; ev.sample_time = sample_time; ev.f = v;
  return push_event(it->second, ev);
}
int po_graph_remove_voice(po_graph* g, int voice) {  // MixerMessage::RemoveSource (mixed.rs:149-151,400-402)
  auto it = g->voice_mixer.find(voice);
  if (it == g->voice_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::Message m;
  m.kind = MixedSource::Message::RemoveSource; m.id = voice;
  it->second->message_queue.push_back(std::move(m));
  g->voice_mixer.erase(it);   // (Player::remove_generator drops the handle's entry: later calls with this id fail)
  return PG_OK;
}
int po_graph_stop_voice(po_graph* g, int voice, uint64_t sample_time) {
  auto it = g->voice_mixer.find(voice);
  if (it == g->voice_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::Message m;
  m.kind = MixedSource::Message::StopSource; m.id = voice; m.sample_time = sample_time;
  it->second->message_queue.push_back(std::move(m));
  return PG_OK;
}
int po_graph_set_voice_speed(po_graph* g, int voice, double speed, float glide, uint64_t sample_time) {
  auto it = g->voice_mixer.find(voice);
  if (it == g->voice_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::MixerEvent ev;
  ev.kind = MixedSource::MixerEvent::SetSourceSpeed; ev.id = voice; ev.sample_time = sample_time; ev.a = speed; ev.f = glide; ev.flag = glide > 0.0f;
  return push_event(it->second, ev);
}
int po_graph_seek_voice(po_graph* g, int voice, double seconds, uint64_t sample_time) {
  auto it = g->voice_mixer.find(voice);
  if (it == g->voice_mixer.end()) return PG_ERR_NOT_FOUND;
  MixedSource::MixerEvent ev;
  ev.kind = MixedSource::MixerEvent::SeekSource; ev.id = voice; ev.sample_time = sample_time; ev.a = seconds;
  return push_event(it->second, ev);
}
size_t po_graph_write(po_graph* g, float* out, size_t n_samples, uint64_t pos) { return g->main->write(out, n_samples, pos); }

// CPU baseline helper for bench.py: runs `n_blocks` write() calls of `block_samples` on each of `n_graphs`
// independent graphs with `threads` worker threads (graphs are distributed round-robin: the reference's
// sub-mixer thread pool distributes independent sub-mixers the same way, thread_pool.rs:92-121). Each
// graph's output is summed into `out` (n_blocks*block_samples) by the caller thread afterwards.
int po_graphs_render_parallel(po_graph** graphs, int n_graphs, int threads, float* outs, size_t block_samples, size_t n_blocks,
                              uint64_t start_pos) {
  if (threads < 1) threads = 1;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([=]() {
      for (int gi = t; gi < n_graphs; gi += threads) {
        float* o = outs + (size_t)gi * block_samples * n_blocks;
        uint64_t pos = start_pos;
        for (size_t b = 0; b < n_blocks; ++b) {
          graphs[gi]->main->write(o + b * block_samples, block_samples, pos);
          pos += block_samples / graphs[gi]->channels;
        }
      }
    });
  }
  for (auto& th : pool) th.join();
  return PG_OK;
}

// index-stream log (test hook): arm, run effects on this thread, collect
static thread_local std::vector<int32_t> g_index_log_storage;
void po_index_log_begin(void) { g_index_log_storage.clear(); index_log() = &g_index_log_storage; }
size_t po_index_log_end(int32_t* out, size_t cap) {
  index_log() = nullptr;
  size_t n = g_index_log_storage.size();
  for (size_t i = 0; i < n && i < cap; ++i) out[i] = g_index_log_storage[i];
  return n;
}

// knee-edge log (test hook, po_utils.hpp): arm, render on this thread, collect the sample times
static thread_local std::vector<uint64_t> g_knee_log_storage;
void po_knee_log_begin(void) { g_knee_log_storage.clear(); knee_log() = &g_knee_log_storage; }
size_t po_knee_log_end(uint64_t* out, size_t cap) {
  knee_log() = nullptr;
  size_t n = g_knee_log_storage.size();
  for (size_t i = 0; i < n && i < cap; ++i) out[i] = g_knee_log_storage[i];
  return n;
}

#ifndef PO_BUILD_FLAGS
#define PO_BUILD_FLAGS "unknown"
#endif
const char* po_build_flags(void) { return PO_BUILD_FLAGS; }  // compiler flags of this build (reported by bench.py's cpu_baseline)

// ---- primitive probes for the KAT tests ------------------------------------------------------
void po_clear_buffer(float* d, size_t n) { clear_buffer(d, n); }
void po_scale_buffer(float* d, size_t n, float v) { scale_buffer(d, n, v); }
void po_add_buffers(float* d, const float* s, size_t n) { add_buffers(d, s, n); }
void po_copy_buffers(float* d, const float* s, size_t n) { copy_buffers(d, s, n); }
float po_max_abs_sample(const float* b, size_t n) { return max_abs_sample(b, n); }
void po_remap_buffer_channels(const float* in, size_t in_ch, float* out, size_t out_ch, size_t frames) { remap_buffer_channels(in, in_ch, out, out_ch, frames); }
float po_db_to_linear(float v) { return db_to_linear(v); }
float po_linear_to_db(float v) { return linear_to_db(v); }
void po_panning_factors(float p, float* l, float* r) { panning_factors(p, *l, *r); }
float po_sine_approx(float x) { return sine_approx(x); }
// the generator behind the LFO's random shapes: n x next_u64 and n x random::<f32>() from the given Xoshiro256++ state (known-answer tests)
void po_small_rng_run(const uint64_t state[4], size_t n, uint64_t* out_u64, float* out_f32) {
  SmallRng a(state), b(state);
  for (size_t i = 0; i < n; ++i) { if (out_u64) out_u64[i] = a.next_u64(); if (out_f32) out_f32[i] = b.random_f32(); }
}

// smoothers: kind 0 = exponential(inertia arg), 1 = linear(step arg), 2 = spring(duration arg)
// ops are replayed on a fresh smoother created with (value, sample_rate); returns current after `n_ramps` ramps
// and writes {current, target, need_ramp, pending_steps/velocity}.
void po_smoother_run(int kind, float init, uint32_t sample_rate, float arg, int has_arg, float target, int has_duration, uint32_t duration,
                     uint32_t n_ramps, float* trace, float* out4) {
  if (kind == 0) {
    ExponentialSmoothedValue s(init, sample_rate);
    if (has_arg) s.inertia_ = arg;
    s.set_target(target);
    for (uint32_t i = 0; i < n_ramps; ++i) { s.ramp(); if (trace) trace[i] = s.current(); }
    out4[0] = s.current(); out4[1] = s.target(); out4[2] = s.need_ramp() ? 1.0f : 0.0f; out4[3] = 0.0f;
  } else if (kind == 1) {
    LinearSmoothedValue s(init, sample_rate);
    if (has_arg) s.set_step(arg);
    s.set_target_with_duration(target, has_duration != 0, duration);
    float pending0 = (float)s.num_pending_steps;
    for (uint32_t i = 0; i < n_ramps; ++i) { s.ramp(); if (trace) trace[i] = s.current(); }
    out4[0] = s.current(); out4[1] = s.target(); out4[2] = s.need_ramp() ? 1.0f : 0.0f; out4[3] = pending0;
  } else {
    SpringSmoothedValue s(init, sample_rate);
    if (has_arg) s = SpringSmoothedValue(init, sample_rate).with_duration((size_t)arg);
    s.set_target(target);
    for (uint32_t i = 0; i < n_ramps; ++i) { s.ramp(); if (trace) trace[i] = s.current(); }
    out4[0] = s.current(); out4[1] = s.target(); out4[2] = s.need_ramp() ? 1.0f : 0.0f; out4[3] = s.velocity();
  }
}

// biquad / svf / dc single-filter probes (analytic known answers)
void po_biquad_run(int type, uint32_t sr, float cutoff, float q, float gain, float* buf, size_t n) {
  BiquadCoefficients c;
  c.set((BiquadType)type, sr, cutoff, q, gain);
  BiquadFilter f;
  for (size_t i = 0; i < n; ++i) buf[i] = (float)f.process_sample(c, (double)buf[i]);
}
void po_svf_run(int type, uint32_t sr, float cutoff, float res, float* buf, size_t n) {
  SvfCoefficients c;
  c.set((SvfType)type, sr, cutoff, res);
  SvfFilter f;
  for (size_t i = 0; i < n; ++i) buf[i] = (float)f.process_sample(c, (double)buf[i]);
}
void po_dc_run(int mode, uint32_t sr, float* buf, size_t n) {
  DcFilter f(sr, (DcMode)mode);
  for (size_t i = 0; i < n; ++i) buf[i] = (float)f.process_sample((double)buf[i]);
}
// cubic resampler over an interleaved buffer, repeated process() calls with `out_chunk` sized outputs;
// returns produced samples, writes consumed samples to *consumed
size_t po_cubic_resample(const float* in, size_t in_len, uint32_t in_rate, uint32_t out_rate, size_t channels, float* out, size_t out_len,
                         size_t out_chunk, size_t* consumed_total) {
  CubicResampler r(in_rate, out_rate, channels);
  size_t consumed = 0, produced = 0;
  while (produced < out_len) {
    size_t c, p;
    size_t want = std::min(out_chunk, out_len - produced);
    r.process(in + consumed, in_len - consumed, out + produced, want, c, p);
    consumed += c; produced += p;
    if (p == 0) break;
  }
  *consumed_total = consumed;
  return produced;
}
void po_allpass_run(size_t size, size_t delay, double* buf /*[n][2]*/, size_t n) {
  AllpassDelayLine<2> a(size);
  a.set_delay(delay);
  for (size_t i = 0; i < n; ++i) { double o[2]; a.process(buf + 2 * i, o); buf[2 * i] = o[0]; buf[2 * i + 1] = o[1]; }
}
void po_interp_delay_run(size_t max_size, float feedback, float delay, float* buf, size_t n) {
  InterpolatedDelayLine<1> d(max_size);
  for (size_t i = 0; i < n; ++i) { float o; d.process(buf + i, feedback, delay, &o); buf[i] = o; }
}

}  // extern "C"
