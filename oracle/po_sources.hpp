// ORACLE — TEST INFRASTRUCTURE ONLY (see po_utils.hpp header). Parity pinned only by the loose
// reference test src/source/file/preloaded.rs:486-533 (resampling) — otherwise unpinned.
//
// CPU restatement of src/source/{file/preloaded,file/common,amplified,panned,mapped,converted,
// resampled,mixed}.rs, src/source/mixed/{effect,submixer}.rs and src/utils/event.rs.
#pragma once
#include <deque>
#include <memory>

#include "po_effects.hpp"

namespace po {

struct Source {  // trait Source, src/source.rs:80-110
  virtual ~Source() {}
  virtual size_t channel_count() const = 0;
  virtual uint32_t sample_rate() const = 0;
  virtual bool is_exhausted() const = 0;
  virtual size_t weight() const = 0;
  virtual size_t write(float* output, size_t n_samples, uint64_t pos_in_frames) = 0;
};

constexpr size_t MAX_MIX_BUFFER_SAMPLES = 8 * 1024;  // src/source/mixed.rs:216

// ---- src/source/file/buffer.rs:13-150 --------------------------------------------------------
struct AudioFileBuffer {
  std::vector<float> buffer;  // interleaved, including the extra zero frame (:103-104)
  uint32_t sample_rate = 0;
  size_t channel_count = 0;
  bool has_loop = false;
  size_t loop_start = 0, loop_end = 0;  // frames
  size_t frame_count() const { return buffer.size() / channel_count; }
};

struct FileOptions {  // FilePlaybackOptions, src/source/file.rs:34-112
  float volume = 1.0f, panning = 0.0f;
  double speed = 1.0;
  bool has_repeat = false;
  size_t repeat = 0;
  bool has_loop_range = false;
  uint64_t loop_start = 0, loop_end = 0;
  float fade_in_seconds = 0.0f;    // None
  float fade_out_seconds = 0.05f;  // Some(50 ms); < 0 = None
};

// ---- src/source/file/preloaded.rs + file/common.rs (cubic resampler only; rubato excluded) ----
struct PreloadedFileSource : Source {
  enum class Msg { Seek, SetSpeed, Stop, Kill };
  struct Message { Msg kind; double a; float glide; bool has_glide; };
  std::shared_ptr<AudioFileBuffer> file_buffer;
  // FileSourceImpl (common.rs:30-139)
  FileOptions options;
  VolumeFader volume_fader;
  CubicResampler resampler;
  float fade_out_seconds;
  uint32_t output_sample_rate;
  size_t output_channel_count;
  std::vector<Message> message_queue;
  bool playback_finished = false;
  size_t samples_to_next_speed_update = 0;
  float speed_glide_rate = 0.0f;
  double current_speed, target_speed;
  // PreloadedFileSource (preloaded.rs:29-37)
  size_t playback_repeat, playback_repeat_count, playback_pos = 0;
  bool playback_pos_eof = false;
  bool has_loop_override = false;
  uint64_t loop_override_start = 0, loop_override_end = 0;
  static constexpr size_t SPEED_UPDATE_CHUNK_SIZE = 64;  // common.rs:56

  PreloadedFileSource(std::shared_ptr<AudioFileBuffer> fb, const FileOptions& opt, uint32_t out_rate)  // from_shared_buffer :71-117
      : file_buffer(fb), options(opt), volume_fader(fb->channel_count, out_rate),
        resampler(fb->sample_rate, as_u32((double)out_rate / opt.speed), fb->channel_count),  // common.rs:78-86
        fade_out_seconds(opt.fade_out_seconds), output_sample_rate(out_rate), output_channel_count(fb->channel_count),
        current_speed(opt.speed), target_speed(opt.speed) {
    if (opt.fade_in_seconds > 0.0f) volume_fader.start_fade_in(opt.fade_in_seconds);  // common.rs:70-75
    playback_repeat = opt.has_repeat ? opt.repeat : (fb->has_loop ? USIZE_MAX : 0);    // :89-95
    playback_repeat_count = playback_repeat;
    if (opt.has_loop_range) {  // :101-104
      uint64_t fc = (uint64_t)fb->frame_count();
      has_loop_override = true;
      loop_override_start = std::min(opt.loop_start, fc > 0 ? fc - 1 : 0);
      loop_override_end = std::min(opt.loop_end, fc);
    }
  }
  size_t channel_count() const override { return output_channel_count; }
  uint32_t sample_rate() const override { return output_sample_rate; }
  bool is_exhausted() const override { return playback_finished; }
  size_t weight() const override { return 1; }

  bool loop_range(uint64_t& s, uint64_t& e) const {  // :147-153
    if (has_loop_override) { s = loop_override_start; e = loop_override_end; return true; }
    if (file_buffer->has_loop) { s = file_buffer->loop_start; e = file_buffer->loop_end; return true; }
    return false;
  }
  void update_speed(uint32_t input_sample_rate) {  // common.rs:141-169
    double speed_diff = target_speed - current_speed;
    if (speed_glide_rate > 0.0f && std::fabs(speed_diff) > 0.0001) {
      double semitone_diff = std::fabs(12.0 * std::log2(target_speed / current_speed));
      float duration_secs = (float)semitone_diff / speed_glide_rate;
      if (duration_secs > 0.0f) {
        float duration_frames = duration_secs * (float)output_sample_rate;
        double speed_step_per_frame = (target_speed - current_speed) / (double)duration_frames;
        double speed_change_this_call = speed_step_per_frame * (double)SPEED_UPDATE_CHUNK_SIZE;
        if (std::fabs(target_speed - current_speed) < std::fabs(speed_change_this_call)) current_speed = target_speed;
        else current_speed += speed_change_this_call;
      } else current_speed = target_speed;
    } else current_speed = target_speed;
    uint32_t new_output_rate = as_u32((double)output_sample_rate / current_speed);
    resampler.update(input_sample_rate, new_output_rate);
  }
  void seek(double position_secs) {  // :139-147
    if (!is_exhausted()) {
      double buffer_pos = position_secs * (double)file_buffer->sample_rate * (double)file_buffer->channel_count;
      playback_pos = std::min(as_usize(buffer_pos), file_buffer->buffer.size());
      resampler.reset();
    }
  }
  void set_speed(double speed, bool has_glide, float glide) {  // :181-192
    if (!is_exhausted()) {
      samples_to_next_speed_update = 0;
      target_speed = speed;
      speed_glide_rate = has_glide ? glide : 0.0f;
      if (speed_glide_rate == 0.0f) { current_speed = speed; update_speed(file_buffer->sample_rate); }
    }
  }
  void stop() {  // :195-208
    if (!is_exhausted()) {
      if (fade_out_seconds > 0.0f) volume_fader.start_fade_out(fade_out_seconds);
      else playback_finished = true;
    }
  }
  void kill() { if (!is_exhausted()) playback_finished = true; }  // :232-238
  void process_messages() {  // :250-268
    std::vector<Message> q;
    q.swap(message_queue);
    for (auto& m : q) {
      switch (m.kind) {
        case Msg::Seek: seek(m.a); break;
        case Msg::SetSpeed: set_speed(m.a, m.has_glide, m.glide); break;
        case Msg::Stop: stop(); break;
        case Msg::Kill: kill(); break;
      }
    }
  }
  size_t write_buffer(float* output, size_t out_len) {  // :270-332
    size_t written = 0;
    size_t buf_len = file_buffer->buffer.size();
    size_t lr_start = 0, lr_end = buf_len;
    if (playback_repeat > 0) {
      uint64_t s, e;
      if (loop_range(s, e)) { lr_start = (size_t)s * file_buffer->channel_count; lr_end = (size_t)e * file_buffer->channel_count; }
    }
    while (written < out_len) {
      size_t remaining_input_len = lr_end > playback_pos ? lr_end - playback_pos : 0;  // saturating_sub
      const float* remaining_input = file_buffer->buffer.data() + playback_pos;
      size_t input_consumed, output_written;
      // cubic: required_input_buffer_size() == None -> always the direct branch (:304-307)
      resampler.process(remaining_input, remaining_input_len, output + written, out_len - written, input_consumed, output_written);
      playback_pos += input_consumed;
      written += output_written;
      if (playback_pos >= lr_end) {
        if (playback_repeat_count > 0) {
          if (playback_repeat_count != USIZE_MAX) playback_repeat_count -= 1;
          playback_pos = lr_start;
        } else playback_pos_eof = true;
      }
      if (playback_pos_eof && output_written == 0) break;
    }
    return written;
  }
  size_t write(float* output, size_t out_len, uint64_t) override {  // :396-475
    process_messages();
    if (playback_finished) return 0;
    size_t total_written = 0;
    if (current_speed != target_speed) {
      while (total_written < out_len) {
        if (samples_to_next_speed_update == 0) {
          if (current_speed != target_speed) update_speed(file_buffer->sample_rate);
          samples_to_next_speed_update = SPEED_UPDATE_CHUNK_SIZE * output_channel_count;
        }
        size_t chunk_length = std::min(out_len - total_written, samples_to_next_speed_update);
        size_t written = write_buffer(output + total_written, chunk_length);
        samples_to_next_speed_update -= written;
        total_written += written;
        if (written < chunk_length) break;
      }
    } else {
      samples_to_next_speed_update = 0;
      total_written = write_buffer(output, out_len);
    }
    volume_fader.process(output, total_written);
    bool fade_out_completed = volume_fader.state == FaderState::Finished && volume_fader.target_volume() == 0.0f;
    if (playback_pos_eof || fade_out_completed) playback_finished = true;
    return total_written;
  }
};

// ---- src/source/mapped.rs --------------------------------------------------------------------
struct ChannelMappedSource : Source {
  std::unique_ptr<Source> source;
  size_t input_channels, output_channels;
  std::vector<float> input_buffer;
  ChannelMappedSource(std::unique_ptr<Source> s, size_t out_ch) : source(std::move(s)), output_channels(out_ch) {
    input_channels = source->channel_count();
    input_buffer.assign(MAX_MIX_BUFFER_SAMPLES / output_channels * input_channels, 0.0f);  // :25
  }
  size_t channel_count() const override { return output_channels; }
  uint32_t sample_rate() const override { return source->sample_rate(); }
  bool is_exhausted() const override { return source->is_exhausted(); }
  size_t weight() const override { return source->weight(); }
  size_t write(float* output, size_t out_len, uint64_t pos) override {  // :61-99
    if (out_len == 0 || input_channels == output_channels) return source->write(output, out_len, pos);
    size_t total_written = 0;
    while (total_written < out_len) {
      size_t input_max = ((out_len - total_written) / output_channels) * input_channels;
      size_t buffer_max = std::min(input_max, input_buffer.size());
      uint64_t source_time = pos + (uint64_t)(total_written / output_channels);
      size_t written = source->write(input_buffer.data(), buffer_max, source_time);
      if (written == 0) break;
      size_t out_chunk = (written / input_channels) * output_channels;
      remap_buffer_channels(input_buffer.data(), input_channels, output + total_written, output_channels, written / input_channels);
      total_written += out_chunk;
    }
    return total_written;
  }
};

// ---- src/source/resampled.rs (cubic only) -----------------------------------------------------
struct ResampledSource : Source {
  std::unique_ptr<Source> source;
  std::unique_ptr<CubicResampler> resampler;
  uint32_t output_sample_rate;
  TempBuffer input_buffer, output_buffer;
  ResampledSource(std::unique_ptr<Source> s, uint32_t out_rate) : source(std::move(s)), output_sample_rate(out_rate) {  // :44-98
    const size_t DEFAULT_CHUNK_SIZE = 512;
    if (source->sample_rate() != out_rate) {
      resampler.reset(new CubicResampler(source->sample_rate(), out_rate, source->channel_count()));
      input_buffer = TempBuffer(DEFAULT_CHUNK_SIZE * source->channel_count());
      output_buffer = TempBuffer(DEFAULT_CHUNK_SIZE * source->channel_count());
    }
  }
  size_t channel_count() const override { return source->channel_count(); }
  uint32_t sample_rate() const override { return output_sample_rate; }
  bool is_exhausted() const override { return source->is_exhausted() && input_buffer.is_empty() && output_buffer.is_empty(); }
  size_t weight() const override { return source->weight() + (resampler ? 1 : 0); }
  size_t write(float* output, size_t out_len, uint64_t pos) override {  // :101-152
    if (!resampler) return source->write(output, out_len, pos);
    if (out_len == 0) return source->write(output, out_len, pos);
    size_t total_written = 0;
    while (total_written < out_len) {
      if (output_buffer.is_empty()) {
        output_buffer.reset_range();
        if (input_buffer.is_empty()) {
          uint64_t source_time = pos + (uint64_t)(total_written / source->channel_count());
          input_buffer.reset_range();
          size_t input_read = source->write(input_buffer.get(), input_buffer.len(), source_time);
          // cubic has no required input size: the zero-padding branch (:120-127) never triggers, but the range
          // is NOT shrunk to input_read either — the reference resamples the whole (stale-tailed) buffer.
          (void)input_read;
        }
        size_t input_consumed, output_written;
        resampler->process(input_buffer.get(), input_buffer.len(), output_buffer.get(), output_buffer.len(), input_consumed, output_written);
        input_buffer.consume(input_consumed);
        output_buffer.set_range(0, output_written);
        if (source->is_exhausted() && output_written == 0) break;
      }
      size_t written = output_buffer.copy_to(output + total_written, out_len - total_written);
      output_buffer.consume(written);
      total_written += written;
    }
    return total_written;
  }
};

// ---- src/source/amplified.rs / panned.rs -----------------------------------------------------
struct AmplifiedSource : Source {
  std::unique_ptr<Source> source;
  ExponentialSmoothedValue volume;
  bool has_message = false;  // ArrayQueue capacity 1, force_push (mixed.rs:815-822)
  float message_volume = 0.0f;
  AmplifiedSource(std::unique_ptr<Source> s, float vol) : source(std::move(s)), volume(vol, source->sample_rate()) {}
  size_t channel_count() const override { return source->channel_count(); }
  uint32_t sample_rate() const override { return source->sample_rate(); }
  bool is_exhausted() const override { return source->is_exhausted(); }
  size_t weight() const override { return source->weight(); }
  size_t write(float* output, size_t n, uint64_t pos) override {  // :93-104
    if (has_message) { volume.set_target(message_volume); has_message = false; }
    size_t written = source->write(output, n, pos);
    apply_smoothed_gain(output, written, volume);
    return written;
  }
};
struct PannedSource : Source {
  std::unique_ptr<Source> source;
  ExponentialSmoothedValue panning;
  bool has_message = false;
  float message_panning = 0.0f;
  PannedSource(std::unique_ptr<Source> s, float pan) : source(std::move(s)), panning(pan, source->sample_rate()) {}
  size_t channel_count() const override { return source->channel_count(); }
  uint32_t sample_rate() const override { return source->sample_rate(); }
  bool is_exhausted() const override { return source->is_exhausted(); }
  size_t weight() const override { return source->weight(); }
  size_t write(float* output, size_t n, uint64_t pos) override {  // :93-104
    if (has_message) { panning.set_target(message_panning); has_message = false; }
    size_t written = source->write(output, n, pos);
    apply_smoothed_panning(output, written, source->channel_count(), panning);
    return written;
  }
};

// ---- src/source/mixed/effect.rs --------------------------------------------------------------
struct EffectProcessor {
  static constexpr float SILENCE_THRESHOLD = 0.001f;
  static constexpr size_t SILENCE_SECONDS = 2;
  std::unique_ptr<Effect> effect;
  bool bypassed = true;
  size_t tail_counter = 0, silence_counter = USIZE_MAX;
  explicit EffectProcessor(std::unique_ptr<Effect> e) : effect(std::move(e)) {}
  size_t weight() const { return !bypassed ? effect->weight() : 1; }
  bool should_bypass(bool input_bypassed) const { return input_bypassed && tail_counter == 0 && silence_counter == USIZE_MAX; }  // :88-91
  void reset_tail_counters() { tail_counter = USIZE_MAX; silence_counter = 0; }  // :148-152
  void update_bypass_state(bool should) {  // :94-108
    if (should && !bypassed) { effect->process_stopped(); bypassed = true; }
    else if (!should && bypassed) { effect->process_started(); bypassed = false; reset_tail_counters(); }
  }
  void update_tail_counters(const float* output, size_t n, size_t channel_count, uint32_t sample_rate) {  // :111-145
    size_t tail_frames;
    if (effect->process_tail(tail_frames)) {
      if (tail_frames == USIZE_MAX) tail_counter = tail_frames;
      else {
        if (tail_counter == USIZE_MAX) tail_counter = tail_frames;
        else {
          size_t frames_processed = n / channel_count;
          tail_counter = tail_counter > frames_processed ? tail_counter - frames_processed : 0;
        }
      }
      silence_counter = USIZE_MAX;
    } else {
      float max_sample = max_abs_sample(output, n);
      if (max_sample < SILENCE_THRESHOLD) {
        size_t frames_processed = n / channel_count;
        silence_counter = (silence_counter > USIZE_MAX - frames_processed) ? USIZE_MAX : silence_counter + frames_processed;
        if (silence_counter >= SILENCE_SECONDS * (size_t)sample_rate) { tail_counter = 0; silence_counter = USIZE_MAX; }
      } else silence_counter = 0;
    }
  }
  bool process(float* output, size_t n, size_t channel_count, uint32_t sample_rate, bool input_bypassed) {  // :56-84
    update_bypass_state(should_bypass(input_bypassed));
    if (!bypassed) {
      effect->process(output, n);
      if (input_bypassed) update_tail_counters(output, n, channel_count, sample_rate);
      else reset_tail_counters();
      return true;
    }
    return false;
  }
};

// ---- src/source/mixed.rs ---------------------------------------------------------------------
struct MixedSource;
struct SubMixerProcessor {  // src/source/mixed/submixer.rs
  std::unique_ptr<MixedSource> mixer;
  size_t silence_counter = 0;
  bool process(float* output, float* mix_buffer, size_t n, size_t channel_count, uint32_t sample_rate, uint64_t pos);
  size_t weight() const;
};

struct PlaybackQueues {  // PlaybackMessageQueue::File
  PreloadedFileSource* file = nullptr;
  AmplifiedSource* amplified = nullptr;
  PannedSource* panned = nullptr;
};

struct MixedSource : Source {
  struct PlayingSource {
    bool is_active = true, is_transient = true;
    int playback_id = 0;
    PlaybackQueues queues;
    std::unique_ptr<Source> source;
    uint64_t start_time = 0;
    bool has_stop_time = false;
    uint64_t stop_time = 0;
  };
  struct MixerEvent {
    enum Kind { SeekSource, SetSourceSpeed, SetSourceVolume, SetSourcePanning, EffectReset, EffectParam } kind;
    int id;  // playback id or effect id
    uint64_t sample_time;
    double a = 0.0; float f = 0.0f; bool flag = false;
    uint32_t param_id = 0; ParamUpdate update{false, 0.0f};
  };
  struct Message {
    enum Kind { RemoveAllPendingEvents, AddSource, StopSource, RemoveSource, AddMixer, RemoveMixer, AddEffect, RemoveEffect, MoveEffect, Event } kind;
    int movement = 0, offset = 0;  // EffectMovement: 0 = Direction(offset), 1 = Start, 2 = End (src/player.rs)
    std::unique_ptr<PlayingSource> source;
    int id = 0; uint64_t sample_time = 0;
    std::unique_ptr<SubMixerProcessor> mixer;
    std::unique_ptr<EffectProcessor> effect;
    MixerEvent event;
  };
  std::deque<std::unique_ptr<PlayingSource>> playing_sources;
  std::vector<std::pair<int, std::unique_ptr<SubMixerProcessor>>> mixers;
  std::vector<std::pair<int, std::unique_ptr<EffectProcessor>>> effects;
  bool effects_bypassed = true;
  std::vector<Message> message_queue;
  std::deque<MixerEvent> events;
  size_t channel_count_;
  uint32_t sample_rate_;
  std::vector<float> mix_buffer;

  MixedSource(size_t ch, uint32_t sr) : channel_count_(ch), sample_rate_(sr), mix_buffer(MAX_MIX_BUFFER_SAMPLES, 0.0f) {}
  size_t channel_count() const override { return channel_count_; }
  uint32_t sample_rate() const override { return sample_rate_; }
  bool is_exhausted() const override { return false; }
  size_t weight() const override {
    size_t w = 0;
    for (auto& s : playing_sources) w += s->source->weight();
    for (auto& e : effects) w += e.second->weight();
    for (auto& m : mixers) w += m.second->weight();
    return w;
  }
  void insert_event(const MixerEvent& ev) {  // src/utils/event.rs:31-38
    size_t pos = 0;
    while (pos < events.size() && events[pos].sample_time <= ev.sample_time) ++pos;
    events.insert(events.begin() + pos, ev);
  }
  size_t time_until_next_event(uint64_t current_time) const {  // event.rs:24-28
    if (events.empty()) return USIZE_MAX;
    return (size_t)(events.front().sample_time - current_time);
  }
  PlayingSource* find_source(int id) { for (auto& s : playing_sources) if (s->playback_id == id) return s.get(); return nullptr; }
  EffectProcessor* find_effect(int id) { for (auto& e : effects) if (e.first == id) return e.second.get(); return nullptr; }
  void process_event(const MixerEvent& ev) {  // :761-924
    switch (ev.kind) {
      case MixerEvent::SeekSource: if (auto s = find_source(ev.id)) if (s->queues.file) s->queues.file->message_queue.push_back({PreloadedFileSource::Msg::Seek, ev.a, 0.0f, false}); break;
      case MixerEvent::SetSourceSpeed: if (auto s = find_source(ev.id)) if (s->queues.file) s->queues.file->message_queue.push_back({PreloadedFileSource::Msg::SetSpeed, ev.a, ev.f, ev.flag}); break;
      case MixerEvent::SetSourceVolume: if (auto s = find_source(ev.id)) if (s->queues.amplified) { s->queues.amplified->has_message = true; s->queues.amplified->message_volume = ev.f; } break;
      case MixerEvent::SetSourcePanning: if (auto s = find_source(ev.id)) if (s->queues.panned) { s->queues.panned->has_message = true; s->queues.panned->message_panning = ev.f; } break;
      case MixerEvent::EffectReset: if (auto e = find_effect(ev.id)) e->effect->process_reset_message(); break;
      case MixerEvent::EffectParam: if (auto e = find_effect(ev.id)) e->effect->process_parameter_update(ev.param_id, ev.update); break;
    }
  }
  void process_events(uint64_t current_time) {  // event.rs:41-50
    while (!events.empty() && events.front().sample_time <= current_time) {
      MixerEvent ev = events.front();
      events.pop_front();
      process_event(ev);
    }
  }
  void process_messages(uint64_t pos_in_frames) {  // :294-499
    std::vector<Message> q;
    q.swap(message_queue);
    for (auto& m : q) {
      switch (m.kind) {
        case Message::RemoveAllPendingEvents: {  // :298-305
          for (size_t i = 0; i < playing_sources.size();) { if (playing_sources[i]->is_transient && playing_sources[i]->start_time > pos_in_frames) playing_sources.erase(playing_sources.begin() + i); else ++i; }
          for (size_t i = 0; i < events.size();) { if (events[i].sample_time > pos_in_frames) events.erase(events.begin() + i); else ++i; }
        } break;
        case Message::AddSource: {
          size_t insert_pos = 0;  // partition_point(|e| e.start_time < sample_time) :326-329
          while (insert_pos < playing_sources.size() && playing_sources[insert_pos]->start_time < m.source->start_time) ++insert_pos;
          playing_sources.insert(playing_sources.begin() + insert_pos, std::move(m.source));
        } break;
        case Message::StopSource: if (auto s = find_source(m.id)) { s->has_stop_time = true; s->stop_time = m.sample_time; } break;
        case Message::RemoveSource:
          for (size_t i = 0; i < playing_sources.size();) { if (playing_sources[i]->playback_id == m.id) playing_sources.erase(playing_sources.begin() + i); else ++i; }
          break;
        case Message::AddMixer: mixers.emplace_back(m.id, std::move(m.mixer)); break;
        case Message::RemoveMixer:  // :422-424
          for (size_t i = 0; i < mixers.size();) { if (mixers[i].first == m.id) mixers.erase(mixers.begin() + i); else ++i; }
          break;
        case Message::AddEffect: effects.emplace_back(m.id, std::move(m.effect)); effects_bypassed = false; break;
        case Message::RemoveEffect: {  // :433-440
          for (size_t pos = 0; pos < effects.size(); ++pos) if (effects[pos].first == m.id) {
            effects.erase(effects.begin() + pos);
            if (effects.empty()) effects_bypassed = true;
            break;
          }
        } break;
        case Message::MoveEffect: {  // :441-462
          for (size_t current_pos = 0; current_pos < effects.size(); ++current_pos) if (effects[current_pos].first == m.id) {
            auto effect = std::move(effects[current_pos]);
            effects.erase(effects.begin() + current_pos);
            size_t new_pos;
            if (m.movement == 0) {
              int target = (int)current_pos + m.offset;
              int hi = (int)effects.size();
              new_pos = (size_t)(target < 0 ? 0 : (target > hi ? hi : target));
            } else new_pos = m.movement == 1 ? 0 : effects.size();
            effects.insert(effects.begin() + new_pos, std::move(effect));
            break;
          }
        } break;
        case Message::Event: insert_event(m.event); break;
      }
    }
  }
  bool process_sub_mixers(float* output, size_t n, uint64_t pos) {  // sequential branch :539-553
    bool produced = false;
    for (auto& m : mixers) produced |= m.second->process(output, mix_buffer.data(), n, channel_count_, sample_rate_, pos);
    return produced;
  }
  bool process_sources(float* output, size_t out_len, uint64_t pos) {  // :558-624
    bool produced_output = false;
    size_t output_frame_count = out_len / channel_count_;
    for (auto& ps : playing_sources) {
      size_t total_written = 0;
      if (ps->start_time > pos) {
        size_t frames_until_source_starts = (size_t)(ps->start_time - pos);
        if (frames_until_source_starts > 0) {
          if (frames_until_source_starts >= output_frame_count) break;
          total_written += frames_until_source_starts * channel_count_;
        }
      }
      while (total_written < out_len) {
        uint64_t source_time = pos + (uint64_t)(total_written / channel_count_);
        uint64_t samples_until_stop = UINT64_MAX;
        if (ps->has_stop_time) {
          uint64_t d = ps->stop_time > source_time ? ps->stop_time - source_time : 0;
          samples_until_stop = d * (uint64_t)channel_count_;
        }
        if (samples_until_stop == 0) {
          if (ps->queues.file) ps->queues.file->message_queue.push_back({PreloadedFileSource::Msg::Stop, 0.0, 0.0f, false});
          ps->has_stop_time = false;
          samples_until_stop = UINT64_MAX;
        }
        size_t remaining = (size_t)std::min<uint64_t>((uint64_t)(out_len - total_written), samples_until_stop);
        size_t to_write = std::min(remaining, mix_buffer.size());
        size_t written = ps->source->write(mix_buffer.data(), to_write, source_time);
        add_buffers(output + total_written, mix_buffer.data(), written);
        total_written += written;
        produced_output |= written > 0;
        if (ps->is_transient && ps->source->is_exhausted()) { ps->is_active = false; break; }
        else if (written == 0) break;
      }
    }
    return produced_output;
  }
  void process_effects(float* output, size_t n, bool input_bypassed) {  // :627-655
    if (effects_bypassed && input_bypassed) return;
    bool all_bypassed = true;
    for (auto& e : effects) {
      bool is_active = e.second->process(output, n, channel_count_, sample_rate_, input_bypassed);
      if (is_active) { input_bypassed = false; all_bypassed = false; }
    }
    effects_bypassed = all_bypassed;
  }
  size_t write(float* output, size_t out_len, uint64_t pos) override {  // :659-719
    process_messages(pos);
    if (playing_sources.empty() && effects.empty() && mixers.empty() && events.empty()) return 0;
    clear_buffer(output, out_len);
    size_t output_frame_count = out_len / channel_count_;
    size_t total_frames_written = 0;
    while (total_frames_written < output_frame_count) {
      uint64_t current_time_in_frames = pos + (uint64_t)total_frames_written;
      process_events(current_time_in_frames);
      size_t frames_remaining = output_frame_count - total_frames_written;
      size_t frames_in_temp_out = mix_buffer.size() / channel_count_;
      size_t frames_until_next_event = time_until_next_event(current_time_in_frames);
      size_t frames_to_process = std::min(std::min(frames_remaining, frames_in_temp_out), frames_until_next_event);
      if (frames_to_process > 0) {
        uint64_t chunk_time = pos + (uint64_t)total_frames_written;
        float* chunk_output = output + total_frames_written * channel_count_;
        size_t chunk_len = frames_to_process * channel_count_;
        bool audible_input = process_sub_mixers(chunk_output, chunk_len, chunk_time);
        audible_input |= process_sources(chunk_output, chunk_len, chunk_time);
        chunk_time_now() = chunk_time;   // (test hook: po_utils.hpp log_knee_edge)
        process_effects(chunk_output, chunk_len, !audible_input);
        total_frames_written += frames_to_process;
      }
    }
    for (size_t i = 0; i < playing_sources.size();) {  // :715
      if (playing_sources[i]->is_transient && !playing_sources[i]->is_active) playing_sources.erase(playing_sources.begin() + i);
      else ++i;
    }
    return out_len;
  }
};

inline size_t SubMixerProcessor::weight() const { return mixer->weight(); }
inline bool SubMixerProcessor::process(float* output, float* mix_buffer, size_t n, size_t channel_count, uint32_t sample_rate, uint64_t pos) {  // submixer.rs:47-77
  size_t written = mixer->write(mix_buffer, n, pos);
  float max_sample = max_abs_sample(mix_buffer, written);
  if (max_sample < EffectProcessor::SILENCE_THRESHOLD) {
    size_t frames_processed = n / channel_count;
    silence_counter += frames_processed;
    if (silence_counter < EffectProcessor::SILENCE_SECONDS * (size_t)sample_rate) {
      add_buffers(output, mix_buffer, written);
      return true;
    }
    return false;
  }
  silence_counter = 0;
  add_buffers(output, mix_buffer, written);
  return true;
}

}  // namespace po
