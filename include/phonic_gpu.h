/*
 * phonic_gpu.h — C ABI of the MI355X-native phonic DSP hot path.
 *
 * This is the drop-in boundary: a Rust shim (`impl Effect`, `impl Source`,
 * see INTEGRATION.md) binds exactly these entry points. Every function cites the reference
 * interface it replaces (paths relative to the reference repo, emuell/phonic v0.16.0).
 *
 * Conventions
 *   - audio buffers are interleaved f32; all stock effects and the graph are stereo
 *     (reference: `enforce_stereo_playback`, src/player.rs:134,179).
 *   - threading (reference: src/source/mixed.rs:113-194,233-234,294-499): exactly one thread at a time may be inside
 *     process/write of a handle (the reference passes `&mut self`, src/effect.rs:155, src/source.rs:95), and the calls that CHANGE a graph
 *     (pg_graph_add_* / remove_* / move_effect / set_*) belong to that owner too. The CONTROL calls — pg_graph_schedule_param,
 *     pg_graph_schedule_reset, pg_graph_set_voice_volume / _panning / _speed, pg_graph_seek_voice, pg_graph_stop_voice,
 *     pg_graph_stop_all_voices — may be called from ANY thread at ANY time, concurrently with write and with each other: like the
 *     reference's handles they only push a record into a lock-free queue (PG_ERR_QUEUE_FULL when 65536 records wait), which write drains
 *     at its top exactly like MixedSource::process_messages.
 *   - all functions returning `int` return a pg_status; the message of the last failure
 *     on the calling thread is available from pg_last_error_message().
 *   - host/device allocations happen in create/initialize/add_* / set_* only (grow-by-doubling); process/write allocate nothing,
 *     free nothing and — on a caller's stream — never wait for the device (reference: assert_no_alloc, src/output/cpal.rs:712-715;
 *     checked with pg_debug_hip_calls). The graph-changing calls first wait for the work of earlier writes (also on the caller's
 *     stream the last write used): none of them may run between two asynchronous writes without that wait.
 */
#ifndef PHONIC_GPU_H
#define PHONIC_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error taxonomy: reference `Error` enum (src/error.rs:8-22) ------------------------- */
typedef enum pg_status {
  PG_OK = 0,
  PG_ERR_PARAMETER = 1,   /* Error::ParameterError: unknown FourCC, non-stereo I/O, bad value */
  PG_ERR_NOT_FOUND = 2,   /* Error::EffectNotFoundError / MixerNotFoundError / unknown voice  */
  PG_ERR_QUEUE_FULL = 3,  /* Error::SendError: message queue of a mixer is full               */
  PG_ERR_DEVICE = 4,      /* HIP failure; the handle becomes silent (GuardedSource semantics) */
  PG_ERR_STATE = 5        /* call order violated (e.g. process before initialize)            */
} pg_status;

const char* pg_last_error_message(void);

/* Number of HIP devices visible, or a negative pg_status. */
int pg_device_count(void);

/* ---- effect kinds: the ten stock effects (src/effect/ *.rs) ------------------------------ */
typedef enum pg_effect_kind {
  PG_FX_GAIN = 0,        /* src/effect/gain.rs        "Gain"       */
  PG_FX_PANNING = 1,     /* src/effect/pan.rs         "Panning"    */
  PG_FX_FILTER = 2,      /* src/effect/filter.rs      "Filter"     */
  PG_FX_EQ5 = 3,         /* src/effect/eq5.rs         "Eq5"        */
  PG_FX_DELAY = 4,       /* src/effect/delay.rs       "Delay"      */
  PG_FX_REVERB = 5,      /* src/effect/reverb.rs      "Reverb"     */
  PG_FX_CHORUS = 6,      /* src/effect/chorus.rs      "Chorus"     */
  PG_FX_COMPRESSOR = 7,  /* src/effect/compressor.rs  "Compressor" */
  PG_FX_GATE = 8,        /* src/effect/gate.rs        "Gate"       */
  PG_FX_DISTORTION = 9,  /* src/effect/distortion.rs  "Distortion" */
  PG_FX_KIND_COUNT = 10
} pg_effect_kind;

/* FourCC of a parameter id, e.g. PG_FOURCC('r','o','o','m') == FourCC(*b"room"). */
#define PG_FOURCC(a, b, c, d) \
  ((uint32_t)(uint8_t)(a) << 24 | (uint32_t)(uint8_t)(b) << 16 | (uint32_t)(uint8_t)(c) << 8 | (uint32_t)(uint8_t)(d))

#define PG_MAX_INIT_PARAMS 16

/*
 * Construction-time values of an effect: the `with_parameters(..)` constructors of the
 * reference (e.g. ReverbEffect::with_parameters, src/effect/reverb.rs:154-159). Values are
 * raw (not normalized); enum parameters take the variant index, booleans 0/1. Parameters
 * not listed keep the reference default. n_params == 0 is `Effect::new()`.
 *
 * The reverb draws fpd_l/fpd_r and the 16 vibrato phases from rand::rng()
 * (src/effect/reverb.rs:95-103,532-538); here they are explicit so results are reproducible.
 * vib_phase index = line * 2 + channel, lines in order a..h.
 */
typedef struct pg_effect_init {
  uint32_t n_params;
  uint32_t fourcc[PG_MAX_INIT_PARAMS];
  float value[PG_MAX_INIT_PARAMS];
  uint32_t has_reverb_seeds;
  uint32_t reverb_fpd_l;
  uint32_t reverb_fpd_r;
  double reverb_vib_phase[16];
  /* The Delay's LFO shapes Random and Smooth Random (LfoWaveform, src/utils/dsp/lfo.rs:35-46) draw from `SmallRng::from_os_rng()`
   * (lfo.rs:73): the reference is not reproducible there. Here the generator's state is an explicit input, like the reverb's seeds:
   * rand ^0.9's SmallRng on 64-bit targets is Xoshiro256++ (Blackman / Vigna; state = four u64), `random::<f32>()` = (next_u64() >> 40) *
   * 2^-24 (StandardUniform: the upper 24 bits of next_u32 = the upper 32 of next_u64). `lfo_rng_state` is that state as
   * Effect::initialize finds it (Lfo::new then draws sample_hold, jitter_current, jitter_target from it, lfo.rs:74-76). Without a
   * seed (or with an all-zero state, which xoshiro cannot hold) the state is SplitMix64(0x5EED0000) x 4 — every unseeded Delay then
   * runs the same random sequence. */
  uint32_t has_lfo_seed;
  uint32_t reserved_lfo;
  uint64_t lfo_rng_state[4];
} pg_effect_init;

/* ---- parameter descriptors: `Effect::parameters()` (src/effect.rs:105-111) -------------- */
typedef enum pg_param_type { PG_PARAM_FLOAT = 0, PG_PARAM_ENUM = 1, PG_PARAM_BOOL = 2 } pg_param_type;
typedef enum pg_param_scaling {
  PG_SCALE_LINEAR = 0,
  PG_SCALE_EXPONENTIAL = 1, /* src/parameter/scaling.rs:21  arg0 = factor          */
  PG_SCALE_DECIBEL = 2      /* src/parameter/scaling.rs:31  arg0/1 = min_db/max_db */
} pg_param_scaling;

typedef struct pg_param_desc {
  uint32_t fourcc;
  int32_t type;          /* pg_param_type */
  float min, max;        /* float range; for enums 0 .. n_values-1 */
  float default_value;   /* raw default */
  int32_t scaling;       /* pg_param_scaling */
  float scaling_arg0, scaling_arg1;
  int32_t n_values;      /* enum variant count, else 0 */
  const char* name;
} pg_param_desc;

const char* pg_effect_kind_name(int kind);                   /* Effect::name()   src/effect.rs:96 */
int pg_effect_kind_weight(int kind);                         /* Effect::weight() src/effect.rs:101 */
int pg_effect_kind_param_count(int kind);
int pg_effect_kind_param(int kind, int index, pg_param_desc* out);

/* ---- standalone effect: the `Effect` trait (src/effect.rs:86-215) ----------------------- */
typedef struct pg_effect pg_effect;

/* Effect::new()/with_parameters(); `device` = HIP device ordinal. NULL on failure. */
pg_effect* pg_effect_create(int kind, const pg_effect_init* init, int device);
/* Effect::initialize(sample_rate, channel_count, max_frames)  src/effect.rs:113-125 */
int pg_effect_initialize(pg_effect* fx, uint32_t sample_rate, size_t channel_count, size_t max_frames);
/* Effect::process_started / process_stopped  src/effect.rs:127-139 */
int pg_effect_process_started(pg_effect* fx);
int pg_effect_process_stopped(pg_effect* fx);
/* Effect::process(&mut output, time): in place, host buffer, n_samples <= max_frames*channels
 * and a multiple of the channel count (src/effect.rs:141-155). */
int pg_effect_process(pg_effect* fx, float* interleaved, size_t n_samples, uint64_t pos_in_frames);
/* Effect::process_tail(): -1 = None, INT64_MAX = Some(usize::MAX)  src/effect.rs:157-176 */
int64_t pg_effect_tail(pg_effect* fx);
/* Effect::process_parameter_update(id, Raw|Normalized)  src/effect.rs:178-195 */
int pg_effect_set_parameter(pg_effect* fx, uint32_t fourcc, float value, int is_normalized);
/* Effect::process_message(Reset) (ReverbEffectMessage::Reset etc., src/effect/reverb.rs:24-27) */
int pg_effect_message_reset(pg_effect* fx);
void pg_effect_destroy(pg_effect* fx);
/* Test hook (index streams are compared separately from sample values). With out == NULL the effect (re)arms a log of
 * `words` slots, all -1; the following process calls record the floor()-derived ring index of every delay-line read their time-parallel
 * path takes — Reverb: slot ((frame * 8 + line) * 2 + channel) = read_1 of ReverbDelayLine::get (src/effect/reverb.rs:563-570); Delay and
 * Chorus: slot (frame * 2 + channel) = read_idx1 of InterpolatedDelayLine::process (src/utils/dsp/delay.rs:120-133); frame counts from the
 * start of each process call. With out != NULL the log is copied out. */
int pg_effect_debug_index_log(pg_effect* fx, int32_t* out, size_t words);

/* ---- batched mixer graph: `MixedSource` as a `Source` (src/source/mixed.rs, src/source.rs:80-110)
 *
 * Mixer 0 is the main mixer. Sub-mixers (Player::add_mixer, src/player.rs:773-822) hang off
 * the main mixer; each owns its sources and its effect chain and is one GPU workgroup.
 */
typedef struct pg_graph pg_graph;

#define PG_MAIN_MIXER 0
#define PG_REPEAT_FOREVER UINT64_MAX

/* FilePlaybackOptions (src/source/file.rs:34-75) for a preloaded source. */
typedef struct pg_voice_options {
  float volume;            /* default 1.0 */
  float panning;           /* default 0.0, -1..1 */
  double speed;            /* default 1.0 */
  uint64_t repeat;         /* 0 = play once, PG_REPEAT_FOREVER; (Option<usize>: has_repeat) */
  uint32_t has_repeat;     /* 0 = None: forever iff a loop range is set (preloaded.rs:89-95) */
  uint32_t has_loop_range; /* loop_range override in source frames (preloaded.rs:101-104)   */
  uint64_t loop_start, loop_end;
  uint64_t start_time;     /* sample time in output frames at which the source starts       */
  float fade_in_seconds;   /* < 0 or 0 = none                                               */
  float fade_out_seconds;  /* default 0.05 (file.rs:106); < 0 = none                        */
  uint32_t source_rate;    /* output rate the file source itself is created with (PreloadedFileSource::from_shared_buffer(.., sample_rate),
                              preloaded.rs:71-117); 0 = the graph's rate (what Player passes). When it differs, ConvertedSource puts a cubic
                              ResampledSource — 512-frame input / output staging, src/source/resampled.rs:44-152 — between the file source
                              and the channel mapping (src/source/converted.rs:15-45), exactly as for any source whose rate is not the mixer's */
  uint32_t non_transient;  /* 0 (default) = PlayingSource::is_transient (src/source/mixed.rs:34-42,117-123): the mixer drops the source when it is exhausted
                              (mixed.rs:612-616,715) and RemoveAllPendingEvents takes it when it has not started yet (:298-305). 1 = a source the
                              mixer keeps: exhausted, it stays in the list (asked once per chunk, delivering nothing), stop_all_voices does not take it,
                              write never returns 0 for want of sources — until pg_graph_remove_voice. (In the reference: generators; here: a host-fed
                              source the host wants to keep across pauses of its stream.) */
} pg_voice_options;

void pg_voice_options_default(pg_voice_options* opt);

/* MixedSource::new(channel_count, sample_rate) for the main mixer (src/source/mixed.rs:222-264).
 * max_frames (1..=4096) is the kernels' PIECE size — how many frames a workgroup holds in LDS at a time (the staged kernels of reverb-terminated
 * chains take pieces of <= 1024 frames) — and nothing else: a write of any length is walked in the reference's chunks, min(remaining, 4096)
 * frames (MAX_MIX_BUFFER_SAMPLES / 2, src/source/mixed.rs:216) from the call's start and from every main-mixer event (mixed.rs:679-712),
 * whatever max_frames is; a chunk is rendered as pieces of max_frames frames, and everything the reference decides once per chunk — the effect
 * processors' bypass and tail counters (src/source/mixed/effect.rs:56-145), the sub-mixers' silence gate (submixer.rs:47-77), `audible_input`,
 * an effect's own call-end bookkeeping, a source's fader arrival / end-of-file / is_exhausted — is decided once per chunk here too.
 * 1024 is the size the kernels are tuned for; a host with 2048- or 4096-frame callbacks keeps it and gets the same chunk grid as the reference. */
pg_graph* pg_graph_create(uint32_t sample_rate, uint32_t channel_count, size_t max_frames, int device);
void pg_graph_destroy(pg_graph* g);

/* Player::add_mixer(parent = main) -> mixer id > 0 (src/player.rs:773-822). */
int pg_graph_add_mixer(pg_graph* g);
/* Player::add_mixer(parent_mixer_id) (src/player.rs:771-822): the new mixer is a child of `parent_mixer_id` (0 = main mixer); the
 * parent sums its sub-mixers first, then its sources, then runs its effects (MixedSource::write, src/source/mixed.rs:696-703;
 * SubMixerProcessor::process, src/source/mixed/submixer.rs:47-77). PG_ERR_NOT_FOUND for an unknown parent. */
int pg_graph_add_mixer_to(pg_graph* g, int parent_mixer_id);
/* Player::add_effect(effect, mixer) -> effect id >= 0 (src/player.rs:893-939). */
int pg_graph_add_effect(pg_graph* g, int mixer_id, int kind, const pg_effect_init* init);
/* Player::remove_mixer(mixer_id) (src/player.rs:825-867 -> MixerMessage::RemoveMixer to the parent, src/source/mixed.rs:422-424): the
 * sub-mixer leaves its parent at the start of the next write, with its effects, sources and nested sub-mixers (their ids return
 * PG_ERR_NOT_FOUND afterwards). PG_ERR_PARAMETER for the main mixer. */
int pg_graph_remove_mixer(pg_graph* g, int mixer_id);
/* Player::remove_effect(effect_id) (src/player.rs:977-990 -> MixerMessage::RemoveEffect, src/source/mixed.rs:433-440): takes effect at
 * the start of the next write; later calls with this id return PG_ERR_NOT_FOUND. */
int pg_graph_remove_effect(pg_graph* g, int effect_id);
/* Player::move_effect(movement, effect_id, mixer_id) (src/player.rs:942-972 -> MixerMessage::MoveEffect, src/source/mixed.rs:441-462):
 * EffectMovement::Direction(offset) / Start / End (src/player.rs:75-82). PG_ERR_PARAMETER when the effect is not in `mixer_id`. */
#define PG_MOVE_DIRECTION 0
#define PG_MOVE_START 1
#define PG_MOVE_END 2
int pg_graph_move_effect(pg_graph* g, int effect_id, int mixer_id, int movement, int offset);
/* Player::play_file_source(PreloadedFileSource::from_shared_buffer(..), start_time)
 * (src/player.rs:519-602, src/source/file/preloaded.rs:71-117). `pcm` is the decoded interleaved
 * buffer INCLUDING the extra zero frame symphonia decoding appends (file/buffer.rs:103-104);
 * it is copied to the device. Returns a voice (playback) id >= 0. */
int pg_graph_add_voice(pg_graph* g, int mixer_id, const float* pcm, size_t n_frames, uint32_t src_channels,
                       uint32_t src_rate, const pg_voice_options* opt);

/* Player::play_synth_source / any `dyn Source` (MixerMessage::AddSource{source: Box<dyn Source>}, src/source/mixed.rs:117-123;
 * SynthSourceImpl, src/source/synth/common.rs:194-263; streamed files): a source whose samples the HOST produces. The voice is a ring
 * of `capacity_frames` frames (>= 1024) in device memory at the source's own `rate` and channel count (1 or 2); the host keeps it filled
 * with what it pulls from its source (pg_graph_feed_voice), the device reads it where a file voice reads its preloaded buffer, behind
 * the same adapter chain: ResampledSource when `rate` is not the mixer's (src/source/converted.rs:15-45, 512-frame staging), mono ->
 * stereo, volume, panning, start time (`opt`: volume, panning, start_time are used). A short ring read is a source that delivered
 * less (the rest of the block is silent); the voice ends when the host has ended the stream and everything fed has been played, or
 * at a stop. Device ring and pinned staging ring are reserved here: feed and write allocate nothing. Returns a voice id >= 0 (valid
 * for set_voice_volume / _panning / stop_voice / remove_voice like any other; set_voice_speed and seek_voice return PG_ERR_PARAMETER: those
 * exist on FilePlaybackHandle only, src/player/handles/file.rs). In steady state such a voice is rendered by the same time-parallel kernels
 * as a file voice (the ring read is a copy), also inside a write of several blocks. */
int pg_graph_add_stream_voice(pg_graph* g, int mixer_id, uint32_t channels, uint32_t rate, size_t capacity_frames, const pg_voice_options* opt);
/* The next n_frames frames of the host's source (interleaved). Owner thread, before the write that should play them; they reach the
 * device on that write's stream. PG_ERR_QUEUE_FULL (nothing taken) when fed - consumed + n_frames would exceed the capacity, with
 * `consumed` as last reported by pg_graph_stream_voice_consumed. */
int pg_graph_feed_voice(pg_graph* g, int voice_id, const float* frames, size_t n_frames);
/* Source::is_exhausted (src/source.rs:88-93): nothing more will be fed. */
int pg_graph_end_stream_voice(pg_graph* g, int voice_id);
/* Frames of the stream the device has read so far (waits for the graph's work; negative on failure): frees that much ring for feeds. */
int64_t pg_graph_stream_voice_consumed(pg_graph* g, int voice_id);

/* MixerMessage::RemoveSource (src/source/mixed.rs:149-151,400-402): the source leaves its mixer at the start of the next write, at once — no
 * fade-out (stop_voice is the call that fades) — whether transient or not, started or not; events already scheduled for it find no source and
 * are dropped when they come due. Any thread. Later calls with this id return PG_ERR_NOT_FOUND. */
int pg_graph_remove_voice(pg_graph* g, int voice_id);

/* EffectHandle::set_parameter((id, update), sample_time) (src/player/handles/effect.rs:67-95) */
int pg_graph_schedule_param(pg_graph* g, int effect_id, uint32_t fourcc, float value, int is_normalized,
                            uint64_t sample_time);
/* EffectHandle::send_message(Reset, sample_time) */
int pg_graph_schedule_reset(pg_graph* g, int effect_id, uint64_t sample_time);
/* FilePlaybackHandle::set_volume / set_panning / stop (src/player/handles/file.rs) */
int pg_graph_set_voice_volume(pg_graph* g, int voice_id, float volume, uint64_t sample_time);
int pg_graph_set_voice_panning(pg_graph* g, int voice_id, float panning, uint64_t sample_time);
int pg_graph_stop_voice(pg_graph* g, int voice_id, uint64_t sample_time);
/* Player::stop_all_sources() (src/player.rs:1012-1045): stops every playing source from the next write on (with its fade-out) and sends
 * MixerMessage::RemoveAllPendingEvents to every mixer (src/source/mixed.rs:298-305): sources that have not started by then and events
 * scheduled after that write's position are dropped. */
int pg_graph_stop_all_voices(pg_graph* g);
/* FilePlaybackHandle::set_speed(speed, glide) / seek(position) (src/player/handles/file.rs -> MixerMessage::SetSourceSpeed /
 * SeekSource, src/source/mixed.rs:338-383; PreloadedFileSource::set_speed / seek, src/source/file/preloaded.rs:139-192).
 * glide_semitones_per_second <= 0 = no glide (Option<f32>::None). */
int pg_graph_set_voice_speed(pg_graph* g, int voice_id, double speed, float glide_semitones_per_second, uint64_t sample_time);
int pg_graph_seek_voice(pg_graph* g, int voice_id, double position_seconds, uint64_t sample_time);

/* Source::write(&mut output, &SourceTime{pos_in_frames}) of the main MixedSource
 * (src/source/mixed.rs:659-719): returns the samples written == n_samples, or 0 when the
 * graph is empty (or after a device failure: GuardedSource, src/source/guarded.rs:87-107).
 * `out` is a host buffer. */
size_t pg_graph_write(pg_graph* g, float* out, size_t n_samples, uint64_t pos_in_frames);
/* Same, output left in device memory (`d_out` = device pointer, >= n_samples floats), enqueued
 * on `hip_stream` (a hipStream_t, NULL = the graph's own stream); asynchronous when a stream is
 * given. Used for the multi-GPU master-bus reduce and by bench.py. */
size_t pg_graph_write_device(pg_graph* g, float* d_out, size_t n_samples, uint64_t pos_in_frames, void* hip_stream);
/* Offline rendering (the reference's WavOutput pull loop, src/output/wav.rs:210-250, has no deadline per block): a write*() call that
 * spans several blocks of max_frames may render up to `n_blocks` of them in ONE launch sequence when nothing is scheduled inside them and
 * every unit is in steady state (MixedSource::write walks its chunks inside one call the same way, src/source/mixed.rs:679-712). All
 * per-chunk semantics (bypass counters, tails, silence gates) stay per chunk of the reference's grid (see pg_graph_create): a launch sequence
 * covers whole chunks, its blocks are their pieces. Default 1; sizes the per-unit output table (max(n_blocks, 4096 / max_frames) x units x
 * max_frames x 8 bytes), allocated at the next graph mutation / first write — call it while building the graph. The result is bit-identical
 * to the same calls without super-block launches.
 * (pg_graph_write with a host buffer renders ONE write call whatever its length: its staging holds whole chunks — at least 4096 frames —
 * and the call is copied out span by span.) */
/* A unit that leaves the steady state while a super-block launch is in flight (the host only launches them for graphs it knows to be
 * steady: this is a consistency violation, PG_DEVERR_SUPER_DEFERRED) would miss its later blocks: the kernels mirror that flag to the host
 * and the NEXT write*() call disables the graph (returns 0 from then on, pg_last_error_message names the flag) — wrong audio is never
 * handed out twice; pg_graph_device_errors reports the flag at any time. */
int pg_graph_set_max_blocks_per_launch(pg_graph* g, int n_blocks);
/* Process-wide counters of the library's own HIP calls: out[0] = allocations (hipMalloc / hipHostMalloc), out[1] = releases, out[2] = host
 * waits for a stream (hipStreamSynchronize), out[3] = blocking copies / fills. The reference runs its audio callback under
 * assert_no_alloc (src/output/cpal.rs:712-715); with these counters a test asserts the same of pg_graph_write*: on a built graph it
 * allocates nothing and frees nothing, and on a caller's stream it never blocks the host. */
void pg_debug_hip_calls(uint64_t out[4]);
/* Test hook: the nth launch round from now (process-wide, any graph) fails the way a HIP launch failure does — the graph it hits becomes
 * silent for good: write returns 0, like the reference's GuardedSource after a panic (src/source/guarded.rs:87-107). 0 disarms. Armed only in
 * a process with PHONIC_DEBUG_HOOKS=1 in its environment; a no-op otherwise. */
void pg_debug_fail_launch_round(int nth);
/* Build check hook (tools/check_kernel_resources.py): dynamic LDS bytes of the staged single launch (which 0) / of a fast unit kernel for the
 * effect kinds of kind_mask (which 1) at n_frames frames per block. */
size_t pg_debug_lds_bytes(int which, uint32_t n_frames, uint32_t kind_mask);
/* Sticky consistency flags raised by the kernels (0 = none; see PG_DEVERR_* in phonic_amd/csrc/pg_dev.h): conditions the host-side
 * routing of units to kernel variants must make impossible. Synchronises the graph's own stream. Negative pg_status on failure. */
int pg_graph_device_errors(pg_graph* g);
/* Multi-GPU: when set, bus effects are skipped in write*(): the caller reduces the partial bus of
 * all ranks (RCCL) and then runs them once on the root with pg_graph_process_bus_device(). */
int pg_graph_set_defer_bus(pg_graph* g, int defer);
int pg_graph_process_bus_device(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, void* hip_stream);
/* One process per GPU: the ranks' `audible` words ride with their partial buses. pg_graph_export_audible writes the words of the LAST
 * pg_graph_write_device call of a deferred-bus graph — ONE PER PIECE the call was rendered in, in order: a chunk of the reference's grid
 * (min(remaining, 4096) frames from the call's start and from every main-mixer event, src/source/mixed.rs:216,679-712) is ceil(chunk / max_frames)
 * pieces, a chunk's flag sits in the word of its last piece; all 0 when that call had nothing to render — as floats (0 / 1) to d_dst, e.g. right
 * behind the call's samples: ONE sum-reduce then carries samples and words (OR = sum > 0). pg_graph_audible_words = how many words the last
 * call left (an event-free call of n frames: one per block of max_frames when max_frames divides 4096); a deferred-bus call leaves at most
 * max(64, 4096 / max_frames) words — a call that would need more is rendered up to there and returns the samples it rendered (the caller goes on
 * with another call). pg_graph_process_bus_device_flags is pg_graph_process_bus_device with those summed words, consumed in the same order: the
 * root's bus chain takes EffectProcessor's decisions (bypass, tails; src/source/mixed/effect.rs:56-145) per chunk as the one main mixer would.
 * The chain walks the SAME chunk grid as the write: it restarts at the main mixer's effect events (queued by the write) and at the offsets where
 * any other main-mixer event cut this graph's own deferred write of the same position. Ranks whose main-mixer SOURCES take events the root's
 * graph does not hold must end their calls there on every rank: pg_graph_next_main_event (drains the control ring — writing thread only —
 * and returns the sample time of the first main-mixer event behind pos_in_frames, UINT64_MAX when there is none) is what a caller min-reduces
 * over the ranks to size the next call (phonic_amd/parallel.py: next_call_frames). */
int pg_graph_export_audible(pg_graph* g, float* d_dst, int n_words, void* hip_stream);
int pg_graph_audible_words(pg_graph* g);
uint64_t pg_graph_next_main_event(pg_graph* g, uint64_t pos_in_frames);
int pg_graph_process_bus_device_flags(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, void* hip_stream, const float* d_flags, int n_words);
/* Block until all work of the graph's stream has finished. */
int pg_graph_synchronize(pg_graph* g);

/* ---- voice-sharded graph: the same main MixedSource spread over several GPUs of one node (SURVEY.md §8b `n_gpus`, §8e) ----------
 *
 * The reference's parallel axis is independent sub-mixers rendered by worker threads into private buffers which the caller sums
 * (SubMixerThreadPool, src/source/mixed/submixer/thread_pool.rs:92-121,350-412; src/source/mixed.rs:522-536). Here the workers are
 * devices: one handle owns one pg_graph per entry of `devices` (the first is the root). Every sub-mixer of the main mixer — with its
 * effects, sources and nested sub-mixers — and every main-mixer source is placed on the least loaded shard when it is added (the greedy
 * placement of WorkerTaskBatcher) and never moves; effects added to mixer 0 form the bus chain on the root. write = one asynchronous
 * render per shard on its own device and stream, the partial buses summed on the root device, then the bus chain.
 *
 * The handle IS the main MixedSource: it takes every call pg_graph takes (the reference's MixerMessage set, src/source/mixed.rs:124-145,
 * 163-178,422-462) and routes it to the shard that owns the target; ids returned here are global (valid for the pg_sharded_* calls
 * only) and never reused. A write is cut at the main mixer's event times of ALL shards, so every shard splits its chunks where the one
 * mixer would (mixed.rs:679-712); the bus chain gets one `audible_input` per chunk, OR-ed over the shards (mixed.rs:696-706); write
 * returns 0 exactly when pg_graph_write would (mixed.rs:664-670). Threading as for pg_graph: add_* / remove_* / move_* / write from the
 * owner thread, the control calls (schedule_*, set_voice_*, seek, stop_*) from any thread. A device may be listed more than once
 * (several shards on one GPU: how the single-GPU test-suite exercises this path; not with PG_REDUCE_RCCL). bench.py's measured multi-GPU
 * path is one process per GPU with an RCCL reduce through torch.distributed (phonic_amd/parallel.py); both sit on the same kernels and
 * per-graph host code. */
typedef struct pg_sharded_graph pg_sharded_graph;
pg_sharded_graph* pg_sharded_create(uint32_t sample_rate, uint32_t channel_count, size_t max_frames, const int* devices, int n_devices);
void pg_sharded_destroy(pg_sharded_graph* s);
int pg_sharded_shard_count(pg_sharded_graph* s);
int pg_sharded_set_max_blocks_per_launch(pg_sharded_graph* s, int n_blocks);   /* a write holds at most n_blocks x max_frames frames */
/* How the shards' partial buses meet on the root device.
 *   PG_REDUCE_PEER_COPY (default): hipMemcpyPeerAsync of every partial to the root + one sum kernel, f32 adds in shard order (deterministic).
 *   PG_REDUCE_RCCL: ncclReduce(sum, float32, root = shard 0) over xGMI on the shards' own streams inside one ncclGroupStart/End, the
 *     `audible` words by ncclReduce(max) in the same group (north_star: "RCCL reduce over xGMI for the master-bus sum"). The sum order
 *     is RCCL's, inside the 1e-5 RMS gate. One communicator per shard from ncclCommInitAll over `devices`, created by this call: every
 *     device may be listed once only; RCCL is looked up at run time (librccl.so.1 — the one already in the process, e.g. PyTorch's, else
 *     ROCm's). On failure the call returns PG_ERR_DEVICE / PG_ERR_PARAMETER with RCCL's error text and the mode stays as it was. */
#define PG_REDUCE_PEER_COPY 0
#define PG_REDUCE_RCCL 1
int pg_sharded_set_reduce(pg_sharded_graph* s, int mode);
int pg_sharded_reduce_mode(pg_sharded_graph* s);
int pg_sharded_add_mixer(pg_sharded_graph* s);                                  /* Player::add_mixer(None) */
int pg_sharded_add_mixer_to(pg_sharded_graph* s, int parent_mixer_id);          /* nested: lives on its parent's shard */
int pg_sharded_add_effect(pg_sharded_graph* s, int mixer_id, int kind, const pg_effect_init* init);
int pg_sharded_add_voice(pg_sharded_graph* s, int mixer_id, const float* pcm, size_t n_frames, uint32_t src_channels, uint32_t src_rate,
                         const pg_voice_options* opt);
/* host-fed sources on the sharded mixer: pg_graph_add_stream_voice / feed_voice / end_stream_voice / stream_voice_consumed on the owning shard */
int pg_sharded_add_stream_voice(pg_sharded_graph* s, int mixer_id, uint32_t channels, uint32_t rate, size_t capacity_frames, const pg_voice_options* opt);
int pg_sharded_feed_voice(pg_sharded_graph* s, int voice_id, const float* frames, size_t n_frames);
int pg_sharded_end_stream_voice(pg_sharded_graph* s, int voice_id);
int64_t pg_sharded_stream_voice_consumed(pg_sharded_graph* s, int voice_id);
int pg_sharded_shard_of_mixer(pg_sharded_graph* s, int mixer_id);
/* Player::remove_mixer / remove_effect / move_effect (src/player.rs:825-867,942-990; MixerMessage::RemoveMixer / RemoveEffect / MoveEffect,
 * src/source/mixed.rs:422-462): semantics and errors of pg_graph_remove_mixer / _remove_effect / _move_effect. */
int pg_sharded_remove_mixer(pg_sharded_graph* s, int mixer_id);
int pg_sharded_remove_effect(pg_sharded_graph* s, int effect_id);
int pg_sharded_move_effect(pg_sharded_graph* s, int effect_id, int mixer_id, int movement, int offset);
int pg_sharded_schedule_param(pg_sharded_graph* s, int effect_id, uint32_t fourcc, float value, int is_normalized, uint64_t sample_time);
int pg_sharded_schedule_reset(pg_sharded_graph* s, int effect_id, uint64_t sample_time);
int pg_sharded_set_voice_volume(pg_sharded_graph* s, int voice_id, float volume, uint64_t sample_time);
int pg_sharded_set_voice_panning(pg_sharded_graph* s, int voice_id, float panning, uint64_t sample_time);
/* FilePlaybackHandle::set_speed / seek (src/player/handles/file.rs:111,150 -> MixerMessage::SetSourceSpeed / SeekSource, mixed.rs:338-383) */
int pg_sharded_set_voice_speed(pg_sharded_graph* s, int voice_id, double speed, float glide_semitones_per_second, uint64_t sample_time);
int pg_sharded_seek_voice(pg_sharded_graph* s, int voice_id, double position_seconds, uint64_t sample_time);
int pg_sharded_stop_voice(pg_sharded_graph* s, int voice_id, uint64_t sample_time);
int pg_sharded_stop_all_voices(pg_sharded_graph* s);
int pg_sharded_remove_voice(pg_sharded_graph* s, int voice_id);
int pg_sharded_is_voice_playing(pg_sharded_graph* s, int voice_id);
/* Source::write: host buffer (waits for the result) / buffer on the root device (asynchronous on the shards' streams, several calls may be
 * enqueued before pg_sharded_synchronize). A call of ANY length is ONE write of the one main mixer — messages processed once on every shard,
 * one call end — walked on the reference's chunk grid (min(remaining, 4096) frames from the call's start and from every main-mixer event of
 * any shard, src/source/mixed.rs:216,679-712) whatever max_blocks x max_frames is. Both return the samples written, or 0 when
 * the main mixer has nothing to do (mixed.rs:664-670: no playing source, sub-mixer or pending event on any shard and no effect on
 * mixer 0; the host learns that sources have ended from the device after a pg_sharded_write, as pg_graph_write does). */
size_t pg_sharded_write(pg_sharded_graph* s, float* out, size_t n_samples, uint64_t pos_in_frames);
size_t pg_sharded_write_device(pg_sharded_graph* s, float* d_out, size_t n_samples, uint64_t pos_in_frames);
int pg_sharded_synchronize(pg_sharded_graph* s);
int pg_sharded_device_errors(pg_sharded_graph* s);

/* Introspection used by the harness */
int pg_graph_voice_count(pg_graph* g);
int pg_graph_is_voice_playing(pg_graph* g, int voice_id);
/* Number of units (sub-mixers and main-mixer sources) the time-parallel kernels handed to the exact serial kernel in the last launch
 * round that had any to hand over or to check (ramping parameters, a command inside the block, a chain without a time-parallel
 * path); 0 in steady state. Synchronises the graph's own stream: call it after pg_graph_write, or after the caller synchronised the
 * stream given to pg_graph_write_device. */
int pg_graph_deferred_units(pg_graph* g);
/* Average device time (ms) of the dominant kernel launch(es) (see pg_graph_dominant_kernel) over the launches
 * since the last call with reset != 0, measured with hipEvents on the graph's stream; launches = count. */
double pg_graph_kernel_ms(pg_graph* g, int reset, uint64_t* launches);
/* The same measurement as sums: total device time (ms) of the timed launches, their count, and the number of max_frames blocks they
 * rendered (a super-block launch renders several blocks per unit). Waits for the timed launches (also on a caller's stream). */
int pg_graph_kernel_stats(pg_graph* g, int reset, double* total_ms, uint64_t* launches, uint64_t* blocks);
/* The same for the launches of the main mixer's effect chain (one workgroup per effect behind the sum: a latency chain — for graphs whose
 * work is mostly on the bus, BASELINE configs 2 and 4, it is the launch that dominates by GPU time), and its name. */
int pg_graph_bus_kernel_stats(pg_graph* g, int reset, double* total_ms, uint64_t* launches, uint64_t* blocks);
/* What a workload off the steady state costs (events, voices that start and end, ramps): out[0] = unit-blocks rendered since the last reset
 * (units x blocks of max_frames), out[1] = unit-blocks that left the time-parallel kernels for the generic kernel, out[2] = generic launches
 * issued, out[3] = those that found work; *generic_ms / *generic_timed = GPU time and count of the generic launches that were hipEvent-timed
 * (pg_graph_set_timing_period). Waits for the graph's stream. Measurement only (bench.py --workload dyn). */
int pg_graph_dynamic_stats(pg_graph* g, int reset, uint64_t out[4], double* generic_ms, uint64_t* generic_timed);
const char* pg_graph_bus_kernel(pg_graph* g);
/* The hipEvent pair behind pg_graph_kernel_ms costs ~8 us of stream time per round: time every n-th round only (default 1 = every
 * round, 0 = never). pg_graph_kernel_ms then averages over the timed rounds and reports their count. */
int pg_graph_set_timing_period(pg_graph* g, int every_n_rounds);
/* Name(s) of the kernel launch(es) the pg_graph_kernel_ms events bracket for this graph (static string). */
const char* pg_graph_dominant_kernel(pg_graph* g);
/* 0 = exact serial filters, 1 = time-parallel (blocked) evaluation of linear filters (default) */
int pg_graph_set_fast_math(pg_graph* g, int level);
/* How sub-mixers whose chain ends in a Reverb behind Gain / Panning (and, in mode 1, Filter / Eq5 / Delay / Distortion) effects
 * are rendered: 1 (default) = staged kernel (stage functions over a per-stage LDS plan in one launch, four workgroups per CU),
 * 2 = one launch per stage (profiling; Gain / Panning chains only), 0 = the fused fast kernel. Same stage functions in every mode: results agree up to f64 rounding. */
int pg_graph_set_staged(pg_graph* g, int mode);

#ifdef __cplusplus
}
#endif
#endif /* PHONIC_GPU_H */
