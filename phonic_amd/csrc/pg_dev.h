// Device-resident state of the phonic DSP hot path (HBM layout). Plain-old-data, shared by the host
// side (graph construction, C ABI) and the gfx950 kernels. All delay lines keep the reference's f64
// ring layout and index arithmetic (reference src/utils/dsp/delay.rs) so state is comparable with the
// CPU oracle after every block.
#pragma once
#include <stdint.h>

#define PG_USIZE_MAX 0xFFFFFFFFFFFFFFFFull
#define PG_MAX_FRAMES 4096  // MixedSource::MAX_MIX_BUFFER_SAMPLES / 2 (src/source/mixed.rs:216)
// Semantic chunks and kernel pieces. MixedSource::write walks a call in chunks of min(remaining, PG_MAX_FRAMES) frames from the call's
// start and from every event (mixed.rs:679-712); sources, effect processors and sub-mixers are called once per chunk and take their
// per-call decisions (bypass, tails, silence gates, block-end bookkeeping) there. The kernels render a chunk as PIECES of at most the
// graph's max_frames frames (LDS-resident, <= 1024 for the staged kernels): per-chunk decisions are taken at a chunk's first piece, kept in
// the unit / effect / voice records and closed at its last piece, so the chunk grid is the reference's whatever max_frames is.

enum PgSmoothKind { SM_EXP = 0, SM_LIN = 1, SM_SPRING = 2 };

// ExponentialSmoothedValue / LinearSmoothedValue / SpringSmoothedValue (src/utils/smoothing.rs)
struct PgSmooth {
  int32_t kind;
  float current, target;
  float a;      // inertia | step | omega
  float b;      // -       | current_step | velocity
  float comp;   // 44100 / sample_rate
  uint32_t pending;  // linear: num_pending_steps
};

// BiquadFilterCoefficients (src/utils/dsp/filters/biquad.rs:31-43)
struct PgBiquadCoef {
  int32_t type;
  uint32_t sample_rate;
  float cutoff, q, gain;
  int32_t pad;
  double a1, a2, a3, m0, m1, m2;
};
// SvfFilterCoefficients (src/utils/dsp/filters/svf.rs:30-41)
struct PgSvfCoef {
  int32_t type;
  uint32_t sample_rate;
  float cutoff, resonance;
  double g, k, a1, a2, a3;
};
struct PgState2 { double ic1eq, ic2eq; };      // BiquadFilter / SvfFilter state
struct PgDc { double y1, x1, r; };             // DcFilter (src/utils/dsp/filters/dc.rs:35-39)
struct PgLfo { float phase, phase_inc; int32_t waveform; };  // Lfo (src/utils/dsp/lfo.rs:52-60), deterministic shapes

struct PgGain {  // GainEffect (src/effect/gain.rs:47-55)
  PgSmooth gain;
  int32_t dc_mode;  // 0 Off, 1 Slow, 2 Default, 3 Fast
  PgDc dc[2];
};
struct PgPan {  // PanningEffect (src/effect/pan.rs)
  PgSmooth pan, width;
  int32_t invert_l, invert_r;
};
struct PgFilter {  // FilterEffect (src/effect/filter.rs:47-56)
  PgBiquadCoef coef;
  PgState2 st[2];
  int32_t type;  // FilterEffectType
  PgSmooth cutoff, q;
};
struct PgEq5 {  // Eq5Effect (src/effect/eq5.rs:18-28)
  PgSmooth gains[5], freqs[5], bws[5];
  PgBiquadCoef coef[5];
  PgState2 st[2][5];
};
struct PgDelay {  // DelayEffect (src/effect/delay.rs:88-116)
  int32_t mode, filter_type, lfo_shape;
  PgSmooth delay_time, feedback, cutoff, drive, wet, width, lfo_rate, d_time, d_feedback, d_filter;
  double* line[2];  // InterpolatedDelayLine<1> left/right
  uint32_t mask, write_pos[2];
  PgLfo lfo;
  PgSvfCoef coef;
  PgState2 flt[2];
  PgDc dc[2];
  float fb[2];
  // Lfo's random shapes (lfo.rs:56-59): sample & hold value, the two ends of the smooth-random segment, and the generator —
  // rand ^0.9 SmallRng = Xoshiro256++, its state an explicit input (pg_effect_init::lfo_rng_state)
  float lfo_sample_hold, lfo_jitter_current, lfo_jitter_target, lfo_pad;
  uint64_t lfo_rng[4];
};
struct PgReverbLine {  // ReverbDelayLine<2> (src/effect/reverb.rs:518-529)
  double* buf;         // [(size+1)][2]
  uint32_t frames, count, delay, pad;
  double feedback[2];
  double depth;
  double vib_phase[2];
};
struct PgAllpass {  // AllpassDelayLine<2> (src/utils/dsp/delay.rs:283-287)
  double* buf;      // [size][2]
  uint32_t frames, delay, write_pos, pad;
};
struct PgReverb {  // ReverbEffect (src/effect/reverb.rs:41-73)
  PgSmooth room, wet;
  PgBiquadCoef ca, cb, cc;
  PgState2 sa[2], sb[2], sc[2];
  uint32_t fpd_l, fpd_r;
  PgReverbLine line[8];
  PgAllpass ap[4];
  double* pre;  // DelayLine<2>, pow2 4096 frames
  uint32_t pre_mask, pre_write_pos;
  // block parameters of the steady state (fast path): valid while room/wet targets are unchanged
  float cache_room, cache_wet;
  int32_t cache_valid;
  uint32_t c_predelay;
  double c_blend, c_regen;
  const double* vib_tab;  // [8][129][2]: cos(j*d_i), sin(j*d_i) of the per-frame vibrato increment d_i = depth_i * 0.1 (shared, read-only)
};
struct PgChorus {  // ChorusEffect (src/effect/chorus.rs:47-75)
  PgSmooth rate, phase, depth, feedback, delay, wet, freq, res;
  int32_t filter_type;
  float lfo_range;
  double current_phase;
  PgLfo osc[2];
  double* line[2];
  uint32_t mask, write_pos[2];
  PgSvfCoef coef;
  PgState2 flt[2];
};
struct PgComp {  // CompressorEffect (src/effect/compressor.rs:24-38)
  float threshold, ratio, knee, attack, release, lookahead;
  PgSmooth makeup;
  float env_current, env_attack, env_release;
  double* line;  // LookupDelayLine<2>
  uint32_t line_frames;  // allocated frames (pow2 >= ceil(0.2*sr))
  uint32_t mask, delay_frames, write_pos, peak_pos;
  double peak_value;
};
struct PgGate {  // GateEffect (src/effect/gate.rs)
  float threshold, attack, hold, release, range;
  float env_current, env_attack, env_release;
  uint32_t hold_counter;
  float gate_gain_db, attack_coeff, release_coeff;
};
struct PgDist {  // DistortionEffect (src/effect/distortion.rs:194-204)
  int32_t type;
  PgSmooth drive, mix;
  const float* luts;  // [5][256], shared, built on the host at create
};

struct PgFx {
  int32_t kind;
  uint32_t sample_rate;
  // EffectProcessor (src/source/mixed/effect.rs:12-17)
  int32_t bypassed;
  int32_t standalone;  // 1: plain Effect::process without the processor's bypass logic
  uint64_t tail_counter, silence_counter;
  // the process call (semantic chunk) in progress, across its pieces: frames the effect has rendered so far and — while the processor
  // watches for silence (unknown tail, effect.rs:128-144) — the peak of what it put out
  uint32_t call_frames;
  float call_max;
  // ... and the branch the effect's process() took where the call began: FilterEffect and Eq5Effect test `value_need_ramp` once per call
  // (filter.rs:167, eq5.rs:298-303) and keep recomputing their coefficients from the smoothers for the rest of it — a smoother that stops
  // ramping inside the call then hands out its TARGET, while the other branch runs on the coefficients the last ramp frame left (from the
  // smoother's `current`, up to the ramp threshold away): the branch must not change between the pieces of a call
  uint32_t call_ramp;
  uint32_t pad_call;
  union {
    PgGain gain; PgPan pan; PgFilter filter; PgEq5 eq5; PgDelay delay; PgReverb reverb; PgChorus chorus; PgComp comp; PgGate gate; PgDist dist;
  } u;
};

// PreloadedFileSource + ChannelMapped + Amplified + Panned + PlayingSource (one "voice")
struct PgVoice {
  const float* pcm;        // device copy of AudioFileBuffer (+1 zero frame)
  uint64_t n_samples;      // buffer.len()
  uint32_t channels;       // file channels (1 or 2)
  uint32_t src_rate, out_rate;
  // CubicInterpolator per channel (src/utils/resampler/cubic.rs:10-15)
  float input[2][4];
  float sub_pos[2];
  float ratio;
  int32_t initialized[2];
  // PreloadedFileSource (src/source/file/preloaded.rs:29-37)
  uint64_t playback_pos, repeat, repeat_count;
  int32_t pos_eof, finished;
  int32_t has_loop;
  uint64_t loop_start, loop_end;  // frames
  // VolumeFader (src/utils/fader.rs:27-34)
  int32_t fader_state;  // 0 Stopped 1 Running 2 Finished
  float fader_current, fader_target, fader_inertia;
  float fade_out_seconds;
  // AmplifiedSource / PannedSource smoothers
  PgSmooth volume, panning;
  // PlayingSource (src/source/mixed.rs:34-42)
  uint64_t start_time, stop_time;
  int32_t has_stop, active;
  // speed / glide (FileSourceImpl, src/source/file/common.rs:47-52)
  double current_speed, target_speed;
  float speed_glide_rate;
  uint32_t samples_to_next_speed_update;
  int32_t sched_class, sched_rep;  // resampler schedule cache: class of voices sharing a ratio; 1 = this voice publishes the schedule
  int32_t sched_hit, pad_sched;    // device: how the last resampling piece got its schedule: 0 serial replay, 1 schedule cache, 2 time-parallel
  // ResampledSource around the file source (src/source/resampled.rs:27-98), present when the file source runs at a rate (`out_rate`) other
  // than the mixer's (pg_voice_options::source_rate): cubic interpolator state per channel + the two 512-frame TempBuffers
  int32_t outer_on, outer_pending_stop;
  int32_t outer_init[2];
  float outer_ratio, pad_outer;
  float outer_sub_pos[2];
  float outer_input[2][4];
  uint32_t in_start, in_end, out_start, out_end;  // TempBuffer ranges (samples)
  float* stage_in;   // device memory: 512 * channels floats each
  float* stage_out;
  // Host-fed source (pg_graph_add_stream_voice): `pcm` is a device ring of stream_cap frames that the host fills (pg_graph_feed_voice),
  // playback_pos counts the frames read; stream_fed = frames fed so far, bit 63 = the host has ended the stream
  int32_t stream_on;
  uint32_t stream_cap;
  uint64_t stream_fed;
  // MixedSource::process_sources marks an exhausted transient source inactive but keeps calling it for the rest of the write; it is removed
  // when the write ends (mixed.rs:612-616, 715). Only a ResampledSource makes that audible (asked again, it refills its stale input range and
  // plays on): the end position of the write in which such a voice was marked.
  uint64_t zombie_end;
  // process_sources leaves a source alone for the rest of a chunk once a write returned nothing (`written == 0` -> break 'source,
  // mixed.rs:617-620): set when that happens in a piece, cleared at the first piece of the mixer's next chunk
  int32_t chunk_skip;
  int32_t persistent;  // host: !PlayingSource::is_transient — an exhausted source stays in the mixer's list (mixed.rs:612-620)
};

// Parameter indices per effect kind = order of `Effect::parameters()` in the reference.
enum { P_GAIN_GAIN = 0, P_GAIN_DCFM };
enum { P_PAN_PAN = 0, P_PAN_WIDTH, P_PAN_INVL, P_PAN_INVR };
enum { P_FILTER_TYPE = 0, P_FILTER_CUTOFF, P_FILTER_Q };
// Eq5: index = band * 3 + {0 gain, 1 frequency, 2 bandwidth}
enum { P_DELAY_MODE = 0, P_DELAY_TIME, P_DELAY_FEEDBACK, P_DELAY_FTYPE, P_DELAY_CUTOFF, P_DELAY_DRIVE, P_DELAY_WET, P_DELAY_WIDTH,
       P_DELAY_LFO_RATE, P_DELAY_LFO_SHAPE, P_DELAY_D_TIME, P_DELAY_D_FEEDBACK, P_DELAY_D_FILTER };
enum { P_REVERB_ROOM = 0, P_REVERB_WET };
enum { P_CHORUS_RATE = 0, P_CHORUS_DEPTH, P_CHORUS_FEEDBACK, P_CHORUS_DELAY, P_CHORUS_WET, P_CHORUS_PHASE, P_CHORUS_FTYPE, P_CHORUS_FREQ,
       P_CHORUS_RES };
enum { P_COMP_THRESHOLD = 0, P_COMP_RATIO, P_COMP_KNEE, P_COMP_ATTACK, P_COMP_RELEASE, P_COMP_MAKEUP, P_COMP_LOOKAHEAD };
enum { P_GATE_THRESHOLD = 0, P_GATE_ATTACK, P_GATE_HOLD, P_GATE_RELEASE, P_GATE_RANGE };
enum { P_DIST_TYPE = 0, P_DIST_DRIVE, P_DIST_MIX };

// Resampler schedule cache. The f32 `sub_pos` recurrence of the cubic resampler depends only on (ratio, sub_pos, number of
// output frames) — not on the audio — so voices that run in lock step (same ratio, same state) share one schedule: the
// class representative computes the NEXT block's schedule at the end of its workgroup and every voice whose state
// matches the key bit for bit copies it instead of replaying the serial recurrence (any mismatch -> own replay).
#define PG_SCHED_CAP 1024
struct PgSchedEntry {
  uint32_t ratio_bits, subpos_in_bits;  // key
  int32_t piece, valid;
  uint32_t subpos_out_bits;             // state after the piece
  int32_t c_total;                      // input frames consumed by the piece
  uint16_t sched_c[PG_SCHED_CAP];
  float sched_f[PG_SCHED_CAP];
};

enum PgUnitKind { UNIT_SUBMIXER = 0, UNIT_SOURCE = 1, UNIT_BUS = 2, UNIT_EFFECT = 3 };

struct PgUnit {
  int32_t kind;
  int32_t n_voices, voice_off;  // into the voice index table
  int32_t n_fx, fx_off;         // into the fx index table
  int32_t effects_bypassed;     // MixedSource::effects_bypassed (src/source/mixed.rs:202)
  uint64_t silence_counter;     // SubMixerProcessor (src/source/mixed/submixer.rs:23)
  int32_t audible;              // result of the last chunk: contributes to the parent's `audible_input`
  int32_t deferred;             // set by the fast kernel when the unit must be rendered by the generic kernel
  int32_t static_defer;         // host: the chain holds an effect kind without a time-parallel path -> always generic kernel (no stock kind or layout sets it any more)
  int32_t maybe_ramping;        // device: a parameter command was applied and some smoother may still ramp (cleared by the generic
                                // kernel once every effect of the unit is back in steady state)
  int32_t voice0;               // host: device index of the unit's first voice (skips one dependent load at kernel start)
  int32_t fx0;                  // host: device index of the unit's first effect (prefetched during the source stage)
  int32_t staged;               // host: reverb-terminated chain, eligible for the staged pipeline: 1 lean, 2 wide leading effects, 3 wide + source adapters (pg_stage_body.inl)
  int32_t stage_flags;          // device: hand-over between the stage kernels of one block (PG_STAGE_*)
  int32_t child_off, n_children;  // host: nested sub-mixers of this mixer, entries of PgLaunch::child_rows (summed before the sources)
  int32_t seg_idx;              // device: semantic chunks this mixer has begun in the main mixer's current chunk, minus one (indexes its sub-mixers' call_audible)
  uint64_t call_audible;        // device: bit k = result of SubMixerProcessor::process for the k-th call of the main mixer's current chunk (a parent
                                // with events inside it calls its sub-mixers once per segment, mixed.rs:679-712)
  // device: the chunk (this mixer's own MixedSource::write chunk) and the call (SubMixerProcessor::process of its parent) in progress, across pieces
  int32_t chunk_audible_input;  // audible_input of the chunk, decided at its first piece (process_effects and every processor of the chain use it)
  int32_t chunk_any_audible;    // main-mixer sources: OR of `produced_output` over the pieces so far
  float call_max;               // peak of the call's output so far (submixer.rs:57: max_abs over the whole call)
  uint32_t call_frames;         // frames of the call rendered in earlier pieces
  int32_t call_idx;             // calls finished in the main mixer's current chunk
  int32_t pad_call;
};
enum { PG_STAGE_ACTIVE = 1, PG_STAGE_INPUT_BYPASSED = 2, PG_STAGE_ALL_BYPASSED = 4, PG_STAGE_AUDIBLE = 8, PG_STAGE_SKIPPED = 16,
       PG_STAGE_FIRST = 32, PG_STAGE_LAST = 64 };  // the block is the first / last piece of its chunk

enum PgCmdType {
  CMD_FX_PARAM = 0,     // target = fx index, param = parameter index, value = raw (already denormalized/clamped) value
  CMD_FX_RESET = 1,
  CMD_VOICE_VOLUME = 2, // target = voice index
  CMD_VOICE_PAN = 3,
  CMD_VOICE_STOP = 4,   // sets stop_time = value64
  CMD_VOICE_SPEED = 5,  // value64 = f64 bits of the speed, value = glide (semitones/s, <= 0: none)
  CMD_VOICE_SEEK = 6,   // value64 = f64 bits of the position in seconds
  CMD_CALL_SPLIT = 7,   // nested sub-mixers: an ancestor splits its block at `frame` -> this unit's write() call ends there and a new one begins
  CMD_NOP = 8,          // an event whose effect was removed before it came due: nothing to apply, but the mixer's block still ends a segment at its
                        // time (the reference pops the event, logs "not found" and carries on, mixed.rs:862-924 — per-call logic counts calls)
  // Markers (frame = the launch's frame count: never applied, never a split): where the unit's current chunk / call ends when that is NOT the
  // end of the main mixer's chunk — the unit (CHUNK_END) or one of its ancestors (CALL_END) has its next event at position value64, inside the
  // main chunk but beyond this piece. The kernels close per-chunk / per-call state at a piece's end when the marker's position is that end.
  CMD_CHUNK_END = 9,
  CMD_CALL_END = 10,
};
#define PG_MAX_CALLS 64  // calls of one sub-mixer per launch round (bits of PgUnit::call_audible); the host bounds the round accordingly
struct PgCmd {
  int32_t type, unit, target, param;
  uint32_t frame;  // offset in frames from the start of this launch at which the command applies
  float value;
  uint64_t value64;
};

// A round's command list as a kernel argument of the decision-scan kernel (rounds of at most PG_CMD_PACK commands: nearly all of them): the scan
// reads its commands from here and leaves them in the device ring for the kernels behind it — no copy kernel in front of the round (5-6 us on
// the stream in a kernel trace of the dynamic rounds).
#define PG_CMD_PACK 16
struct PgCmdPack { PgCmd c[PG_CMD_PACK]; int32_t n; int32_t pad[3]; };
#define PG_BUS_PIPELINE_MAX 16  // effects of a bus chain the pipelined launch takes (one workgroup each); longer chains stay one workgroup
struct PgLaunch {
  PgUnit* units;
  PgVoice* voices;
  PgFx* fx;
  const int32_t* voice_index;
  const int32_t* fx_index;
  const PgCmd* cmds;
  int32_t n_cmds;
  int32_t n_units;
  int32_t unit_base;      // first unit slot this launch processes (when unit_order == nullptr)
  const int32_t* unit_order;  // block b processes unit slot unit_order[b] and writes row b of unit_out
  uint32_t n_frames;      // frames of this launch (<= PG_MAX_FRAMES)
  uint64_t pos;           // SourceTime.pos_in_frames of the first frame
  uint32_t sample_rate;
  int32_t fast;           // 1: time-parallel paths enabled
  int32_t wide;           // fast kernel variant: 0 = lean (Gain/Panning/Reverb), 1 = all fast-capable kinds, 2 = all but Reverb / Compressor (four workgroups per CU)
  int32_t mode;           // 0: generic kernel, all units; 1: fast kernel (defers ineligible units); 2: generic kernel, deferred units only;
                          // 3: generic kernel, the bus chain behind a super-block as a pipeline: workgroup f = effect f of unit `unit_base`
  float* unit_out;        // [n_units][out_stride] per-unit output (sub-mixer / source results)
  uint32_t out_stride;    // floats per unit row
  float* bus;             // bus / external signal for UNIT_BUS and UNIT_EFFECT (in place); unit slot b of the launch: bus + b * bus_unit_stride
  int32_t* bus_audible;   // input flag for UNIT_BUS (1 = audible input)
  PgSchedEntry* sched;    // [n_classes][2 banks]; nullptr disables the schedule cache
  int32_t sched_bank;     // bank read by this launch; the representatives write bank ^ 1
  unsigned long long* diag;  // diagnostic builds (-DPG_DIAG): shader-clock stamps of workgroup 0, else unused
  double* stage_buf;      // [n_units][PG_STAGE_BUF_DOUBLES] f64 hand-over buffer of the staged pipeline (nullptr: pipeline off)
  const int4* slot_info;  // [n_units] per launch slot: {unit slot, device index of the first voice, device index of the last effect, number of voices | PgUnit::staged << 24}
  int32_t* defer_count;   // fast kernels append the launch slots they defer: count of this round ...
  int32_t* defer_list;    // ... and the slots; the generic kernel (mode 2) walks the list
  int32_t* defer_reset;   // the other round's counter, zeroed by the generic kernel for the next round
  int32_t* defer_state;   // pre-scanned rounds: how many of the deferred units were deferred for their STATE (static_defer / maybe_ramping) rather than for a command
  int32_t* defer_state_reset;   // ... in this round (the scan kernel counts, the generic kernel reports it: host_feedback[3]); the other round's word, zeroed like defer_reset
  unsigned long long* host_feedback;  // pinned host word: the generic kernel reports (round << 32 | units it found deferred)
  uint32_t round;         // launch counter of this round
  int32_t staged_on;      // units whose `staged` level is 1 .. staged_on are rendered by the stage kernels of this round, the fused fast kernel skips them
  uint64_t call_end;      // position at which the MixedSource::write call this launch belongs to ends (see PgVoice::zombie_end)
  const float* rows_base; // nested sub-mixers: row 0 of the per-unit output table (unit_out points at this launch's level) ...
  const int2* child_rows; // ... and {row, unit slot} of every nested sub-mixer, indexed by PgUnit::child_off
  // Super-block launch (steady state, no command inside): every workgroup of the fast / staged kernels renders n_chunks consecutive
  // blocks of n_frames for its unit — block c covers frames [pos + c * n_frames, +n_frames) and goes to unit_out + c * chunk_stride —
  // with all per-block decisions (bypass counters, silence gates, tails) taken per block exactly as in n_chunks single launches.
  // Workgroups are independent until the mixer sum, so they drift out of lock-step across the blocks. 0 / 1: one block.
  int32_t n_chunks;
  int32_t pad_chunks;
  uint64_t chunk_stride;  // floats between the per-unit output tables of consecutive blocks
  int32_t* error_word;    // device word of sticky consistency flags (PG_DEVERR_*), nullptr: not collected
  uint32_t fast_scratch_bytes;  // host: LDS arena of the fast kernels for the effect kinds this graph holds (0: the full arena)
  uint32_t pad_scratch;
  int32_t* index_log;     // test hook (generic kernel only, see FastCtx::idx_log): read indices of the time-parallel delay-line paths
  // Per-block `audible` results of this level's units: block c of launch slot b -> audible_tab[c * audible_stride + b]. PgUnit::audible holds
  // the last block only; the mixer sum ORs one row of this table per block (audible_input of process_effects, mixed.rs:696-706), so the bus
  // chain behind a super-block launch sees every block's own flag. nullptr: not collected (standalone effects, bus launches).
  int32_t* audible_tab;
  uint64_t audible_stride;
  unsigned long long* bus_progress;  // mode 3 (pipelined bus chain): [PG_BUS_PIPELINE_MAX] words (round << 32 | `active` flags of the last 24 blocks << 8 | blocks done), then
                                     // [PG_BUS_PIPELINE_MAX] words with one `active` flag per block of the launch; device memory
  const int2* slot_fx;    // [n_units] per launch slot: device indices of the unit's first two effects (-1: none) — with slot_info the fast kernels
                          // request unit record, first voice and the first two effect states side by side instead of one after the other
  const int4* slot_lead;  // [n_units] per launch slot, staged units: device indices of the first three effects in FRONT of the reverb (-1: none) —
                          // stage 1 requests their state blocks when the workgroup starts (LDS-DMA), under the source stage
  uint64_t bus_unit_stride;  // floats between the external buffers of consecutive units of the launch (a standalone effect with more than two
                             // channels runs one stereo unit per channel pair); 0 for the bus
  // Chunk grid of the main mixer (see PG_MAX_FRAMES): block 0 of the launch starts grid_off frames behind a chunk start, chunks of
  // PG_MAX_FRAMES follow each other from there until grid_span frames behind that chunk start (the end of the write call or the next
  // main-mixer event). grid_span == 0: every block is a chunk of its own (standalone effects, launches of one whole chunk).
  uint32_t grid_off, grid_span;
};
// Where block c of a launch sits in the main mixer's chunk grid.
struct PgPiece {
  bool first, last;       // first / last piece of its chunk
  int c_last;             // block (relative to the launch's block 0) that holds the chunk's last frame
  uint64_t chunk_end;     // position (frames) at which the chunk ends
};
// PgLaunch::error_word bits: conditions the host's routing must make impossible; a set bit means wrong audio, never a crash.
enum { PG_DEVERR_FAST_DECLINED = 1,   // a kernel without serial effect code met an effect state its time-parallel path does not take
       PG_DEVERR_SUPER_DEFERRED = 2,  // a unit wanted the generic kernel inside a super-block launch
       PG_DEVERR_BUS_STALLED = 4 };   // a stage of the pipelined bus chain gave up waiting for the stage in front of it
constexpr int PG_STAGE_BUF_DOUBLES = 2 * 1024 + 128 + 8;
