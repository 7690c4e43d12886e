// gfx950 kernels of the phonic DSP hot path.
//
//   pg_unit_kernel   one workgroup per work unit — a sub-mixer (its sources + its effect chain), a lone
//                    main-mixer source, the main mixer's bus chain, or a standalone effect. The unit's block
//                    of interleaved stereo f32 frames lives in LDS from the source stage to the end of the
//                    effect chain (no intermediate round trips to HBM); persistent state (delay lines,
//                    filter state, smoothers, resampler history) lives in HBM.
//   pg_mix_kernel_*  the mixer-graph sum (reference add_buffers per source, src/source/mixed.rs:606-608):
//                    a deterministic two-stage tree over units, coalesced over the sample index.
//
// Written for CDNA4 only: 64-wide wavefronts, LDS-resident signal, f64 vector ALU for the filter/delay
// arithmetic the reference does in f64. No MFMA: nothing on this path is a dense contraction.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <mutex>

#include "pg_dev.h"
#include "pg_dsp_dev.h"
#include "pg_fx_serial.h"
#include "pg_source_dev.h"
#include "pg_fx_fast.h"

using namespace pgd;

// A unit wanted the generic kernel inside a super-block launch: its later blocks stay unrendered. The sticky device word keeps the flag
// for pg_graph_device_errors; the copy in the host-mapped feedback block lets the next write see it WITHOUT a synchronisation of its own and
// fail the graph (GuardedSource semantics) instead of handing out wrong audio silently.
__device__ __forceinline__ void pg_raise_device_error(const PgLaunch& L, int bit) {
  if (L.error_word) atomicOr(L.error_word, bit);
  if (L.host_feedback) { *(volatile unsigned long long*)(L.host_feedback + 2) = (unsigned long long)bit; __threadfence_system(); }
}
__device__ __forceinline__ void pg_raise_super_deferred(const PgLaunch& L) { pg_raise_device_error(L, PG_DEVERR_SUPER_DEFERRED); }

// Where block c of the launch sits in the main mixer's chunk grid (PgLaunch::grid_off / grid_span, pg_dev.h).
__device__ __forceinline__ PgPiece pg_piece(const PgLaunch& L, int c) {
  PgPiece p;
  const uint32_t N = L.n_frames;
  if (L.grid_span == 0) { p.first = true; p.last = true; p.c_last = c; p.chunk_end = L.pos + (uint64_t)(c + 1) * (uint64_t)N; return p; }
  const uint32_t o = L.grid_off + (uint32_t)c * N;
  const uint32_t cs = o & ~((uint32_t)PG_MAX_FRAMES - 1u);
  uint32_t ce = cs + (uint32_t)PG_MAX_FRAMES;
  if (ce > L.grid_span) ce = L.grid_span;
  p.first = o == cs;
  p.last = o + N >= ce;
  p.c_last = c + (int)((ce - o - 1u) / N);
  p.chunk_end = L.pos - (uint64_t)L.grid_off + (uint64_t)ce;
  return p;
}

// ---- parameter updates (Effect::process_parameter_update of each effect), lane 0 --------------------
// Returns 1 when the whole workgroup must flush state afterwards (compressor look-ahead line re-created).
// aux: time-constant coefficients the HOST computed for this update (pg_host.hip: fx_param_aux) — exp(-1 / (t * fs)) sits within 1e-5 of one, where a
// one-ulp difference between two libm implementations of expf is half a percent of the time constant; the reference's value is the host libm's.
__device__ __noinline__ int fx_apply_param(PgFx& fx, int param, float value, unsigned long long aux) {
  const float aux_lo = __uint_as_float((uint32_t)aux), aux_hi = __uint_as_float((uint32_t)(aux >> 32));
  uint32_t sr = fx.sample_rate;
  switch (fx.kind) {
    case 0: {  // gain.rs:177-205
      PgGain& g = fx.u.gain;
      if (param == P_GAIN_GAIN) sm_set_target(g.gain, value);
      else {
        g.dc_mode = (int)value;
        if (g.dc_mode != 0) { double hz = g.dc_mode == 1 ? 1.0 : (g.dc_mode == 2 ? 5.0 : 20.0); g.dc[0].r = dc_r(hz, sr); g.dc[1].r = dc_r(hz, sr); }
        else { g.dc[0].x1 = g.dc[0].y1 = 0.0; g.dc[1].x1 = g.dc[1].y1 = 0.0; }
      }
    } break;
    case 1: {  // pan.rs:164-191
      PgPan& p = fx.u.pan;
      if (param == P_PAN_PAN) sm_set_target(p.pan, value);
      else if (param == P_PAN_WIDTH) sm_set_target(p.width, value);
      else if (param == P_PAN_INVL) p.invert_l = value != 0.0f;
      else p.invert_r = value != 0.0f;
    } break;
    case 2: {  // filter.rs:209-237
      PgFilter& f = fx.u.filter;
      if (param == P_FILTER_TYPE) {
        f.type = (int)value;
        int bt = filter_to_biquad(f.type);
        if (f.coef.type != bt) { f.coef.type = bt; biquad_apply(f.coef); }
      } else if (param == P_FILTER_CUTOFF) sm_set_target(f.cutoff, value);
      else sm_set_target(f.q, value);
    } break;
    case 3: {  // eq5.rs:334-363
      PgEq5& e = fx.u.eq5;
      int band = param / 3, which = param % 3;
      if (which == 0) sm_set_target(e.gains[band], value);
      else if (which == 1) sm_set_target(e.freqs[band], value);
      else sm_set_target(e.bws[band], value);
      eq5_update_filter_coefficients(fx);
    } break;
    case 4: {  // delay.rs:490-520
      PgDelay& d = fx.u.delay;
      switch (param) {
        case P_DELAY_MODE: d.mode = (int)value; break;
        case P_DELAY_TIME: sm_set_target(d.delay_time, value); break;
        case P_DELAY_FEEDBACK: sm_set_target(d.feedback, value); break;
        case P_DELAY_FTYPE: d.filter_type = (int)value; break;
        case P_DELAY_CUTOFF: sm_set_target(d.cutoff, value); break;
        case P_DELAY_DRIVE: sm_set_target(d.drive, value); break;
        case P_DELAY_WET: sm_set_target(d.wet, value); break;
        case P_DELAY_WIDTH: sm_set_target(d.width, value); break;
        case P_DELAY_LFO_RATE: sm_set_target(d.lfo_rate, value); break;
        case P_DELAY_LFO_SHAPE: d.lfo_shape = (int)value; d.lfo.waveform = (int)value; break;
        case P_DELAY_D_TIME: sm_set_target(d.d_time, value); break;
        case P_DELAY_D_FEEDBACK: sm_set_target(d.d_feedback, value); break;
        default: sm_set_target(d.d_filter, value); break;
      }
    } break;
    case 5: {  // reverb.rs:496-512
      if (param == P_REVERB_ROOM) sm_set_target(fx.u.reverb.room, value);
      else sm_set_target(fx.u.reverb.wet, value);
    } break;
    case 6: {  // chorus.rs:433-459
      PgChorus& c = fx.u.chorus;
      switch (param) {
        case P_CHORUS_RATE: sm_set_target(c.rate, value); break;
        case P_CHORUS_DEPTH: sm_set_target(c.depth, value); break;
        case P_CHORUS_FEEDBACK: sm_set_target(c.feedback, value); break;
        case P_CHORUS_DELAY: sm_set_target(c.delay, value); break;
        case P_CHORUS_WET: sm_set_target(c.wet, value); break;
        case P_CHORUS_PHASE: sm_set_target(c.phase, value); break;
        case P_CHORUS_FTYPE: {
          c.filter_type = (int)value;
          int st = delay_to_svf(c.filter_type);
          if (c.coef.type != st) { c.coef.type = st; svf_apply(c.coef); }
        } break;
        case P_CHORUS_FREQ: sm_set_target(c.freq, value); break;
        default: sm_set_target(c.res, value); break;
      }
    } break;
    case 7: {  // compressor.rs:304-330
      PgComp& c = fx.u.comp;
      float old_lookahead = c.lookahead;
      switch (param) {
        case P_COMP_THRESHOLD: c.threshold = value; break;
        case P_COMP_RATIO: c.ratio = value; break;
        case P_COMP_KNEE: c.knee = value; break;
        case P_COMP_ATTACK: c.attack = value; c.env_attack = aux_lo; break;    // (update_coefficients recomputes both from the stored times:
        case P_COMP_RELEASE: c.release = value; c.env_release = aux_lo; break;  //  only the one whose time changed can change)
        case P_COMP_MAKEUP: sm_set_target(c.makeup, value); break;
        default: c.lookahead = value; break;
      }
      if (c.lookahead != old_lookahead) {  // LookupDelayLine::new  delay.rs:182-203
        uint32_t df = (uint32_t)f2u64(ceilf(c.lookahead * (float)sr));
        c.delay_frames = df;
        uint32_t p = 1; while (p < df) p <<= 1;
        c.mask = df > 0 ? p - 1 : 0;
        c.write_pos = 0; c.peak_pos = 0; c.peak_value = 0.0;
        return 1;
      }
    } break;
    case 8: {  // gate.rs:203-223
      PgGate& g = fx.u.gate;
      switch (param) {
        case P_GATE_THRESHOLD: g.threshold = value; break;
        case P_GATE_ATTACK: g.attack = value; g.env_attack = aux_lo; g.attack_coeff = aux_hi; break;
        case P_GATE_HOLD: g.hold = value; break;
        case P_GATE_RELEASE: g.release = value; g.env_release = aux_lo; g.release_coeff = aux_hi; break;
        default: g.range = value; break;
      }
    } break;
    default: {  // distortion.rs:368-385
      PgDist& d = fx.u.dist;
      if (param == P_DIST_TYPE) d.type = (int)value;
      else if (param == P_DIST_DRIVE) sm_set_target(d.drive, value);
      else sm_set_target(d.mix, value);
    } break;
  }
  return 0;
}

__device__ void wg_fill_zero(double* p, size_t n) {
  for (size_t i = pg_tid(); i < n; i += blockDim.x) p[i] = 0.0;
}

// Reset messages (DelayEffectMessage::Reset, ReverbEffectMessage::Reset, ChorusEffectMessage::Reset) and the
// compressor's look-ahead re-creation: flushes run on the whole workgroup.
__device__ __noinline__ void fx_flush_wg(PgFx& fx, int reset_message) {
  __syncthreads();
  switch (fx.kind) {
    case 4: {  // DelayEffect::reset  delay.rs:213-223
      PgDelay& d = fx.u.delay;
      wg_fill_zero(d.line[0], (size_t)d.mask + 1);
      wg_fill_zero(d.line[1], (size_t)d.mask + 1);
      if (pg_tid() == 0) {
        d.write_pos[0] = d.write_pos[1] = 0;
        d.flt[0].ic1eq = d.flt[0].ic2eq = d.flt[1].ic1eq = d.flt[1].ic2eq = 0.0;
        d.dc[0].x1 = d.dc[0].y1 = d.dc[1].x1 = d.dc[1].y1 = 0.0;
        delay_lfo_reset(d);
        d.fb[0] = d.fb[1] = 0.0f;
      }
    } break;
    case 5: {  // reverb.rs:469-487 (flush of all 13 lines; DelayLine/Allpass flush also resets write_pos)
      PgReverb& r = fx.u.reverb;
      for (int i = 0; i < 8; ++i) wg_fill_zero(r.line[i].buf, (size_t)r.line[i].frames * 2);
      for (int i = 0; i < 4; ++i) wg_fill_zero(r.ap[i].buf, (size_t)r.ap[i].frames * 2);
      wg_fill_zero(r.pre, ((size_t)r.pre_mask + 1) * 2);
      if (pg_tid() == 0) { for (int i = 0; i < 4; ++i) r.ap[i].write_pos = 0; r.pre_write_pos = 0; }
    } break;
    case 6: {  // ChorusEffect::reset  chorus.rs:201-210
      PgChorus& c = fx.u.chorus;
      wg_fill_zero(c.line[0], (size_t)c.mask + 1);
      wg_fill_zero(c.line[1], (size_t)c.mask + 1);
      if (pg_tid() == 0) {
        c.write_pos[0] = c.write_pos[1] = 0;
        c.flt[0].ic1eq = c.flt[0].ic2eq = c.flt[1].ic1eq = c.flt[1].ic2eq = 0.0;
        sm_init(c.rate, c.rate.target);
        sm_init(c.phase, c.phase.target);
        c.current_phase = 0.0;
        chorus_reset_lfos(fx);
      }
    } break;
    case 7: {
      if (!reset_message) wg_fill_zero(fx.u.comp.line, (size_t)fx.u.comp.line_frames * 2);
    } break;
    default: break;
  }
  __syncthreads();
}

// ---- Effect::process dispatch: time-parallel steady-state path when eligible, exact serial path otherwise ----
// FAST_ONLY kernels contain no serial effect code at all (register budget); eligibility was checked up front.
// The effect's own call-end bookkeeping, behind the last piece of a process call. `n` = this piece's samples; fx.call_frames = the frames of
// the call's earlier pieces (kept by the processor; 0 for a standalone effect, whose every launch is one call).
template <int KMASK>
__device__ __forceinline__ void fx_call_end(PgFx& fx, int n, bool call_last) {
  if constexpr ((KMASK >> 6) & 1) {
    if (fx.kind == 6 && call_last) {
      __syncthreads();
      if (pg_tid() == 0) chorus_call_end(fx, (uint64_t)fx.call_frames + (uint64_t)(n / 2));
      __syncthreads();
    }
  }
}
template <bool FAST_ONLY, int KMASK>
__device__ __forceinline__ void fx_process_wg(PgFx& fx, float* sig, int n, FastCtx& fc, int fast, bool call_last) {
  if (FAST_ONLY) {
    // no serial code in this kernel: the host's routing (lean / wide / staged) and the eligibility check of the previous block must
    // agree with what the time-parallel path accepts. A decline here leaves the effect unapplied for this block: make it visible.
    if (!fx_fast_process<KMASK>(fx, sig, n, fc) && fc.err && pg_tid() == 0) atomicOr(fc.err, PG_DEVERR_FAST_DECLINED);
    fx_call_end<KMASK>(fx, n, call_last);
    return;
  } else {
    // One pass in all cases but one: a Reverb whose room size moves. Its linear smoother arrives after `pending` frames (<= 109 at 44.1 kHz:
    // reverb.rs:78-84, smoothing.rs:370-382) — only those run on the serial lane; behind them the frame loop of reverb.rs:318-338 keeps
    // calling next() on settled smoothers, which is the steady state (or the wet ramp) the time-parallel paths render.
    int off = 0;
#pragma nounroll
    while (off < n) {
      if (fast && fx_fast_process<KMASK>(fx, sig + off, n - off, fc)) break;
      __syncthreads();
      int head = n - off;
      if (fast && off == 0 && fx.kind == 5 && fx.u.reverb.room.kind == SM_LIN && fx.u.reverb.room.pending > 0 &&
          (long long)fx.u.reverb.room.pending * 2 < (long long)head) head = (int)fx.u.reverb.room.pending * 2;
      __syncthreads();  // every lane holds `head` before lane 0 moves the smoother
      float* s = sig + off;
      if (pg_tid() == 0) {
        switch (fx.kind) {
          case 0: gain_serial(fx, s, head); break;
          case 1: pan_serial(fx, s, head); break;
          case 2: filter_serial(fx, s, head); break;
          case 3: eq5_serial(fx, s, head); break;
          case 4: delay_serial(fx, s, head); break;
          case 5: reverb_serial(fx, s, head); break;
          case 6: chorus_serial(fx, s, head); break;
          case 7: comp_serial(fx, s, head); break;
          case 8: gate_serial(fx, s, head); break;
          default: dist_serial(fx, s, head); break;
        }
      }
      __syncthreads();
      off += head;
    }
    fx_call_end<KMASK>(fx, n, call_last);
  }
}

// ---- EffectProcessor::process  src/source/mixed/effect.rs:56-145 -------------------------------------
// One process call of the processor = one chunk of its mixer, rendered as pieces: `first` / `last` name the chunk's first / last piece.
// ctl: LDS words for uniform decisions. Returns true when the effect processed output.
// pre-part: the bypass decision (:88-101), taken at the chunk's first piece and kept for its later ones. Returns true when the effect is
// bypassed for this chunk. All lanes call.
__device__ __forceinline__ bool fx_processor_pre(PgFx& fx, bool input_bypassed, bool first, int* ctl) {
  __syncthreads();
  if (pg_tid() == 0) {
    if (first) {
      bool should_bypass = input_bypassed && fx.tail_counter == 0 && fx.silence_counter == PG_USIZE_MAX;  // :88-91
      if (should_bypass && !fx.bypassed) fx.bypassed = 1;                                                // process_stopped: no-op for stock effects
      else if (!should_bypass && fx.bypassed) { fx.bypassed = 0; fx.tail_counter = PG_USIZE_MAX; fx.silence_counter = 0; }
      fx.call_frames = 0; fx.call_max = 0.0f;
      fx_call_begin(fx);
    }
    ctl[0] = fx.bypassed;
  }
  __syncthreads();
  return ctl[0] != 0;
}
// post-part: update_tail_counters / reset_tail_counters (:111-152) once the effect has rendered the whole chunk: `n` samples of this piece in
// `sig`, fx.call_frames frames in the pieces before it. The counters move at the last piece, by the chunk's length; the silence detection
// looks at the peak over the whole chunk (fx.call_max carries it from piece to piece).
__device__ __forceinline__ void fx_processor_post(PgFx& fx, const float* sig, int n, bool input_bypassed, bool last, uint32_t sample_rate, int* ctl, float* red) {
  if (input_bypassed) {  // update_tail_counters :111-145
    if (pg_tid() == 0) {
      uint64_t tail_frames;
      if (fx_process_tail(fx, tail_frames)) {
        if (last) {
          if (tail_frames == PG_USIZE_MAX) fx.tail_counter = tail_frames;
          else if (fx.tail_counter == PG_USIZE_MAX) fx.tail_counter = tail_frames;
          else { uint64_t fp = (uint64_t)fx.call_frames + (uint64_t)(n / 2); fx.tail_counter = fx.tail_counter > fp ? fx.tail_counter - fp : 0; }
          fx.silence_counter = PG_USIZE_MAX;
        }
        ctl[1] = 0;
      } else ctl[1] = 1;
    }
    __syncthreads();
    if (ctl[1]) {  // unknown tail: detect silence
      float max_sample = wg_max_abs(sig, n, red);
      if (pg_tid() == 0) {
        max_sample = fmaxf(max_sample, fx.call_max);
        if (!last) fx.call_max = max_sample;
        else if (max_sample < 0.001f) {
          uint64_t fp = (uint64_t)fx.call_frames + (uint64_t)(n / 2);
          fx.silence_counter = (fx.silence_counter > PG_USIZE_MAX - fp) ? PG_USIZE_MAX : fx.silence_counter + fp;
          if (fx.silence_counter >= 2ull * (uint64_t)sample_rate) { fx.tail_counter = 0; fx.silence_counter = PG_USIZE_MAX; }
        } else fx.silence_counter = 0;
      }
    }
  } else if (pg_tid() == 0 && last) {
    fx.tail_counter = PG_USIZE_MAX; fx.silence_counter = 0;  // reset_tail_counters :148-152
  }
  if (pg_tid() == 0) { if (last) { fx.call_frames = 0; fx.call_max = 0.0f; fx.call_ramp = 0; } else fx.call_frames += (uint32_t)(n / 2); }
  __syncthreads();
}
template <bool FAST_ONLY, int KMASK>
__device__ __forceinline__ bool fx_processor_process(PgFx& fx, float* sig, int n, bool input_bypassed, bool first, bool last, uint32_t sample_rate, FastCtx& fc, int fast,
                                     int* ctl, float* red) {
  if (fx_processor_pre(fx, input_bypassed, first, ctl)) return false;
  PG_STAMP(fc.diag, 9);
  fx_process_wg<FAST_ONLY, KMASK>(fx, sig, n, fc, fast, last);
  fx_processor_post(fx, sig, n, input_bypassed, last, sample_rate, ctl, red);
  return true;
}

// ---- the unit kernel ----------------------------------------------------------------------------------
// dynamic LDS: [sig 2*n_frames f32][tmp 2*n_frames f32][scratch]
extern __shared__ __attribute__((aligned(16))) char pg_smem[];

// SubMixerProcessor::process (src/source/mixed/submixer.rs:47-77): one call = one write() of this sub-mixer = one chunk of its parent,
// rendered as pieces. Frames [a, b) of this piece belong to the call in progress; `closes`: the call ends at b. The silence gate looks at the
// peak of the WHOLE call (unit.call_max carries it from piece to piece, unit.call_frames the frames of the earlier pieces) and decides when
// the call closes; until then a piece's samples go to the unit's output row as they are, and a call that closes below the gate takes them
// back — its rows of the earlier pieces sit in the tables in front of this one (`table_stride` floats apart, `piece_frames` frames each;
// the mixer sum runs behind a chunk's last piece). Returns whether the call produced output (meaningful when it closes).
// (A sub-mixer without sources, effects or events returns 0 samples: max over an empty slice = 0 -> silent.)
__device__ __forceinline__ bool submixer_call_piece(PgUnit& unit, int* ur_call /* LDS copy of {call_max, call_frames} */, const float* sig, float* out, int a, int b, bool closes,
                                                     uint32_t sample_rate, size_t table_stride, int piece_frames, int* ctl, float* red) {
  const int tid = pg_tid(), nt = blockDim.x;
  // (the piece's samples go out while its peak is found: a call that closes below the gate — 2 s of silence, once — takes them back below)
  for (int i = 2 * a + tid; i < 2 * b; i += nt) out[i] = sig[i];
  const float max_sample = wg_max_abs(sig + 2 * a, 2 * (b - a), red);
  if (tid == 0) {
    const float peak = fmaxf(max_sample, __int_as_float(ur_call[0]));
    const uint32_t before = (uint32_t)ur_call[1];
    int audible = 1, take_back = 0;
    if (!closes) { unit.call_max = peak; unit.call_frames = before + (uint32_t)(b - a); ur_call[0] = __float_as_int(peak); ur_call[1] = (int)(before + (uint32_t)(b - a)); }
    else {
      if (peak < 0.001f) {
        unit.silence_counter += (uint64_t)before + (uint64_t)(b - a);
        audible = unit.silence_counter < 2ull * (uint64_t)sample_rate ? 1 : 0;
      } else unit.silence_counter = 0;
      take_back = (!audible && before > 0) ? (int)before : 0;
      unit.call_max = 0.0f; unit.call_frames = 0; ur_call[0] = 0; ur_call[1] = 0;
    }
    ctl[3] = audible; ctl[4] = take_back;
  }
  __syncthreads();
  const bool audible = ctl[3] != 0;
  if (!audible) { for (int i = 2 * a + tid; i < 2 * b; i += nt) out[i] = 0.0f; }
  int back = ctl[4];
  if (back > 0 && a == 0) {  // (a call that began in an earlier piece reaches this one at its frame 0)
    float* row = out;
    while (back > 0) {
      row -= table_stride;
      const int k = back < piece_frames ? back : piece_frames;
      for (int i = 2 * (piece_frames - k) + tid; i < 2 * piece_frames; i += nt) row[i] = 0.0f;
      back -= k;
    }
  }
  __syncthreads();
  return audible;
}

#define PG_MIN_ROW_FRAMES 64
// What a workgroup of the fast kernels keeps from one block of a super-block launch to the next: the slot tables' entry and the unit record's
// host-written words in registers; the LDS copies of the unit record, of the unit's voice and of its first two effects hold the state the block
// left (every change also goes to global memory, as before). The later blocks then start without the two dependent trips to L2 at the head
// of the body (slot tables -> records) and without re-staging what is already there — on a workgroup whose block is a latency chain
// (C3: 6.5 K of 70 K cycles per block).
struct PgUnitCarry { int4 si; int2 sf; uint32_t unit_w; int resident; int fx_valid; };   // fx_valid: the two effect slots were filled (the chain ran) in an earlier block
template <bool FAST_ONLY, int KMASK>
__device__ __forceinline__ void pg_unit_body(const PgLaunch& L, const int slot, const int chunk, PgUnitCarry& carry) {
  if (slot >= L.n_units) return;
  const int tid = pg_tid(), nt = blockDim.x;
  const bool resident = FAST_ONLY && chunk > 0 && carry.resident != 0;
  // Fast kernels: ONE trip names the unit, its first voice and its first two effects (slot_info / slot_fx, written by the host with the
  // topology); their records are then requested together — the unit record, the voice's state (one dword per lane) and the effect states (one
  // qword per lane each) used to be three dependent trips to L2 at the head of a kernel that is a latency chain.
  const bool tables = FAST_ONLY && L.slot_info != nullptr && L.slot_fx != nullptr;
  int4 si = make_int4(0, 0, 0, 0);
  int2 sf = make_int2(-1, -1);
  if (resident) { si = carry.si; sf = carry.sf; }
  else if (tables) {
    si = L.slot_info[slot]; sf = L.slot_fx[slot];
    si.x = __builtin_amdgcn_readfirstlane(si.x); si.y = __builtin_amdgcn_readfirstlane(si.y); si.w = __builtin_amdgcn_readfirstlane(si.w);
    sf.x = __builtin_amdgcn_readfirstlane(sf.x); sf.y = __builtin_amdgcn_readfirstlane(sf.y);
  }
  const int u = tables ? si.x : (L.unit_order ? L.unit_order[slot] : L.unit_base + slot);
  PgUnit& unit = L.units[u];
  // The host-written fields of the unit record, read ONCE (one dword per lane of wave 0's first lanes): every `unit.x` further down would be
  // another dependent trip to L2 — the record is also written in this body, so the compiler reloads it behind every barrier — on a workgroup
  // whose block is a latency chain. What decides in front of the first barrier comes out of the register by v_readlane; the rest of the body
  // reads the copy in LDS (`ur`, visible behind that barrier): no register lives across the body for it.
  uint32_t unit_w = 0;
  if (resident) unit_w = carry.unit_w;   // (the words read in front of the first barrier are host-written: unchanged since the launch's first block)
  else if ((tid & 63) < (int)(sizeof(PgUnit) / 4)) unit_w = ((const uint32_t*)&unit)[tid & 63];
#define PG_UF(f) ((int)__builtin_amdgcn_readlane((int)unit_w, (int)(offsetof(PgUnit, f) / 4)))
#define PG_UL(f) (ur[offsetof(PgUnit, f) / 4])
  const int u_static_defer = PG_UF(static_defer), u_maybe_ramping = PG_UF(maybe_ramping), u_fx0 = PG_UF(fx0), u_staged = PG_UF(staged), u_n_fx0 = PG_UF(n_fx), u_kind0 = PG_UF(kind);
  uint32_t voice_word = 0;
  unsigned long long fx0_word = 0, fx1_word = 0;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  static_assert(sizeof(PgFx) % 8 == 0 && sizeof(PgFx) / 8 <= 256, "PgFx must fit one qword per lane of the workgroup");
  if (tables && !resident) {
    if ((si.w & 0xffffff) > 0 && tid < (int)(sizeof(PgVoice) / 4)) voice_word = ((const uint32_t*)&L.voices[si.y])[tid];
    if (sf.x >= 0 && tid < n_fx_words / 2) fx0_word = ((const unsigned long long*)&L.fx[sf.x])[tid];
    if (sf.y >= 0 && tid < n_fx_words / 2) fx1_word = ((const unsigned long long*)&L.fx[sf.y])[tid];
  }
  const int N = (int)L.n_frames;
  // The two signal rows hold at least PG_MIN_ROW_FRAMES frames: the ramp paths lay their per-frame parameter sequences out in `tmp`, eight to ten
  // sequences of at least eight frames (delay_ramp_fast, chorus_ramp_fast) — with rows sized by a launch of a handful of frames they would
  // decline, and the fast kernels have no serial code to fall back to (found by the fuzz over block sizes: 1-frame blocks behind a ramp).
  const int NA = N < PG_MIN_ROW_FRAMES ? PG_MIN_ROW_FRAMES : N;
  float* sig = (float*)pg_smem;
  float* tmp = sig + 2 * NA;
  char* scratch = (char*)(tmp + 2 * NA);
  // fixed small areas at the start of scratch
  PgVoice* lv = (PgVoice*)scratch;                 scratch += (sizeof(PgVoice) + 15) & ~15ull;
  PgFx* lfx0 = (PgFx*)scratch;                     scratch += (sizeof(PgFx) + 15) & ~15ull;   // the chain's effects alternate between two slots: the first
  PgFx* lfx1 = (PgFx*)scratch;                     scratch += (sizeof(PgFx) + 15) & ~15ull;   // two keep theirs (what a later block of the launch finds)
  int* ctl = (int*)scratch;                        scratch += 128;
  float* red = (float*)scratch;                    scratch += 64;
  int* ur = (int*)scratch;                         scratch += 128;  // copy of the unit record (sizeof(PgUnit) <= 128)
  if (!resident && tid < (int)(sizeof(PgUnit) / 4)) ur[tid] = (int)unit_w;     // (read behind the barrier of the deferral decision / the block's first barrier)
  SrcScratch S;
  src_carve(scratch, S);
  S.diag = L.diag;
  S.sched_rd = nullptr;
  FastCtx fc;
  fc.tmp = tmp; fc.tmp_floats = 2 * NA; fc.scratch = scratch; fc.ctl = ctl; fc.red = red; fc.diag = L.diag; fc.err = L.error_word;
  fc.idx_log = FAST_ONLY ? nullptr : L.index_log;  // (a constant in the fast kernels: the logging stores are compiled out)
  if (L.mode != 2) PG_STAMP(L.diag, 0);

  // ---- two-kernel protocol: the lean fast kernel defers units it cannot run to the generic kernel ----
  if (FAST_ONLY && resident) {
    // (decided at the launch's first block: a super-block launch carries no commands, and nothing between its blocks changes what decides)
  } else if (FAST_ONLY) {
    carry.resident = 0; carry.fx_valid = 0;
    if (u_staged && u_staged <= L.staged_on) return;  // rendered by the stage kernels of this round
    if (tid == 0) {
      // Ramps only start with a parameter command, and commands are always rendered (and the ramp state re-evaluated at the
      // end of the block) by the generic kernel: the unit record alone decides, no walk over the effect states.
      int ok = !(u_static_defer || u_maybe_ramping);
      for (int ci0 = 0; ok && ci0 < L.n_cmds; ++ci0) if (L.cmds[ci0].unit == u) ok = 0;  // parameter events: exact path
      unit.deferred = ok ? 0 : 1;
      if (!ok && L.n_chunks > 1) pg_raise_super_deferred(L);  // nobody renders the later blocks of this unit
      else if (!ok && L.defer_list) L.defer_list[atomicAdd(L.defer_count, 1)] = slot;
      ctl[5] = ok;
    }
    __syncthreads();
    if (!ctl[5]) return;
    // the later blocks of this launch find the records where this one leaves them: one voice at most (its LDS copy), two effects at most (their slots)
    carry.si = si; carry.sf = sf; carry.unit_w = unit_w;
    carry.resident = (tables && L.n_chunks > 1 && (si.w & 0xffffff) <= 1 && u_n_fx0 <= 2) ? 1 : 0;
  } else if (L.mode == 2) {
    if (tid == 0) { ctl[5] = unit.deferred; unit.deferred = 0; }
    __syncthreads();
    if (!ctl[5]) return;
  }
  // The first effects' state blocks (one qword per lane, ~1 KB each) are in flight since the head of the kernel (or requested here when the
  // launch carries no slot tables): the HBM round trips complete under the source stage; the words wait in registers until the chain stages
  // them in LDS.
  if (FAST_ONLY && !tables && u_n_fx0 > 0 && tid < n_fx_words / 2) fx0_word = ((const unsigned long long*)&L.fx[u_fx0])[tid];
  // (the fast kernels run ONE segment per block: the first two effects' states arrive in fx0_word / fx1_word; the generic kernel applies commands
  // to the global copy first and stages from there)
  const bool external = u_kind0 == UNIT_BUS || u_kind0 == UNIT_EFFECT;
  float* ext = L.bus + (size_t)slot * L.bus_unit_stride + (size_t)chunk * 2 * (size_t)N;  // (a bus launch behind a super-block: block c of the summed bus)
  if (external) {
    for (int i = tid; i < 2 * N; i += nt) sig[i] = ext[i];
  } else {
    for (int i = tid; i < 2 * N; i += nt) sig[i] = 0.0f;  // clear_buffer (mixed.rs:673)
  }
  // Where this piece sits: in the main mixer's chunk grid (pc), and — units with events of their own or of an ancestor inside the main chunk
  // (the generic kernel only) — where the unit's own chunk / its parent's call end before the main chunk does (CMD_CHUNK_END / CMD_CALL_END).
  const PgPiece pc = pg_piece(L, chunk);
  const uint64_t pos0 = L.pos + (uint64_t)chunk * (uint64_t)N;
  uint64_t chunk_end_mark = pc.chunk_end, call_end_mark = pc.chunk_end;
  int ci = 0;  // command cursor (commands are sorted by (unit, frame))
  while (ci < L.n_cmds && L.cmds[ci].unit < u) ++ci;
  if (!FAST_ONLY) {
    for (int cj = ci; cj < L.n_cmds && L.cmds[cj].unit == u; ++cj) {
      if ((int)L.cmds[cj].frame < N) continue;
      if (L.cmds[cj].type == CMD_CHUNK_END && L.cmds[cj].value64 < chunk_end_mark) chunk_end_mark = L.cmds[cj].value64;
      if (L.cmds[cj].type == CMD_CALL_END && L.cmds[cj].value64 < call_end_mark) call_end_mark = L.cmds[cj].value64;
    }
    if (call_end_mark < chunk_end_mark) chunk_end_mark = call_end_mark;  // (a call boundary ends the chunk as well)
  }
  if (tid == 0 && pc.first) {  // a new chunk of the main mixer: its sub-mixers' calls and segments count from here
    unit.call_idx = 0; unit.call_audible = 0; unit.seg_idx = -1; unit.chunk_any_audible = 0;
    PG_UL(call_idx) = 0; PG_UL(seg_idx) = -1; PG_UL(chunk_any_audible) = 0;
    ((unsigned long long*)&PG_UL(call_audible))[0] = 0ull;
  }
  __syncthreads();

  // ---- event-split loop of MixedSource::write (mixed.rs:679-712) for this unit ----
  int frame0 = 0;
  // nested sub-mixers: an ancestor that splits its chunk at events calls this unit once per segment (CMD_CALL_SPLIT marks the
  // boundaries); the silence gate and the `audible` result are per call. Only the generic kernel sees more than one call per piece.
  float* const out = external ? nullptr : L.unit_out + (size_t)chunk * L.chunk_stride + (size_t)slot * L.out_stride;
  int call_start = 0;
  bool cmd_at_0 = false;
  while (frame0 < N) {
    // apply all commands due at frame0 (process_events, event.rs:41-50)
    while (!FAST_ONLY && ci < L.n_cmds && L.cmds[ci].unit == u && (int)L.cmds[ci].frame <= frame0) {  // (the fast kernel defers units with commands)
      const PgCmd cmd = L.cmds[ci];
      if (frame0 == 0) cmd_at_0 = true;
      if (cmd.type == CMD_CALL_SPLIT) {
        if ((frame0 > call_start || PG_UL(call_frames) > 0) && PG_UL(kind) == UNIT_SUBMIXER && PG_UL(call_idx) < PG_MAX_CALLS - 1) {
          __syncthreads();
          const bool aud = submixer_call_piece(unit, &PG_UL(call_max), sig, out, call_start, frame0, true, L.sample_rate, (size_t)L.chunk_stride, (int)(L.out_stride / 2), ctl, red);
          if (tid == 0) {
            const unsigned long long m = ((unsigned long long*)&PG_UL(call_audible))[0] | (aud ? 1ull << PG_UL(call_idx) : 0ull);
            ((unsigned long long*)&PG_UL(call_audible))[0] = m; unit.call_audible = m;
            if (PG_UL(call_idx) == 0) unit.audible = aud ? 1 : 0;
            PG_UL(call_idx) += 1; unit.call_idx = PG_UL(call_idx);
          }
          __syncthreads();
          call_start = frame0;
        }
        ++ci;
        continue;
      }
      int flush = 0;
      __syncthreads();
      if (tid == 0) {
        if (cmd.type == CMD_FX_PARAM) flush = fx_apply_param(L.fx[cmd.target], cmd.param, cmd.value, cmd.value64);
        else if (cmd.type == CMD_VOICE_VOLUME) sm_set_target(L.voices[cmd.target].volume, cmd.value);
        else if (cmd.type == CMD_VOICE_PAN) sm_set_target(L.voices[cmd.target].panning, cmd.value);
        else if (cmd.type == CMD_VOICE_STOP) { L.voices[cmd.target].has_stop = 1; L.voices[cmd.target].stop_time = cmd.value64; }
        else if (cmd.type == CMD_VOICE_SPEED) voice_set_speed(&L.voices[cmd.target], __longlong_as_double((long long)cmd.value64), cmd.value);
        else if (cmd.type == CMD_VOICE_SEEK) voice_seek(&L.voices[cmd.target], __longlong_as_double((long long)cmd.value64));
        ctl[2] = flush;
      }
      __syncthreads();
      if (cmd.type == CMD_FX_RESET) fx_flush_wg(L.fx[cmd.target], 1);
      else if (ctl[2]) fx_flush_wg(L.fx[cmd.target], 0);
      __threadfence_block();
      ++ci;
    }
    int frame1 = N;
    if (ci < L.n_cmds && L.cmds[ci].unit == u && (int)L.cmds[ci].frame < N) frame1 = (int)L.cmds[ci].frame;
    const int seg = frame1 - frame0;
    float* sseg = sig + 2 * frame0;
    const uint64_t pos = pos0 + (uint64_t)frame0;
    // this segment within the unit's chunk (one MixedSource::write chunk = one call of every source, processor and sub-mixer under it):
    // a chunk begins with the main mixer's chunk and at every command of the unit; it ends where the next one begins
    const bool seg_first = frame0 > 0 || pc.first || cmd_at_0;
    const bool seg_last = frame1 < N || pos0 + (uint64_t)N >= chunk_end_mark;
    const uint64_t seg_chunk_end = frame1 < N ? pos0 + (uint64_t)frame1 : chunk_end_mark;
    if (seg_first) { __syncthreads(); if (tid == 0) { PG_UL(seg_idx) += 1; unit.seg_idx = PG_UL(seg_idx); } __syncthreads(); }
    bool audible_input;
    if (external) {
      // (the main mixer's chunk: the flag of the summed input sits in the word of the chunk's last piece)
      audible_input = (PG_UL(kind) == UNIT_EFFECT) ? true : (L.bus_audible ? (L.bus_audible[pc.c_last] != 0) : true);
    } else {
      audible_input = false;
      // (not in the four-per-CU kernel, whose registers are spoken for: the host sends graphs with nested mixers to the wide kernel instead)
      if (!(FAST_ONLY && KMASK == (0x7ff & ~((1 << 5) | (1 << 7)))) && PG_UL(n_children) > 0) {  // process_sub_mixers (mixed.rs:505-554): add_buffers per sub-mixer, in the order they were added
        const int k = PG_UL(seg_idx) < PG_MAX_CALLS - 1 ? PG_UL(seg_idx) : PG_MAX_CALLS - 1;
        for (int c = 0; c < PG_UL(n_children); ++c) {
          const int2 cr = L.child_rows[PG_UL(child_off) + c];
          const float* row = L.rows_base + (size_t)chunk * L.chunk_stride + (size_t)cr.x * L.out_stride + 2 * frame0;
          for (int i = tid; i < 2 * seg; i += nt) sseg[i] += row[i];
          const PgUnit& cu = L.units[cr.y];
          audible_input |= k == 0 ? cu.audible != 0 : ((cu.call_audible >> k) & 1ull) != 0;
        }
        __syncthreads();
      }
      // where the MixedSource::write call that this segment belongs to ends (PgVoice::zombie_end): the whole write for a source of the main
      // mixer; for a sub-mixer the parent's current chunk — up to its next call boundary (CMD_CALL_SPLIT, CMD_CALL_END) or the end of the main chunk
      uint64_t call_end_pos = L.call_end;
      if (PG_UL(kind) != UNIT_SOURCE) {
        call_end_pos = call_end_mark;
        if (!FAST_ONLY) for (int cj = ci; cj < L.n_cmds && L.cmds[cj].unit == u; ++cj) if (L.cmds[cj].type == CMD_CALL_SPLIT && (int)L.cmds[cj].frame > frame0 && (int)L.cmds[cj].frame < N) { call_end_pos = pos0 + (uint64_t)L.cmds[cj].frame; break; }
      }
      int later = 0;
      for (int vi = 0; vi < PG_UL(n_voices); ++vi) {
        PgVoice* gv = &L.voices[vi == 0 ? PG_UL(voice0) : L.voice_index[PG_UL(voice_off) + vi]];
        const int r = voice_process<!FAST_ONLY, (FAST_ONLY && KMASK == (0x7ff & ~((1 << 5) | (1 << 7)))) ? 1 : 2>(gv, lv, sseg, tmp, seg, pos, S, L.sched, L.sched_bank, tables && vi == 0, voice_word, call_end_pos, seg_first, seg_chunk_end,
                                                                                                                      resident && vi == 0);
        audible_input |= (r & 1) != 0;
        later |= r & 2;
      }
      if (PG_UL(kind) == UNIT_SOURCE) {  // (no chain: the unit's result is whether its source produced output anywhere in the chunk)
        if (audible_input && tid == 0) { unit.chunk_any_audible = 1; PG_UL(chunk_any_audible) = 1; }
      } else {
        // audible_input of the chunk (mixed.rs:696-706) is decided where the chunk begins: sub-mixers audible in it, sources that produced output
        // in this piece, sources that start in one of its later pieces; the later pieces take the decision from the unit record
        __syncthreads();
        if (seg_first) { if (tid == 0) { const int ai = (audible_input || later) ? 1 : 0; unit.chunk_audible_input = ai; PG_UL(chunk_audible_input) = ai; } }
        __syncthreads();
        audible_input = PG_UL(chunk_audible_input) != 0;
      }
    }
    PG_STAMP(L.diag, 1);
    // process_effects (mixed.rs:627-655)
    if (PG_UL(n_fx) > 0) {
      bool input_bypassed = !audible_input;
      if (!(PG_UL(effects_bypassed) && input_bypassed)) {
        bool all_bypassed = true;
        for (int fi = 0; fi < PG_UL(n_fx); ++fi) {
          // stage the effect's state block in LDS: the per-block bookkeeping of lane 0 (smoother checks, coefficient and
          // delay-length updates, ring positions) then costs LDS instead of HBM round trips; written back afterwards
          PgFx& gfx = (resident && fi < 2) ? L.fx[fi == 0 ? sf.x : sf.y] : L.fx[L.fx_index[PG_UL(fx_off) + fi]];   // (resident: no trip through the index table)
          PgFx* const lfx = (fi & 1) ? lfx1 : lfx0;
          __syncthreads();
          if (resident && fi < 2 && carry.fx_valid) { /* the slot holds what an earlier block of the launch left */ }
          else if (!resident && fi == 0 && FAST_ONLY) { if (tid < n_fx_words / 2) ((unsigned long long*)lfx)[tid] = fx0_word; }
          else if (!resident && fi == 1 && tables && sf.y >= 0) { if (tid < n_fx_words / 2) ((unsigned long long*)lfx)[tid] = fx1_word; }
          else for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];
          __syncthreads();
          PgFx& fx = *lfx;
          PG_STAMP(L.diag, 8);
          bool is_active;
          if (fx.standalone) {  // (plain Effect::process: every launch is one call)
            __syncthreads();
            if (tid == 0) fx_call_begin(fx);
            __syncthreads();
            fx_process_wg<FAST_ONLY, KMASK>(fx, sseg, seg * 2, fc, L.fast, true);
            if (tid == 0) fx.call_ramp = 0;
            is_active = true;
          }
          else is_active = fx_processor_process<FAST_ONLY, KMASK>(fx, sseg, seg * 2, input_bypassed, seg_first, seg_last, L.sample_rate, fc, L.fast, ctl, red);
          if (is_active) { input_bypassed = false; all_bypassed = false; }
          __syncthreads();
          for (int i = tid; i < (int)(sizeof(PgFx) / 4); i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
        }
        carry.fx_valid = 1;
        __syncthreads();
        // (the chain's result counts from the next chunk on: the later pieces of this one still see the flag the chunk began with)
        if (tid == 0 && seg_last) { unit.effects_bypassed = all_bypassed ? 1 : 0; PG_UL(effects_bypassed) = all_bypassed ? 1 : 0; }   // (the segment loop ends in a barrier)
      }
    }
    frame0 = frame1;
    __syncthreads();
  }

  if (!FAST_ONLY && tid == 0) {  // back in steady state? (decides whether the fast kernel may take the unit next block)
    int ramping = 0;
    for (int fi = 0; fi < PG_UL(n_fx); ++fi) ramping |= fx_fast_eligible(L.fx[L.fx_index[PG_UL(fx_off) + fi]], PG_UL(staged) != 0 || L.wide == 0) ? 0 : 1;  // (staged and lean kernels carry no ramp paths)
    for (int vi = 0; vi < PG_UL(n_voices); ++vi) {  // a pitch glide in progress is rendered here as well
      const PgVoice& vv = L.voices[L.voice_index[PG_UL(voice_off) + vi]];
      ramping |= (vv.current_speed != vv.target_speed) ? 1 : 0;
    }
    unit.maybe_ramping = ramping;
  }
  PG_STAMP(L.diag, 14);
  // ---- hand the block to the parent mixer ----
  if (external) {
    for (int i = tid; i < 2 * N; i += nt) ext[i] = sig[i];
    return;
  }
  if (PG_UL(kind) == UNIT_SUBMIXER) {
    const bool closes = pos0 + (uint64_t)N >= call_end_mark;
    const bool aud = submixer_call_piece(unit, &PG_UL(call_max), sig, out, call_start, N, closes, L.sample_rate, (size_t)L.chunk_stride, (int)(L.out_stride / 2), ctl, red);
    if (tid == 0) {
      unsigned long long m = ((unsigned long long*)&PG_UL(call_audible))[0];
      if (closes) {
        if (aud) m |= 1ull << PG_UL(call_idx);
        unit.call_audible = m;
        if (PG_UL(call_idx) == 0) unit.audible = aud ? 1 : 0;  // the first call; later calls of the main chunk (nested sub-mixers only) in call_audible
        unit.call_idx = PG_UL(call_idx) + 1;
      }
      // the main mixer reads one flag per chunk, in the word of the chunk's last piece (mixers of the main mixer are never split by an ancestor: one call)
      if (L.audible_tab) L.audible_tab[(size_t)chunk * L.audible_stride + slot] = (pc.last && m != 0) ? 1 : 0;
    }
  } else {
    for (int i = tid; i < 2 * N; i += nt) out[i] = sig[i];
    if (tid == 0) {
      const int any = PG_UL(chunk_any_audible) != 0 ? 1 : 0;
      if (pc.last) unit.audible = any;
      if (L.audible_tab) L.audible_tab[(size_t)chunk * L.audible_stride + slot] = (pc.last && any) ? 1 : 0;
    }
  }
  PG_STAMP(L.diag, 15);
  // schedule cache: representatives replay the next block's resampler schedule (piece = this launch's length, capped)
  if (L.sched && tid == 0) {
    const int piece = N < SRC_OUT_CAP ? N : SRC_OUT_CAP;
    for (int vi = 0; vi < PG_UL(n_voices); ++vi) sched_publish(&L.voices[L.voice_index[PG_UL(voice_off) + vi]], L.sched, L.sched_bank, piece);  // (single-block rounds only; read here, not kept in registers across the body)
  }
}

#undef PG_UF
#undef PG_UL
#ifndef PG_FAST_WAVES
#define PG_FAST_WAVES 2
#endif
// Fast-kernel variants by the effect kinds compiled in: the lean one (Gain, Panning, Reverb = the headline per-voice chain)
// keeps the hot loop free of spills; the wide one adds Filter, Eq5 and Distortion. The host picks by the kinds present.
#define PG_KMASK_LEAN ((1 << 0) | (1 << 1) | (1 << 5))
#define PG_KMASK_ALL 0x7ff  // bits 0..9: effect kinds; bit 10: the ramp paths (FilterEffect cutoff / Q)
#define PG_KMASK_GENERIC 0xfff  // ... bit 11: the generic kernel's lone workgroups (four reverb sub-chunks per trip: registers to spare, latency to hide)
#define PG_KMASK_GAINPAN ((1 << 0) | (1 << 1))
// leading effects of the wide staged kernel: every kind with a time-parallel path whose LDS needs fit stage 1's arena (no Chorus)
#define PG_KMASK_LEADING ((1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 9))
// Super-block launches (L.n_chunks > 1): the workgroup renders its unit's consecutive blocks one after the other; everything a block
// leaves behind (effect / voice / unit state) went to global memory and is read back by the same workgroup after a barrier.
__global__ void __launch_bounds__(256, PG_FAST_WAVES) pg_unit_kernel_fast(PgLaunch L) {
  PgUnitCarry carry;
  carry.resident = 0; carry.fx_valid = 0;
  pg_unit_body<true, PG_KMASK_LEAN>(L, (int)blockIdx.x, 0, carry);
  for (int c = 1; c < L.n_chunks; ++c) { __syncthreads(); pg_unit_body<true, PG_KMASK_LEAN>(L, (int)blockIdx.x, c, carry); }
}
__global__ void __launch_bounds__(256, PG_FAST_WAVES) pg_unit_kernel_fast_wide(PgLaunch L) {
  PgUnitCarry carry;
  carry.resident = 0; carry.fx_valid = 0;
  pg_unit_body<true, PG_KMASK_ALL>(L, (int)blockIdx.x, 0, carry);
  for (int c = 1; c < L.n_chunks; ++c) { __syncthreads(); pg_unit_body<true, PG_KMASK_ALL>(L, (int)blockIdx.x, c, carry); }
}
// Chains without Reverb and Compressor (C3: Filter -> Chorus per voice): those two carry the large register footprints and LDS arenas.
// Without them the same body compiles for four workgroups per CU (128 VGPRs) and its arena fits 40 KB.
#define PG_KMASK_MID (PG_KMASK_ALL & ~((1 << 5) | (1 << 7)))
static_assert(PG_KMASK_MID == (0x7ff & ~((1 << 5) | (1 << 7))), "pg_unit_body's test for the four-per-CU kernel");
#ifndef PG_MID_WAVES
#define PG_MID_WAVES 4
#endif
__global__ void __launch_bounds__(256, PG_MID_WAVES) pg_unit_kernel_fast_mid(PgLaunch L) {
  PgUnitCarry carry;
  carry.resident = 0; carry.fx_valid = 0;
  pg_unit_body<true, PG_KMASK_MID>(L, (int)blockIdx.x, 0, carry);
  for (int c = 1; c < L.n_chunks; ++c) { __syncthreads(); pg_unit_body<true, PG_KMASK_MID>(L, (int)blockIdx.x, c, carry); }
}
// The main mixer's effect chain behind a super-block launch as a PIPELINE over the blocks: workgroup f runs effect f of the chain over the
// summed blocks 0, 1, 2 ... in order and hands block c on to workgroup f + 1 through the bus buffer itself (in place) and a progress word —
// effect f works on block c while effect f + 1 works on block c - 1. A lone workgroup running the whole chain is a pure latency chain (C2: Eq5
// 50 K + Reverb 96 K cycles per block, tools/diag_bus.py); the chain's stages are independent state machines, so its time per block becomes
// the slowest effect's instead of their sum. Per block every effect takes EffectProcessor::process's decisions with the same inputs as in the
// serial order (mixed.rs:627-655): audible_input of the block (L.bus_audible[c]) and whether an earlier effect of the chain was active on
// it (travels with the progress word); MixedSource's shortcut `effects_bypassed && input_bypassed -> skip the chain` changes nothing an
// effect would not decide for itself (a bypassed processor with silent input stays bypassed, effect.rs:88-101) and is kept as state only.
// Workgroup f waits for workgroup f - 1 only. Residency: the launch has at most PG_BUS_PIPELINE_MAX (16) workgroups of 256 lanes, one per CU at
// most, on a device with 256 CUs and nothing else in front of them on this stream — all of them are resident at once, whatever order the
// dispatcher starts them in (the guide lists dispatch order as undefined: nothing here relies on it). Should a producer never publish
// (a fault, a preempted queue), the consumer gives up after a bounded number of polls, raises PG_DEVERR_BUS_STALLED and passes its blocks
// on unprocessed: a stuck stream would be invisible to the host, a raised flag disables the graph at the next write.
template <int KMASK>
__device__ __forceinline__ void pg_bus_pipeline(const PgLaunch& L) {
  const int f = (int)blockIdx.x, n_stages = (int)gridDim.x;
  const bool last_stage = f == n_stages - 1;
  PgUnit& unit = L.units[L.unit_base];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  const int NA = N < PG_MIN_ROW_FRAMES ? PG_MIN_ROW_FRAMES : N;
  float* sig = (float*)pg_smem;
  float* tmp = sig + 2 * NA;
  float* nxt = tmp + 2 * NA;   // the NEXT block's input, on its way global -> LDS while this block is processed (pg_launch_units adds the room in mode 3)
  char* scratch = (char*)(nxt + 2 * NA);
  scratch += (sizeof(PgVoice) + 15) & ~15ull;
  PgFx* lfx = (PgFx*)scratch;                      scratch += (sizeof(PgFx) + 15) & ~15ull;
  int* ctl = (int*)scratch;                        scratch += 128;
  float* red = (float*)scratch;                    scratch += 64;
  FastCtx fc;
  fc.tmp = tmp; fc.tmp_floats = 2 * NA; fc.scratch = scratch; fc.ctl = ctl; fc.red = red; fc.diag = L.diag; fc.err = L.error_word; fc.idx_log = nullptr;   // (diag: stamps of stage 0 in diagnostic builds, tools/diag_stamps.py bus)
  PgFx& gfx = L.fx[L.fx_index[unit.fx_off + f]];
  for (int i = tid; i < (int)(sizeof(PgFx) / 4); i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];  // the effect's state stays in LDS over all blocks
  __syncthreads();
  const int n_chunks = L.n_chunks > 1 ? L.n_chunks : 1;
  int any_active = 0;
  int prefetched = -1;   // the block whose input was requested into `nxt` (uniform)
  unsigned long long hist = 0, mask = 0;   // `an effect up to this one was active` per block: the last 24 blocks / all (<= 64) blocks of the launch
  if (tid == 0) ctl[7] = 0;
  for (int c = 0; c < n_chunks; ++c) {
    int active_before = 0;
    int next_ready = (f == 0 && c + 1 < n_chunks) ? 1 : 0;   // stage 0 reads the mixer sum: complete before this launch began
    if (f > 0 && !ctl[7]) {
      if (tid == 0) {
        unsigned long long w;
        unsigned polls = 0;
        bool ok;
        do {
          w = __hip_atomic_load(&L.bus_progress[f - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          ok = (uint32_t)(w >> 32) == L.round && (int)(w & 0xffull) > c;
          if (!ok) __builtin_amdgcn_s_sleep(8);   // (~0.2 us: the producer's block takes tens of microseconds)
        } while (!ok && ++polls < (1u << 24));     // (seconds: far beyond any block time)
        if (!ok) { pg_raise_device_error(L, PG_DEVERR_BUS_STALLED); ctl[7] = 1; }
        // `an earlier effect was active on THIS block`: the producer may be several blocks ahead, so the word carries the flags of its last 24
        // blocks (bit 8 = the latest); further back, the producer's per-block mask word (stored before the count was released)
        const int cnt = ok ? (int)(w & 0xffull) : 0, back = cnt - 1 - c;
        int act = 0;
        if (ok) act = back < 24 ? (int)((w >> (8 + back)) & 1ull)
                                : (int)((__hip_atomic_load(&L.bus_progress[PG_BUS_PIPELINE_MAX + f - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> c) & 1ull);
        ctl[6] = act;
        ctl[5] = cnt;   // blocks the producer has published
      }
      __syncthreads();
      __threadfence();  // the producer's stores of the published blocks are visible (the L1 is invalidated behind the acquire)
      active_before = ctl[6];
      next_ready = (ctl[5] > c + 1 && c + 1 < n_chunks) ? 1 : 0;
    }
    PG_STAMP(L.diag, 60);
    float* blk = L.bus + (size_t)c * 2 * (size_t)N;
    if (prefetched == c) {   // requested while the block before was processed: in LDS by now, or nearly
      lds_dma_wait();
      __syncthreads();
      float* t = sig; sig = nxt; nxt = t;
    } else {
      for (int i = tid; i < 2 * N; i += nt) sig[i] = __builtin_nontemporal_load(blk + i);
      __syncthreads();
    }
    // The next block's input: one dword per lane and trip, global -> LDS directly (no registers; nothing waits for it until the next trip
    // of this loop) — the load at the top of a block was a round trip on a workgroup whose block is a latency chain.
    if (next_ready) {
      const float* nb = blk + 2 * (size_t)N;
      for (int k = 0; k * 256 < 2 * N; ++k) { const int i = tid + k * 256; if (i < 2 * N) lds_dma_dword(nb + i, nxt + k * 256 + (tid & ~63)); }
      prefetched = c + 1;
    }
    PG_STAMP(L.diag, 61);
    // (per chunk of the main mixer: the flag of its summed input sits in the word of its last piece, the processor decides at its first)
    const PgPiece pc = pg_piece(L, c);
    const bool audible_input = L.bus_audible ? (L.bus_audible[pc.c_last] != 0) : true;
    const bool input_bypassed = !audible_input && !active_before;
    const bool is_active = fx_processor_process<false, KMASK>(*lfx, sig, 2 * N, input_bypassed, pc.first, pc.last, L.sample_rate, fc, L.fast, ctl, red);
    __syncthreads();
    PG_STAMP(L.diag, 62);
    if (is_active) for (int i = tid; i < 2 * N; i += nt) blk[i] = sig[i];
    any_active = (active_before || is_active) ? 1 : 0;
    PG_STAMP(L.diag, 63);
    hist = ((hist << 1) | (unsigned long long)any_active) & 0xffffffull;
    mask |= (unsigned long long)any_active << c;
    if (!last_stage) {   // (nobody reads the last stage's words: its stores are complete when the kernel ends)
      __threadfence();
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(&L.bus_progress[PG_BUS_PIPELINE_MAX + f], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ordered before the release below)
        __hip_atomic_store(&L.bus_progress[f], ((unsigned long long)L.round << 32) | (hist << 8) | (unsigned long long)(c + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (prefetched >= n_chunks) lds_dma_wait();   // (never: the last block requests nothing)
  __syncthreads();
  for (int i = tid; i < (int)(sizeof(PgFx) / 4); i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
  if (last_stage && tid == 0) unit.effects_bypassed = any_active ? 0 : 1;  // of the last block, as the serial order leaves it
}

// The generic kernel holds one workgroup per CU (its register footprint): the grid is capped at the CU count and every workgroup
// walks its share of the units, so the launch that finds nothing deferred costs 256 workgroup starts instead of n_units.
__global__ void __launch_bounds__(256) pg_unit_kernel(PgLaunch L) {
  if (L.mode == 3) { pg_bus_pipeline<PG_KMASK_GENERIC>(L); return; }
  if (L.mode == 2 && L.defer_list) {  // deferred units only: the compact list the fast kernels of this round appended to
    const int n = *L.defer_count;
    if (blockIdx.x == 0 && pg_tid() == 0) {
      *L.defer_reset = 0;
      // tell the host how many units this round deferred: after a round with none (and no change since) it skips this launch
      if (L.host_feedback) { *(volatile unsigned long long*)L.host_feedback = ((unsigned long long)L.round << 32) | (unsigned long long)(uint32_t)n; __threadfence_system(); }
    }
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
      PgUnitCarry carry;
      carry.resident = 0; carry.fx_valid = 0;
      pg_unit_body<false, PG_KMASK_GENERIC>(L, L.defer_list[i], 0, carry);
      __syncthreads();
    }
    return;
  }
  // n_chunks > 1 reaches this kernel only as the bus launch behind a super-block (the host renders steady-state graphs that way): the
  // chain runs over the summed blocks one after the other, every per-block decision (audible_input, bypass, tails) taken per block
  const int n_chunks = L.n_chunks > 1 ? L.n_chunks : 1;
  for (int slot = (int)blockIdx.x; slot < L.n_units; slot += (int)gridDim.x) {
    for (int c = 0; c < n_chunks; ++c) {
      PgUnitCarry carry;
      carry.resident = 0; carry.fx_valid = 0;
      pg_unit_body<false, PG_KMASK_GENERIC>(L, slot, c, carry);
      __syncthreads();
    }
  }
}

// ---- staged pipeline for [Gain|Panning]* -> Reverb units -------------------------------------------------------------------
// The fused fast kernel above is one function at 256 VGPRs / 55 KB LDS: two workgroups per CU, latency-bound. Here the same
// stage functions of the reverb are separate units of register allocation with an LDS plan of 38 KB, so four workgroups fit a CU:
//   stage 1: deferral decision, source stage, leading Gain/Panning effects, reverb bypass logic + front (predelay, biquad A)
//   stage 2: reverb mid (allpasses + vibrato lines)
//   stage 3: reverb tail (biquad B, asin, biquad C, dry mix), tail counters, sub-mixer silence gate, unit output
// Two drivers: pg_stage_fused_kernel runs the three stages back to back in one launch (chunk buffer and effect state stay in
// LDS, the dry signal is parked in the unit's output row during stage 2); pg_stage1/2/3_kernel are one launch per stage with
// the chunk buffer handed over through HBM/L2 (kept for profiling the stages in isolation).
// LDS plan (all stages): [PgFx][ctl 128][red 64][arena ...]; the arena starts with bufA in every stage.
//   stage 1: arena = union(source scratch, bufA .. xchg), then [sig][tmp][PgVoice]                        36.6 KB at 1024 frames
//   stage 2: single launch: bufA .. xchg (inside the union), [sig] where stage 1 left it, then anchors + rotation table   38.9 KB
//            (round 2: the table holds |j| <= 64 only, which makes room for the dry signal: it no longer travels to the unit's output
//            row and back — 16 B per voice-frame less HBM traffic and no reload at the top of stage 3);
//            per-stage launches: the plain reverb arena (bufA, records, anchors, rotation table)           30.4 KB
//   stage 3: single launch: bufA .. xchg, [sig] in place; per-stage launches: bufA .. xchg, then [sig]     27.9 KB
__device__ __forceinline__ int stage_image_doubles(int T) { return 2 * T + (T >> 3) + 2; }
__device__ __forceinline__ void stage_store_image(double* g, const double* lds, int T) {
  const int n2 = (stage_image_doubles(T) + 1) >> 1;
  for (int i = pg_tid(); i < n2; i += blockDim.x) ((double2*)g)[i] = ((const double2*)lds)[i];
}
__device__ __forceinline__ void stage_load_image(double* lds, const double* g, int T) {
  const int n2 = (stage_image_doubles(T) + 1) >> 1;
  for (int i = pg_tid(); i < n2; i += blockDim.x) ((double2*)lds)[i] = ((const double2*)g)[i];
}
constexpr int PG_STAGE_LEAD = 3;  // state blocks of effects in front of the reverb that stage 1 requests ahead (slot_lead)
constexpr size_t STAGE_ARENA_PREFIX = (size_t)REV_BUF_DOUBLES * 8 + 16 * sizeof(RevRec) + 16 * 8 + 13 * sizeof(RevDesc) + 4 * 8;  // bufA .. xchg
constexpr size_t STAGE_FIXED = ((sizeof(PgFx) + 15) & ~15ull) + 128 + 64;
constexpr size_t STAGE1_UNION = ((SRC_SCRATCH_BYTES > STAGE_ARENA_PREFIX ? SRC_SCRATCH_BYTES : STAGE_ARENA_PREFIX) + 15) & ~15ull;

struct StageLds { PgFx* lfx; int* ctl; float* red; char* arena; };
// `base`: the dynamic LDS of the kernel. An out-of-line stage function gets it as an argument: naming pg_smem there makes the
// compiler look the kernel's LDS offset up in a table in global memory — a dependent load at the top of the stage.
__device__ __forceinline__ StageLds stage_lds(char* base = pg_smem) {
  StageLds m;
  char* p = base;
  m.lfx = (PgFx*)p;   p += (sizeof(PgFx) + 15) & ~15ull;
  m.ctl = (int*)p;    p += 128;
  m.red = (float*)p;  p += 64;
  m.arena = p;
  return m;
}
__device__ __forceinline__ PgFx& stage_reverb(const PgLaunch& L, const PgUnit& unit) { return L.fx[L.fx_index[unit.fx_off + unit.n_fx - 1]]; }

// Stage 1. RESIDENT: the later stages run in the same launch (bufA and the effect state stay in LDS). Returns false when the
// unit was deferred to the generic kernel.
// Returns the unit's stage flags (PG_STAGE_*, also left in the unit record for the per-stage launches), or -1 when the unit was
// deferred to the generic kernel.
template <int TAG, bool RESIDENT>
__device__ __forceinline__ int stage1_run(const PgLaunch& L, int slot, const int4 si, const int chunk = 0, char* smem = pg_smem) {
  // `si` = L.slot_info[slot], loaded by the kernel: one load names the unit, its first voice and its reverb (and the unit's staged
  // level): their state blocks are then fetched side by side
  const int u = si.x;
  PgUnit& unit = L.units[u];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  float* out = L.unit_out + (size_t)chunk * L.chunk_stride + (size_t)slot * L.out_stride;
  const uint64_t pos0 = L.pos + (uint64_t)chunk * (uint64_t)N;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  PgFx& gfx = L.fx[si.z];
  uint32_t voice_word = 0;
  if ((si.w & 0xffffff) > 0 && tid < (int)(sizeof(PgVoice) / 4)) voice_word = ((const uint32_t*)&L.voices[si.y])[tid];
  unsigned long long fxr_word = 0;  // the reverb's state block (one qword per lane), used after the source stage
  if (tid < n_fx_words / 2) fxr_word = ((const unsigned long long*)&gfx)[tid];
  const StageLds m0 = stage_lds(smem);
  PgFx* lfx = m0.lfx; int* ctl = m0.ctl; float* red = m0.red;
  float* sig = (float*)(m0.arena + STAGE1_UNION);
  float* tmp = sig + 2 * N;
  PgVoice* lv = (PgVoice*)(tmp + 2 * N);
  // The effects in front of the reverb (wide kernel; C5: Filter -> Eq5 -> Delay): their state blocks used to be fetched one after the other,
  // each behind the index table and behind the write-back of the one before — two dependent trips through the loaded memory system per
  // effect on a workgroup whose stage 1 is a latency chain. The host names the first three in the slot table; they travel global -> LDS
  // directly (no registers, nothing waits here) while the source stage runs, into slots behind the voice record (pg_stage_lds_bytes).
  PgFx* const lead = (PgFx*)((char*)lv + ((sizeof(PgVoice) + 15) & ~15ull));
  int lead0 = -1, lead1 = -1, lead2 = -1;
  if (TAG == 3) {
    const int4 sl = L.slot_lead[slot];
    lead0 = __builtin_amdgcn_readfirstlane(sl.x); lead1 = __builtin_amdgcn_readfirstlane(sl.y); lead2 = __builtin_amdgcn_readfirstlane(sl.z);
#pragma unroll
    for (int k = 0; k < PG_STAGE_LEAD; ++k) {
      const int li = k == 0 ? lead0 : k == 1 ? lead1 : lead2;
      if (li < 0) continue;
      const float* src = (const float*)&L.fx[li];
      float* dst = (float*)(lead + k);
#pragma unroll
      for (int w = 0; w < n_fx_words; w += 256) if (w + tid < n_fx_words) lds_dma_dword(src + w + tid, dst + w + (tid & ~63));
    }
  }
  SrcScratch S;
  src_carve(m0.arena, S);
  S.diag = L.diag;
  S.sched_rd = nullptr;
  FastCtx fc;
  fc.tmp = tmp; fc.tmp_floats = 2 * N; fc.scratch = m0.arena; fc.ctl = ctl; fc.red = red; fc.diag = L.diag; fc.err = L.error_word; fc.idx_log = nullptr;
  PG_STAMP(L.diag, 0);
  // deferral decision: identical to the fused fast kernel
  if (tid == 0) {
    int ok = !(unit.static_defer || unit.maybe_ramping);
    for (int ci0 = 0; ok && ci0 < L.n_cmds; ++ci0) if (L.cmds[ci0].unit == u) ok = 0;
    unit.deferred = ok ? 0 : 1;
    if (!ok && L.n_chunks > 1) pg_raise_super_deferred(L);
    else if (!ok && L.defer_list) L.defer_list[atomicAdd(L.defer_count, 1)] = slot;
    ctl[5] = ok;
  }
  __syncthreads();
  if (!ctl[5]) { if (TAG == 3) lds_dma_wait(); return -1; }   // (nothing may still be on its way into LDS when the workgroup moves on)
  // the unit record is read once: every later `unit.x` would be another dependent trip to L2 on this workgroup's critical path
  const int n_voices = unit.n_voices, voice_off = unit.voice_off, n_fx = unit.n_fx, fx_off = unit.fx_off, effects_bypassed = unit.effects_bypassed;
  const int chunk_audible_input = unit.chunk_audible_input;
  for (int i = tid; i < 2 * N; i += nt) sig[i] = 0.0f;  // clear_buffer (mixed.rs:673)
  __syncthreads();
  // (staged units are sub-mixers of the main mixer without commands: their chunks are the main mixer's)
  const PgPiece pc = pg_piece(L, chunk);
  bool audible_input = false;
  int later = 0;
  for (int vi = 0; vi < n_voices; ++vi) {
    PgVoice* gv = &L.voices[vi == 0 ? si.y : L.voice_index[voice_off + vi]];
    const int r = voice_process<false, 0>(gv, lv, sig, tmp, N, pos0, S, L.sched, L.sched_bank, vi == 0, voice_word, pc.chunk_end, pc.first, pc.chunk_end);
    audible_input |= (r & 1) != 0;
    later |= r & 2;
  }
  // audible_input of the chunk: decided at its first piece (sources that wrote here or start in a later piece), kept for the others
  if (pc.first) { audible_input = audible_input || later != 0; if (tid == 0) unit.chunk_audible_input = audible_input ? 1 : 0; }
  else audible_input = chunk_audible_input != 0;
  PG_STAMP(L.diag, 1);
  if (TAG == 3) lds_dma_wait();   // the leading effects' state blocks (requested in front of the source stage: long there); the loop's first barrier publishes them
  int flags = audible_input ? PG_STAGE_AUDIBLE : 0;
  if (pc.first) flags |= PG_STAGE_FIRST;
  if (pc.last) flags |= PG_STAGE_LAST;
  bool input_bypassed = !audible_input;
  if (effects_bypassed && input_bypassed) flags |= PG_STAGE_SKIPPED;  // process_effects (mixed.rs:627-655)
  else {
    bool all_bypassed = true;
    for (int fi = 0; fi + 1 < n_fx; ++fi) {  // leading effects
      const int li = fi == 0 ? lead0 : fi == 1 ? lead1 : fi == 2 ? lead2 : -1;
      const bool pre = TAG == 3 && li >= 0;
      PgFx& g1 = L.fx[pre ? li : L.fx_index[fx_off + fi]];
      PgFx* const cfx = pre ? lead + fi : lfx;
      __syncthreads();
      if (!pre) for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)cfx)[i] = ((const uint32_t*)&g1)[i];
      __syncthreads();
      if (fi == 0) PG_STAMP(L.diag, 46);
      bool is_active;
      constexpr int KM = TAG == 3 ? PG_KMASK_LEADING : PG_KMASK_GAINPAN;
      if (cfx->standalone) {
        __syncthreads();
        if (tid == 0) fx_call_begin(*cfx);
        __syncthreads();
        fx_process_wg<true, KM>(*cfx, sig, N * 2, fc, L.fast, true);
        if (tid == 0) cfx->call_ramp = 0;
        is_active = true;
      }
      else is_active = fx_processor_process<true, KM>(*cfx, sig, N * 2, input_bypassed, pc.first, pc.last, L.sample_rate, fc, L.fast, ctl, red);
      if (is_active) { input_bypassed = false; all_bypassed = false; }
      __syncthreads();
      if (fi == 0) PG_STAMP(L.diag, 47);
      for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&g1)[i] = ((const uint32_t*)cfx)[i];
      PG_STAMP(L.diag, 40 + fi);
    }
    __syncthreads();
    if (tid < n_fx_words / 2) ((unsigned long long*)lfx)[tid] = fxr_word;
    __syncthreads();
    PG_STAMP(L.diag, 8);
    const bool active = lfx->standalone ? true : !fx_processor_pre(*lfx, input_bypassed, pc.first, ctl);
    if (active) {
      flags |= PG_STAGE_ACTIVE;
      const RevLds m = rev_lds(m0.arena);
      RevBlock b;
      (void)rev_block_params(*lfx, m, ctl, b);  // geometry was validated by the eligibility check (reverb_fast_eligible)
      PG_STAMP(L.diag, 11);
      rev_front(lfx->u.reverb, sig, N, m, b, L.diag);
      if (!RESIDENT) stage_store_image(L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, m.bufA, N);
    }
    if (input_bypassed) flags |= PG_STAGE_INPUT_BYPASSED;
    if (all_bypassed) flags |= PG_STAGE_ALL_BYPASSED;
    __syncthreads();
    if (!RESIDENT || !(flags & PG_STAGE_ACTIVE)) for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
  }
  if (!RESIDENT) for (int i = tid; i < 2 * N; i += nt) out[i] = sig[i];  // per-stage launches: the dry signal waits in the unit's output row
  if (!RESIDENT && tid == 0) unit.stage_flags = flags;  // (the single-launch kernels hand the flags over in registers)
  PG_STAMP(L.diag, 14);
  // schedule cache (ratio < 0.5 only): representatives replay the next block's resampler schedule. A single voice that took the
  // time-parallel schedule needs nothing published — decided from its LDS copy, without a trip to the voice table.
  if (L.sched && tid == 0 && !(n_voices == 1 && lv->sched_hit == 2)) {
    const int piece = N < SRC_OUT_CAP ? N : SRC_OUT_CAP;
    for (int vi = 0; vi < n_voices; ++vi) sched_publish(&L.voices[L.voice_index[voice_off + vi]], L.sched, L.sched_bank, piece);
  }
  PG_STAMP(L.diag, 13);
  return flags;
}

template <int TAG, bool RESIDENT>
__device__ __forceinline__ void stage2_run(const PgLaunch& L, int slot, int flags, char* smem = pg_smem) {
  if (!(flags & PG_STAGE_ACTIVE)) return;
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  PgUnit& unit = L.units[u];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  const StageLds m0 = stage_lds(smem);
  PgFx* lfx = m0.lfx;
  const RevLds m = rev_lds(m0.arena, RESIDENT ? m0.arena + STAGE1_UNION + (((size_t)N * 8 + 15) & ~15ull) : nullptr);  // (single launch: behind the dry signal)
  if (!RESIDENT) {
    PgFx& gfx = stage_reverb(L, unit);
    for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];
    stage_load_image(m.bufA, L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, N);
  }
  __syncthreads();
  rev_load_vtab(lfx->u.reverb, m);
  RevBlock b;
  (void)rev_block_params(*lfx, m, m0.ctl, b);
  rev_mid(lfx->u.reverb, N, m, b, m0.ctl, L.diag);
  __syncthreads();
  if (!RESIDENT) {
    PgFx& gfx = stage_reverb(L, unit);
    stage_store_image(L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, m.bufA, N);
    for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
  }
}

template <int TAG, bool RESIDENT>
__device__ __forceinline__ void stage3_run(const PgLaunch& L, int slot, int flags, char* smem = pg_smem, const int chunk = 0) {
  // one load names the unit and its reverb (as in stage 1); it is issued ahead of the dry-signal transfer below so that waiting
  // for it does not wait for the transfer (loads return in order)
  const int4 si = L.slot_info[slot];
  PgUnit& unit = L.units[si.x];
  PgFx& gfx = L.fx[si.z];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  float* out = L.unit_out + (size_t)chunk * L.chunk_stride + (size_t)slot * L.out_stride;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  const StageLds m0 = stage_lds(smem);
  PgFx* lfx = m0.lfx; int* ctl = m0.ctl; float* red = m0.red;
  float* sig = (float*)(m0.arena + (RESIDENT ? STAGE1_UNION : ((STAGE_ARENA_PREFIX + 15) & ~15ull)));  // single launch: where stage 1 left it
  __syncthreads();
  // Per-stage launches: the dry signal (stage 1 left it in the unit's output row) is only needed at the end of the tail. It travels
  // global -> LDS directly (lds_dma_dword: no registers, no wait here) while the two scans run; a dependent load at this point would
  // cost a full trip through the loaded memory system. Lane l of wave w, trip k: sample k * 256 + w * 64 + l.
  if (!RESIDENT) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      if (i < 2 * N) lds_dma_dword(out + i, sig + k * 256 + (tid & ~63));
    }
  }
  if (!(flags & PG_STAGE_SKIPPED)) {
    bool all_bypassed = (flags & PG_STAGE_ALL_BYPASSED) != 0;
    if (flags & PG_STAGE_ACTIVE) {
      const RevLds m = rev_lds(m0.arena);
      if (!RESIDENT) {
        for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];
        stage_load_image(m.bufA, L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, N);
      }
      __syncthreads();
      RevBlock b;
      (void)rev_block_params(*lfx, m, ctl, b);
      rev_tail_impl<!RESIDENT>(lfx->u.reverb, sig, N, m, b, L.diag);
      PG_STAMP(L.diag, 60);
      if (!lfx->standalone) fx_processor_post(*lfx, sig, N * 2, (flags & PG_STAGE_INPUT_BYPASSED) != 0, (flags & PG_STAGE_LAST) != 0, L.sample_rate, ctl, red);
      PG_STAMP(L.diag, 61);
      all_bypassed = false;
      __syncthreads();
      for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
    }
    __syncthreads();
    if (tid == 0 && (flags & PG_STAGE_LAST)) unit.effects_bypassed = all_bypassed ? 1 : 0;  // (counts from the next chunk on)
  }
  if (!RESIDENT) lds_dma_wait();  // (bypassed / skipped reverb: the dry signal is the output)
  __syncthreads();
  PG_STAMP(L.diag, 62);
  // ---- hand the block to the parent mixer: staged units are sub-mixers (SubMixerProcessor::process, submixer.rs:47-77), one call per chunk ----
  const bool closes = (flags & PG_STAGE_LAST) != 0;
  if (tid == 0) { ctl[8] = __float_as_int(unit.call_max); ctl[9] = (int)unit.call_frames; }
  __syncthreads();
  const bool aud = submixer_call_piece(unit, ctl + 8, sig, out, 0, N, closes, L.sample_rate, (size_t)L.chunk_stride, (int)(L.out_stride / 2), ctl, red);
  if (tid == 0) {
    if (closes) { unit.audible = aud ? 1 : 0; unit.call_audible = aud ? 1ull : 0ull; }
    if (L.audible_tab) L.audible_tab[(size_t)chunk * L.audible_stride + slot] = (closes && aud) ? 1 : 0;
  }
  PG_STAMP(L.diag, 15);
}

#ifndef PG_STAGE_WAVES
#define PG_STAGE_WAVES 4
#endif
// unit.staged: 0 no, 1 = leading effects are Gain / Panning only, 2 = any kind of PG_KMASK_LEADING. A launch renders the levels
// up to L.staged_on; LEVEL selects which of them this kernel takes.
template <int LEVEL>
__device__ __forceinline__ bool stage_unit_staged(const PgLaunch& L, int slot) {
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  return L.units[u].staged == LEVEL;
}
// the same from the slot-info word (the single-launch kernels start from that one load)
__device__ __forceinline__ int4 stage_slot_info(const PgLaunch& L, int slot) {
  int4 si = L.slot_info[slot];
  si.x = __builtin_amdgcn_readfirstlane(si.x); si.y = __builtin_amdgcn_readfirstlane(si.y);
  si.z = __builtin_amdgcn_readfirstlane(si.z); si.w = __builtin_amdgcn_readfirstlane(si.w);
  return si;
}
__device__ __forceinline__ int stage_unit_flags(const PgLaunch& L, int slot, bool& deferred) {
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  deferred = L.units[u].deferred != 0;
  return L.units[u].stage_flags;
}
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage1_kernel(PgLaunch L) {
  if ((int)blockIdx.x >= L.n_units || !stage_unit_staged<1>(L, blockIdx.x)) return;
  (void)stage1_run<1, false>(L, blockIdx.x, stage_slot_info(L, blockIdx.x));
}
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage2_kernel(PgLaunch L) {
  if ((int)blockIdx.x >= L.n_units || !stage_unit_staged<1>(L, blockIdx.x)) return;
  bool deferred; const int flags = stage_unit_flags(L, blockIdx.x, deferred);
  if (!deferred) stage2_run<1, false>(L, blockIdx.x, flags);
}
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage3_kernel(PgLaunch L) {
  if ((int)blockIdx.x >= L.n_units || !stage_unit_staged<1>(L, blockIdx.x)) return;
  bool deferred; const int flags = stage_unit_flags(L, blockIdx.x, deferred);
  if (!deferred) stage3_run<1, false>(L, blockIdx.x, flags);
}
// One launch per round; the workgroup runs the three stages of its unit's block back to back (chunk buffer and effect state stay in
// LDS). PG_STAGE_OUTLINE: bit 2 set = the tail stage is an out-of-line call (own register allocation); clear = inlined into the kernel
// function. A callee that needs more than the 80 caller-saved VGPRs saves the callee-saved ones it uses to scratch, and scratch is real
// HBM traffic at 1024 workgroups: fifteen registers = 15 MB written and 15 MB read back per 1024-voice block (-enable-ipra does not remove
// those saves). The tail once fitted the caller-saved set and its call was free; it no longer does (FETCH_SIZE / WRITE_SIZE showed
// 496 MB per block against 444 algorithmic), and inlined the kernel still allocates 125 VGPRs without a spill:
// all inline 0.0916 ms per headline block against 0.0939 with the call, same box, interleaved (C5: -0.6 %).
// The launch structure lives in LDS (written by one lane from the scalar registers the arguments arrive in) so that an
// out-of-line stage can take it by pointer. (The kernarg segment is not addressable from a callee: llvm.amdgcn.kernarg.segment.ptr
// lowers to NULL outside kernels.)
#ifndef PG_STAGE_OUTLINE
#define PG_STAGE_OUTLINE 0
#endif
typedef __attribute__((address_space(3))) char* PgLdsPtr;
#if PG_STAGE_OUTLINE & 4
static __device__ __noinline__ void stage3_call(const PgLaunch* L, int slot, int flags, PgLdsPtr smem, int chunk) { stage3_run<2, true>(*L, slot, flags, (char*)smem, chunk); }
#else
__device__ __forceinline__ void stage3_call(const PgLaunch* L, int slot, int flags, PgLdsPtr smem, int chunk) { stage3_run<2, true>(*L, slot, flags, (char*)smem, chunk); }
#endif
// The kernel's dynamic LDS as an opaque value: handed to the out-of-line stage as is, constant propagation would put the name
// pg_smem (and with it the offset-table lookup) back into the callee.
__device__ __forceinline__ PgLdsPtr stage_smem_arg() {
  uint32_t a = (uint32_t)(uintptr_t)(PgLdsPtr)pg_smem;
  asm volatile("" : "+s"(a));
  return (PgLdsPtr)(uintptr_t)a;
}
#ifdef PG_DIAG
#define PG_SLOT_STAMP(i) do { if (L.diag && threadIdx.x == 0 && slot < 4096) L.diag[64 + 4 * slot + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PG_SLOT_STAMP(i) do { } while (0)
#endif
// Super-block launches (L.n_chunks > 1): the workgroup renders the consecutive blocks of ITS unit one after the other. Units are
// independent until the mixer sum, so nothing synchronises the workgroups of a launch: they drift apart across the blocks, and the
// latency-bound source / front / tail stages of some run under the bandwidth-bound mid stage of others (with one block per launch
// all resident workgroups pass the stages in lock-step and HBM idles during the first and the last fifth of the kernel).
// A block leaves all of its state in global memory; the barrier between two blocks orders it before the next block's loads.
// (Per-lane values derive from pg_tid(), which keeps the stages' address arithmetic from being hoisted out of this loop.)
template <int LEVEL, int TAG>
__device__ __forceinline__ void stage_fused_body(const PgLaunch& L) {
  if ((int)blockIdx.x >= L.n_units) return;
  const int slot = blockIdx.x;
  const int4 si = stage_slot_info(L, slot);
  if ((si.w >> 24) != LEVEL) return;
  __shared__ PgLaunch sL;  // for the out-of-line stage; the barriers in front of that stage make it visible
  if (threadIdx.x == 0) sL = L;
  const int n_chunks = L.n_chunks > 1 ? L.n_chunks : 1;
#ifdef PG_STAGGER_US   // experiment: in a launch of ONE block every other workgroup starts PG_STAGGER_US microseconds late
  if (n_chunks == 1 && (blockIdx.x & 1)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(PG_STAGGER_US) * 100ull) __builtin_amdgcn_s_sleep(32);
  }
#endif
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    PG_SLOT_STAMP(0);
    const int flags = stage1_run<TAG, true>(L, slot, si, chunk);
    if (flags < 0) return;  // deferred to the generic kernel (never inside a super-block: the host launches those in steady state only)
    __syncthreads();
    PG_SLOT_STAMP(1);
    stage2_run<TAG, true>(L, slot, flags);
    __syncthreads();
    PG_SLOT_STAMP(2);
    stage3_call(&sL, slot, flags, stage_smem_arg(), chunk);
    PG_SLOT_STAMP(3);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage_fused_kernel(PgLaunch L) { stage_fused_body<1, 2>(L); }
// The same single launch for reverb units whose leading effects go beyond Gain / Panning (Filter, Eq5, Delay, Distortion:
// C5's per-voice Filter -> Eq5 -> Delay -> Reverb). A kernel of its own so that the lean one keeps its register allocation.
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage_fused_wide_kernel(PgLaunch L) { stage_fused_body<2, 3>(L); }

// ---- mixer-graph sum -------------------------------------------------------------------------------------
// last float4 of a block with an odd frame count: only the samples that exist (the caller's buffer ends there)
__device__ __forceinline__ void mix_store(float* bus, int s4, int n_samples, const float4& acc) {
  if (s4 * 4 + 4 <= n_samples) { *(float4*)(bus + (size_t)s4 * 4) = acc; return; }
  const float v[4] = {acc.x, acc.y, acc.z, acc.w};
  for (int i = 0; i < 4; ++i) if (s4 * 4 + i < n_samples) bus[(size_t)s4 * 4 + i] = v[i];
}
// Stage 1: partial[g][s] = sum over the units of group g (in unit order) of unit_out[u][s].
// Stage 2: bus[s] = sum over groups (in order) of partial[g][s]; audible = OR over units.
// Lanes run over the sample index s (coalesced float4); the f32 sum order is fixed (deterministic).
__global__ void __launch_bounds__(64) pg_mix_kernel_1(const float* __restrict__ unit_out, uint32_t stride, int n_units, int group, float* __restrict__ partial,
                                                      int n_vec4) {
  int s4 = blockIdx.x * blockDim.x + pg_tid();
  int g = blockIdx.y;
  if (s4 >= n_vec4) return;
  int u0 = g * group, u1 = u0 + group;
  if (u1 > n_units) u1 = n_units;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int u = u0; u < u1; ++u) {
    float4 v = *(const float4*)(unit_out + (size_t)u * stride + (size_t)s4 * 4);
    acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
  }
  *(float4*)(partial + (size_t)g * stride + (size_t)s4 * 4) = acc;
}
__global__ void __launch_bounds__(64) pg_mix_kernel_2(const float* __restrict__ partial, uint32_t stride, int n_groups, float* __restrict__ bus, int n_samples,
                                                      const int32_t* __restrict__ audible_row, int n_units,
                                                      int* __restrict__ audible_out) {
  int s4 = blockIdx.x * blockDim.x + pg_tid();
  if (blockIdx.x == gridDim.x - 1 && audible_out) {  // the extra last block: OR of the units' audible flags of this block (wave reduction)
    int a = 0;
    for (int u = pg_tid(); u < n_units; u += 64) a |= audible_row[u];
    for (int off = 32; off > 0; off >>= 1) a |= __shfl_xor(a, off, 64);
    if (pg_tid() == 0) *audible_out = a;
    return;
  }
  const int n_vec4 = (n_samples + 3) / 4;
  if (s4 >= n_vec4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int g = 0; g < n_groups; ++g) {
    float4 v = *(const float4*)(partial + (size_t)g * stride + (size_t)s4 * 4);
    acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
  }
  mix_store(bus, s4, n_samples, acc);
}

// Both stages in one launch (the usual case, <= 4096 units): a workgroup of 256 lanes owns PG_MIX_COLS float4 columns; lane
// (sub, col) sums the 16 units of group sub, sub + 64, ... for its column — the same partial sums, in the same order, as
// pg_mix_kernel_1 — into LDS, then the lanes of sub 0 add the groups' partials in order as pg_mix_kernel_2 does. Bit-identical to
// the two-launch path; one launch gap and one trip of the partials through HBM less.
#define PG_MIX_MAX_GROUPS 256
#define PG_MIX_COLS 4
__global__ void __launch_bounds__(256) pg_mix_kernel(const float* __restrict__ unit_out, uint32_t stride, int n_units, int n_groups, float* __restrict__ bus,
                                                      int n_samples, const int32_t* __restrict__ audible_tab, size_t audible_stride, int* __restrict__ audible_out,
                                                      size_t chunk_stride) {
  // super-block launches: blockIdx.y = block of the super-block (its unit rows chunk_stride floats further, its bus n_samples further)
  unit_out += (size_t)blockIdx.y * chunk_stride;
  bus += (size_t)blockIdx.y * (size_t)n_samples;
  const int n_vec4 = (n_samples + 3) / 4;
  __shared__ float4 part[PG_MIX_MAX_GROUPS][PG_MIX_COLS];
  const int t = pg_tid();
  if (blockIdx.x == gridDim.x - 1) {  // the extra last block: OR of the units' audible flags of block blockIdx.y -> audible_out[blockIdx.y]
    if (!audible_out) return;
    const int32_t* row = audible_tab + (size_t)blockIdx.y * audible_stride;
    int a = 0;
#pragma unroll 4
    for (int u = t; u < n_units; u += 256) a |= row[u];
    a = __syncthreads_or(a);
    if (t == 0) audible_out[blockIdx.y] = a;
    return;
  }
  const int col = t & (PG_MIX_COLS - 1), sub = t / PG_MIX_COLS;  // 64 sub-groups
  const int s4 = blockIdx.x * PG_MIX_COLS + col;
  if (s4 < n_vec4) {
    for (int g = sub; g < n_groups; g += 64) {
      const int u0 = g * 16;
      const int u1 = u0 + 16 < n_units ? u0 + 16 : n_units;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
      for (int u = u0; u < u1; ++u) {
        const float4 v = *(const float4*)(unit_out + (size_t)u * stride + (size_t)s4 * 4);
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
      }
      part[g][col] = acc;
    }
  }
  __syncthreads();
  if (sub == 0 && s4 < n_vec4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < n_groups; ++g) {
      const float4 v = part[g][col];
      acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
    }
    mix_store(bus, s4, n_samples, acc);
  }
}

// ---- host-callable launchers (C++ linkage, used by pg_host.cpp) ------------------------------------------
// LDS arena the time-parallel paths of the effect kinds in `kind_mask` (bit k = pg_effect_kind k) carve up — what a fast-kernel launch
// needs besides the signal rows. Without a Reverb (and without a Compressor) a unit kernel fits three workgroups per CU instead of two.
size_t pg_fast_scratch_bytes(uint32_t kind_mask) {
  size_t need = SRC_SCRATCH_BYTES;
  auto up = [&](size_t b) { if (b > need) need = b; };
  if (kind_mask & (1u << 5)) up(FAST_SCRATCH_BYTES);
  if (kind_mask & (1u << 7)) up(FAST_SCRATCH_COMP_BYTES);
  if (kind_mask & (1u << 6)) up(FAST_SCRATCH_CHORUS_BYTES);
  if (kind_mask & ((1u << 0) | (1u << 2) | (1u << 3) | (1u << 4))) up(FAST_SCRATCH_SCAN_BYTES);
  if (kind_mask & (1u << 8)) up(FAST_SCRATCH_GATE_BYTES);
  return need;
}
size_t pg_unit_lds_bytes(uint32_t n_frames, size_t scratch_bytes) {
  if (n_frames < PG_MIN_ROW_FRAMES) n_frames = PG_MIN_ROW_FRAMES;
  static_assert(sizeof(PgUnit) <= 128, "pg_unit_body keeps a copy of the unit record in 128 bytes of LDS");
  size_t fixed = ((sizeof(PgVoice) + 15) & ~15ull) + 2 * ((sizeof(PgFx) + 15) & ~15ull) + 128 + 64 + 128;
  size_t scratch = pg_fast_scratch_bytes(0xffffffffu);  // the full arena: the largest any effect kind carves up
  if (scratch_bytes && scratch_bytes < scratch) scratch = scratch_bytes < SRC_SCRATCH_BYTES ? SRC_SCRATCH_BYTES : scratch_bytes;
  return (size_t)n_frames * 16 + fixed + ((scratch + 15) & ~15ull);
}
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames, bool wide);
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames) { return pg_stage_lds_bytes(stage, n_frames, false); }
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames, bool wide) {
  const size_t s1 = STAGE_FIXED + STAGE1_UNION + (size_t)n_frames * 16 + ((sizeof(PgVoice) + 15) & ~15ull) + (wide ? PG_STAGE_LEAD * sizeof(PgFx) : 0);  // (wide kernel: + the leading effects' state slots)
  const size_t s2 = STAGE_FIXED + ((FAST_SCRATCH_BYTES + 15) & ~15ull);
  const size_t s3 = STAGE_FIXED + ((STAGE_ARENA_PREFIX + 15) & ~15ull) + (size_t)n_frames * 8;
  if (stage == 1) return s1;
  if (stage == 2) return s2;
  if (stage == 3) return s3;
  const size_t s2r = STAGE_FIXED + STAGE1_UNION + (((size_t)n_frames * 8 + 15) & ~15ull) + REV_TABLES_BYTES;  // stage 0, the single launch: the dry signal stays
  return s1 > s2r ? s1 : s2r;
}
// The staged pipeline of one round (units flagged `staged`): single_launch = pg_stage_fused_kernel, else three launches
// (L.stage_buf must then hold n_units rows).
// The kernels' dynamic-LDS limits are a per-device function attribute: set once for every device the library launches on (graphs of
// different devices, handles used from different threads).
static hipError_t pg_ensure_func_attributes() {
  static std::mutex mtx;
  static bool done[64] = {false};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mtx);
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  const void* fns[] = {(const void*)pg_stage1_kernel, (const void*)pg_stage2_kernel, (const void*)pg_stage3_kernel, (const void*)pg_unit_kernel,
                       (const void*)pg_unit_kernel_fast, (const void*)pg_unit_kernel_fast_wide, (const void*)pg_unit_kernel_fast_mid};
  for (const void* f : fns) if ((e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)pg_stage_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)pg_stage_fused_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)) != hipSuccess) return e;
  if (dev >= 0 && dev < 64) done[dev] = true;
  return hipSuccess;
}
hipError_t pg_launch_stages(const PgLaunch& L, hipStream_t stream, int single_launch, int lean, int wide, hipEvent_t ev0, hipEvent_t ev1) {
  if (L.n_units <= 0) return hipSuccess;
  { hipError_t e = pg_ensure_func_attributes(); if (e != hipSuccess) return e; }
  if (single_launch) {
    // ev0/ev1 (only passed when exactly one of the two launches happens): start / stop timestamps taken from the dispatch itself —
    // no marker packets in the stream, which cost ~7 us per round with hipEventRecord
    if (lean) hipExtLaunchKernelGGL(pg_stage_fused_kernel, dim3(L.n_units), dim3(256), (uint32_t)pg_stage_lds_bytes(0, L.n_frames), stream, ev0, ev1, 0, L);
    if (wide) hipExtLaunchKernelGGL(pg_stage_fused_wide_kernel, dim3(L.n_units), dim3(256), (uint32_t)pg_stage_lds_bytes(0, L.n_frames, true), stream, lean ? nullptr : ev0, lean ? nullptr : ev1, 0, L);
  } else {
    hipLaunchKernelGGL(pg_stage1_kernel, dim3(L.n_units), dim3(256), pg_stage_lds_bytes(1, L.n_frames), stream, L);
    hipLaunchKernelGGL(pg_stage2_kernel, dim3(L.n_units), dim3(256), pg_stage_lds_bytes(2, L.n_frames), stream, L);
    hipLaunchKernelGGL(pg_stage3_kernel, dim3(L.n_units), dim3(256), pg_stage_lds_bytes(3, L.n_frames), stream, L);
  }
  return hipGetLastError();
}
hipError_t pg_launch_units(const PgLaunch& L, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  if (L.n_units <= 0) return hipSuccess;
  size_t lds = pg_unit_lds_bytes(L.n_frames, L.mode == 1 ? L.fast_scratch_bytes : 0);  // (the generic kernel renders any chain: full arena)
  if (L.mode == 3) lds += (size_t)(L.n_frames < PG_MIN_ROW_FRAMES ? PG_MIN_ROW_FRAMES : L.n_frames) * 8;   // pg_bus_pipeline: the next block's input buffer
  { hipError_t e = pg_ensure_func_attributes(); if (e != hipSuccess) return e; }
  if (L.mode == 1 && L.wide == 2) hipExtLaunchKernelGGL(pg_unit_kernel_fast_mid, dim3(L.n_units), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);
  else if (L.mode == 1 && L.wide) hipExtLaunchKernelGGL(pg_unit_kernel_fast_wide, dim3(L.n_units), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);
  else if (L.mode == 1) hipExtLaunchKernelGGL(pg_unit_kernel_fast, dim3(L.n_units), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);
  else hipExtLaunchKernelGGL(pg_unit_kernel, dim3(L.n_units < 256 ? L.n_units : 256), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);  // (mode 3: n_units = stages of the bus pipeline)
  return hipGetLastError();
}
hipError_t pg_launch_mix(const float* unit_out, uint32_t stride, int n_units, float* partial, float* bus, uint32_t n_samples, const int32_t* audible_tab,
                         size_t audible_stride, int* audible_out, hipStream_t stream, int n_chunks, size_t chunk_stride) {
  int n_vec4 = (int)((n_samples + 3) / 4);
  int group = 16;
  int n_groups = (n_units + group - 1) / group;
  if (n_groups < 1) n_groups = 1;
  if (n_chunks < 1) n_chunks = 1;
  if (n_groups <= PG_MIX_MAX_GROUPS) {
    hipLaunchKernelGGL(pg_mix_kernel, dim3((n_vec4 + PG_MIX_COLS - 1) / PG_MIX_COLS + 1, n_chunks), dim3(256), 0, stream, unit_out, stride, n_units, n_groups, bus, (int)n_samples, audible_tab,
                       audible_stride, audible_out, chunk_stride);
    return hipGetLastError();
  }
  dim3 b(64), g1((n_vec4 + 63) / 64, n_groups), g2((n_vec4 + 63) / 64 + 1);  // +1: the flag-reduction block
  for (int c = 0; c < n_chunks; ++c) {  // (the partials buffer holds one block: the launches of consecutive blocks follow each other in stream order)
    hipLaunchKernelGGL(pg_mix_kernel_1, g1, b, 0, stream, unit_out + (size_t)c * chunk_stride, stride, n_units, group, partial, n_vec4);
    hipLaunchKernelGGL(pg_mix_kernel_2, g2, b, 0, stream, partial, stride, n_groups, bus + (size_t)c * n_samples, (int)n_samples, audible_tab + (size_t)c * audible_stride, n_units, audible_out ? audible_out + c : nullptr);
  }
  return hipGetLastError();
}
