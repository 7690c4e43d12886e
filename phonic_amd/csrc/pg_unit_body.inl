// Shared device code of the unit kernels (included by every pg_k_*.hip and by pg_kernels.hip): helpers, EffectProcessor / SubMixerProcessor
// logic and pg_unit_body, the body of the fast and the generic unit kernels. One kernel per translation unit (pg_k_*.hip): the kernels are
// independent units of register allocation and compile side by side (pg_kernels.hip as one file took five minutes).
#pragma once
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
  // (L.pad_chunks: a pre-scanned round — pg_defer_scan_kernel took the deferral decision for every unit of the level and left it in `deferred`; the
  // generic kernel runs BESIDE this one and may already be rewriting maybe_ramping of the units it renders: the decision word is what counts)
  const int u_static_defer = PG_UF(static_defer), u_maybe_ramping = L.pad_chunks ? PG_UF(deferred) : PG_UF(maybe_ramping), u_fx0 = PG_UF(fx0), u_staged = PG_UF(staged), u_n_fx0 = PG_UF(n_fx), u_kind0 = PG_UF(kind);
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
      unit.deferred = ok ? 0 : 1;   // (a pre-scanned round: the value the scan kernel left)
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
    if (tid == 0) { ctl[5] = unit.deferred; if (!L.pad_chunks) unit.deferred = 0; }   // (pad_chunks: a pre-scanned round — the fast kernels beside this one still read the word)
    __syncthreads();
    if (!ctl[5]) return;
    PG_STAMP(L.diag, 57);
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
  if (L.mode == 2) PG_STAMP(L.diag, 58);
  int frame0 = 0;
  // nested sub-mixers: an ancestor that splits its chunk at events calls this unit once per segment (CMD_CALL_SPLIT marks the
  // boundaries); the silence gate and the `audible` result are per call. Only the generic kernel sees more than one call per piece.
  float* const out = external ? nullptr : L.unit_out + (size_t)chunk * L.chunk_stride + (size_t)slot * L.out_stride;
  int call_start = 0;
  bool cmd_at_0 = false;
  int chain_ramping = -1;   // generic kernel, lane 0: does an effect of the chain still ramp? — from the LDS copies the last segment's chain left (-1: that segment did not run it)
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
    if (!FAST_ONLY) chain_ramping = -1;
    if (PG_UL(n_fx) > 0) {
      bool input_bypassed = !audible_input;
      if (!(PG_UL(effects_bypassed) && input_bypassed)) {
        bool all_bypassed = true;
        if (!FAST_ONLY) chain_ramping = 0;
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
          // (the steady-state test below, on the state as it goes back to global memory: read here it costs LDS trips — twelve ring positions of a
          // reverb from global memory were ~15 K cycles at the end of a commanded unit's block, tools/diag_cmd.py)
          if (!FAST_ONLY && tid == 0) chain_ramping |= fx_fast_eligible(fx, PG_UL(staged) != 0 || L.wide == 0) ? 0 : 1;
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
    if (chain_ramping >= 0) ramping = chain_ramping;   // the block's last segment ran the chain: decided there, on the LDS copies
    else for (int fi = 0; fi < PG_UL(n_fx); ++fi) ramping |= fx_fast_eligible(L.fx[L.fx_index[PG_UL(fx_off) + fi]], PG_UL(staged) != 0 || L.wide == 0) ? 0 : 1;  // (staged and lean kernels carry no ramp paths)
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
// Chains without Reverb and Compressor (C3: Filter -> Chorus per voice): those two carry the large register footprints and LDS arenas.
// Without them the same body compiles for four workgroups per CU (128 VGPRs) and its arena fits 40 KB.
#define PG_KMASK_MID (PG_KMASK_ALL & ~((1 << 5) | (1 << 7)))
static_assert(PG_KMASK_MID == (0x7ff & ~((1 << 5) | (1 << 7))), "pg_unit_body's test for the four-per-CU kernel");
#ifndef PG_MID_WAVES
#define PG_MID_WAVES 4
#endif
