// Time-parallel steady-state evaluation of effects by the whole workgroup (signal in LDS).
//
// fx_fast_process() returns false when the effect (in its current state: ramping parameters, short feedback
// lag, unsupported mode) must take the exact serial path of pg_fx_serial.h instead.
#pragma once
#include "pg_dsp_dev.h"
#include "pg_fx_serial.h"

namespace pgd {

struct FastCtx {
  float* tmp;        // LDS: 2*n_frames floats, free while effects run
  int tmp_floats;
  char* scratch;     // LDS arena (FAST_SCRATCH_BYTES)
  int* ctl;          // LDS: 32 ints for uniform decisions
  float* red;        // LDS: 16 floats for reductions
  unsigned long long* diag;
  int32_t* err;      // device word of sticky consistency flags (PgLaunch::error_word), may be nullptr
  int32_t* idx_log;  // test hook (PgLaunch::index_log): the floor()-derived read indices of the delay lines, one slot per (frame[, line], channel)
                     // of the current process call; nullptr (a compile-time constant in the hot kernels: the stores fold away) = not collected
};

constexpr size_t FAST_SCRATCH_BYTES = (2 * 1024 + 128 + 8) * 8 + 16 * 40 + 16 * 8 + 13 * 16 + 4 * 8 + 9 * 16 * 2 * 8 + 8 * 65 * 2 * 8 + 64;  // reverb: f64 chunk buffer + phase records + epilogue gets
// What the other time-parallel paths carve from the arena: a launch whose units hold no Reverb gets a smaller arena (pg_fast_scratch_bytes)
// and with it more resident workgroups per CU.
constexpr size_t FAST_SCRATCH_SCAN_BYTES = (2 * 1024 + 128 + 8) * 8 + 512;                           // Filter / Eq5 / Delay / Gain's DC filter: chunk buffer + scan hand-over + coefficients
constexpr size_t FAST_SCRATCH_CHORUS_BYTES = (2 * 1024 + 128 + 8) * 8 + 32 + 80 + 2 * 56 * 16 + 32 + 64;  // + the LFOs' phase pieces
constexpr size_t FAST_SCRATCH_COMP_BYTES = (2 * 4096 + 1024 + 8) * 4;                                  // Compressor: peak history + block
constexpr size_t FAST_SCRATCH_GATE_BYTES = 2 * 1024 * 4 + 256;

#include "pg_reverb_fast.inl"
#include "pg_delay_fast.inl"
#include "pg_reverb_ramp.inl"
#include "pg_chorus_fast.inl"
#include "pg_comp_fast.inl"
#include "pg_gate_fast.inl"

// FilterEffect (filter.rs:193-200) and Eq5Effect (eq5.rs:297-326) in steady state: cascaded TPT-SVF biquads on the f32
// signal, each stage a blocked scan over the block (see rev_biquad_scan_t). The block is staged as f64 in the skewed
// scratch buffer; `as f32` between stages is reproduced by rounding every stage's output through f32.
DEVO void biquad_chain_fast(float* sig, int n_samples, const PgBiquadCoef* coefs, PgState2* st /*[ch][n_stages]*/, int n_stages, int st_stride,
                            FastCtx& fc) {
  const int tid = pg_tid(), nt = blockDim.x;
  double* buf = (double*)fc.scratch;
  double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);   // two hand-over slots, used alternately: consecutive scans need no barrier between them
  PgState2* lst = (PgState2*)(xchg + 8);  // [n_stages][2] state view for the scans (n_stages <= 5)
  const int frames = n_samples / 2;
  for (int done = 0; done < frames; done += 1024) {
    const int T = frames - done < 1024 ? frames - done : 1024;
    __syncthreads();
    if (n_stages == 1) PG_STAMP(fc.diag, 43);
    for (int s = tid; s < 2 * T; s += nt) buf[REV_IDX(s >> 1, s & 1)] = (double)sig[2 * done + s];
    if (tid < 2 * n_stages) lst[tid] = st[(tid & 1) * st_stride + (tid >> 1)];
    __syncthreads();
    if (n_stages == 1) PG_STAMP(fc.diag, 44);
    for (int k = 0; k < n_stages; ++k) rev_biquad_scan_t<true>(coefs[k], lst + 2 * k, buf, T, xchg + 4 * (k & 1));  // (a lane's pass 1 reads what its own pass 2 wrote)
    __syncthreads();
    if (n_stages == 1) PG_STAMP(fc.diag, 45);
    if (tid < 2 * n_stages) st[(tid & 1) * st_stride + (tid >> 1)] = lst[tid];
    for (int s = tid; s < 2 * T; s += nt) sig[2 * done + s] = (float)buf[REV_IDX(s >> 1, s & 1)];
  }
  __syncthreads();
}
// FilterEffect while cutoff and / or Q ramp (filter.rs:166-192): the two smoothers' f32 value sequences are laid out by one lane (the
// same sm_next calls as the serial loop: exact, including the frame at which a ramp ends), every frame's coefficients are
// recomputed from them as biquad_apply does, and the recurrence runs as the time-varying blocked scan.
DEVO void filter_ramp_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  PgFilter& f = fx.u.filter;
  const int tid = pg_tid(), nt = blockDim.x;
  double* buf = (double*)fc.scratch;
  double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);
  float* cut = fc.tmp;            // [T] cutoff per frame (already clamped), then [T] q per frame
  const int frames = n_samples / 2;
  const int piece = fc.tmp_floats / 2 < 1024 ? fc.tmp_floats / 2 : 1024;
  const float nyq = (float)fx.sample_rate / 2.0f;
  const int btype = filter_to_biquad(f.type);
  const uint32_t sr = fx.sample_rate;
  for (int done = 0; done < frames; done += piece) {
    const int T = frames - done < piece ? frames - done : piece;
    float* qv = cut + T;
    __syncthreads();
    // (the two smoothers do not interact: one lane each, in different waves; the clamp of the cutoff follows on the same lane)
    if (tid == 0) { sm_sequence(f.cutoff, cut, T); for (int k = 0; k < T; ++k) cut[k] = clampf(cut[k], 20.0f, nyq); }
    else if (tid == 64) sm_sequence(f.q, qv, T);
    for (int s = tid; s < 2 * T; s += nt) buf[REV_IDX(s >> 1, s & 1)] = (double)sig[2 * done + s];
    __syncthreads();
    auto coef = [&](int n, double& a1, double& a2, double& a3, double& m0, double& m1, double& m2) {
      PgBiquadCoef c;
      c.type = btype; c.sample_rate = sr; c.cutoff = cut[n]; c.q = qv[n]; c.gain = 0.0f;
      c.a1 = 0.0; c.a2 = 0.0; c.a3 = 0.0; c.m0 = 0.0; c.m1 = 0.0; c.m2 = 0.0;
      (void)biquad_apply(c);
      a1 = c.a1; a2 = c.a2; a3 = c.a3; m0 = c.m0; m1 = c.m1; m2 = c.m2;
    };
    svf_scan_time_varying<true>(coef, f.st, buf, T, xchg);
    __syncthreads();
    if (tid == 0) (void)biquad_set(f.coef, btype, sr, cut[T - 1], qv[T - 1], 0.0f);  // the coefficient set the serial loop ends the block with
    for (int s = tid; s < 2 * T; s += nt) sig[2 * done + s] = (float)buf[REV_IDX(s >> 1, s & 1)];
  }
  __syncthreads();
}
// Eq5Effect while any of its fifteen smoothers ramps (eq5.rs:190-207,297-326): per band, three lanes lay out the band's bandwidth /
// frequency / gain sequences (the same sm_next calls as the serial loop, each smoother on its own: they do not interact), every frame's
// coefficients are recomputed from them as ramp_filter_coefficients does — q = bandwidth for the shelves, 1 / max(bandwidth, 0.001) for the
// bells — and the band runs as the time-varying blocked scan, f32 rounding between the cascaded bands as in the reference.
DEVO void eq5_ramp_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  PgEq5& e = fx.u.eq5;
  const int tid = pg_tid(), nt = blockDim.x;
  double* buf = (double*)fc.scratch;
  double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);
  PgState2* lst = (PgState2*)(xchg + 4);
  const int frames = n_samples / 2;
  const int piece = fc.tmp_floats / 3 < 1024 ? fc.tmp_floats / 3 : 1024;
  float* qv = fc.tmp;  // [piece] q, [piece] clamped cutoff, [piece] gain of the band being processed
  const float nyq = (float)fx.sample_rate / 2.0f;
  const uint32_t sr = fx.sample_rate;
  for (int done = 0; done < frames; done += piece) {
    const int T = frames - done < piece ? frames - done : piece;
    float* cut = qv + T;
    float* gn = cut + T;
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) buf[REV_IDX(s >> 1, s & 1)] = (double)sig[2 * done + s];
    for (int i = 0; i < 5; ++i) {
      __syncthreads();
      if (tid == 0) { sm_sequence(e.bws[i], qv, T); if (!(i == 0 || i == 4)) for (int k = 0; k < T; ++k) qv[k] = 1.0f / fmaxf(qv[k], 0.001f); }
      else if (tid == 64) { sm_sequence(e.freqs[i], cut, T); for (int k = 0; k < T; ++k) cut[k] = clampf(cut[k], 20.0f, nyq); }
      else if (tid == 128) sm_sequence(e.gains[i], gn, T);
      if (tid < 2) lst[tid] = e.st[tid][i];
      __syncthreads();
      const int btype = eq5_band_type(i);
      auto coef = [&](int n, double& a1, double& a2, double& a3, double& m0, double& m1, double& m2) {
        PgBiquadCoef c;
        c.type = btype; c.sample_rate = sr; c.cutoff = cut[n]; c.q = qv[n]; c.gain = gn[n];
        c.a1 = 0.0; c.a2 = 0.0; c.a3 = 0.0; c.m0 = 0.0; c.m1 = 0.0; c.m2 = 0.0;
        (void)biquad_apply(c);
        a1 = c.a1; a2 = c.a2; a3 = c.a3; m0 = c.m0; m1 = c.m1; m2 = c.m2;
      };
      svf_scan_time_varying<true>(coef, lst, buf, T, xchg);
      __syncthreads();
      if (tid < 2) e.st[tid][i] = lst[tid];
      if (tid == 0) (void)biquad_set(e.coef[i], btype, sr, cut[T - 1], qv[T - 1], gn[T - 1]);  // the set the serial loop ends the block with
    }
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) sig[2 * done + s] = (float)buf[REV_IDX(s >> 1, s & 1)];
  }
  __syncthreads();
}
DEVO bool eq5_steady(const PgEq5& e) {
  bool ramp = false;
  for (int i = 0; i < 5; ++i) ramp = ramp || sm_need_ramp(e.freqs[i]) || sm_need_ramp(e.bws[i]) || sm_need_ramp(e.gains[i]);
  return !ramp;
}

// Where a process call of the effect begins (the first piece of its chunk; every launch of a standalone effect): the per-call branch of the
// two effects whose other branch runs on state the ramp left behind (PgFx::call_ramp). Lane 0.
DEVO void fx_call_begin(PgFx& fx) {
  if (fx.kind == 2) fx.call_ramp = (sm_need_ramp(fx.u.filter.cutoff) || sm_need_ramp(fx.u.filter.q)) ? 1u : 0u;
  else if (fx.kind == 3) fx.call_ramp = eq5_steady(fx.u.eq5) ? 0u : 1u;
  else fx.call_ramp = 0u;
}

// Must mirror the acceptance conditions of fx_fast_process exactly: the fast kernel has no serial code to fall back to.
// staged_unit: the kernel that renders the unit in steady state carries no ramp paths (the staged kernels; the lean fast kernel of graphs that
// hold nothing but Gain / Panning / Reverb).
DEVO bool fx_fast_eligible(const PgFx& fx, bool staged_unit) {
  switch (fx.kind) {
    case 0: return !staged_unit || !sm_need_ramp(fx.u.gain.gain);   // ramping: the sequence paths of fx_fast_process (kernel variants with the ramp paths)
    case 1: return !staged_unit || (!sm_need_ramp(fx.u.pan.pan) && !sm_need_ramp(fx.u.pan.width));
    case 2: return !staged_unit || !(fx.call_ramp || sm_need_ramp(fx.u.filter.cutoff) || sm_need_ramp(fx.u.filter.q));  // ramping cutoff / Q (or a call that began so and still has pieces to go): time-varying scan (not in the staged kernels)
    case 3: return !staged_unit || (!fx.call_ramp && eq5_steady(fx.u.eq5));  // ramping: eq5_ramp_fast (like the Filter's ramps: not in the staged kernels)
    case 4: return delay_fast_eligible(fx) || (!staged_unit && delay_ramp_eligible(fx));  // ramping: delay_ramp_fast (not in the staged kernels)
    case 5: return reverb_fast_eligible(fx) || (!staged_unit && reverb_wet_ramp_eligible(fx));  // wet ramping: reverb_wet_ramp_fast (not in the staged kernels)
    case 6: return chorus_fast_eligible(fx) || (!staged_unit && chorus_ramp_eligible(fx));
    case 7: return comp_fast_eligible(fx);
    case 8: return true;
    case 9: return !staged_unit || (!sm_need_ramp(fx.u.dist.mix) && !sm_need_ramp(fx.u.dist.drive) && (fx.u.dist.mix.target == 0.0f || fx.u.dist.mix.target >= 1.0f));
    default: return false;
  }
}

// Memoryless / constant-gain cases: every sample is independent.
// KMASK: bit k set = the code of effect kind k is compiled into this kernel variant (register budget of the hot variants).
template <int KMASK>
DEVO bool fx_fast_process(PgFx& fx, float* sig, int n, FastCtx& fc) {
  const int tid = pg_tid(), nt = blockDim.x;
  if (!((KMASK >> fx.kind) & 1)) return false;
  switch (fx.kind) {
    case 0: {  // GainEffect without ramp: optional DC filter per channel (gain.rs:147-153), then scale_buffer (gain.rs:162-165)
      PgGain& g = fx.u.gain;
      // A moving gain (gain.rs:154-161: one smoother step per frame): one lane lays the value sequence of a piece out in the temporary row,
      // all lanes scale. Only in the kernel variants that carry the ramp paths (bit 10).
      const bool ramp = sm_need_ramp(g.gain);
      if (ramp && !((KMASK >> 10) & 1)) return false;
      float* gseq = fc.tmp;
      const int gcap = fc.tmp_floats < 1024 ? fc.tmp_floats : 1024;
      auto lay_out = [&](int T) {  // the gains of the next T frames -> gseq (caller syncs)
        if (tid == 0) sm_sequence(g.gain, gseq, T);
      };
      float v = g.gain.target;
      if (g.dc_mode != 0) {
        // DcFilter::process_sample per channel as a blocked scan over the block (dc_scan, shared with the DelayEffect's wet path),
        // rounded to f32 and scaled as the serial loop does. Only in the kernel variants that carry the Delay code (bit 4): the host
        // routes such chains to the wide kernels.
        if constexpr ((KMASK >> 4) & 1) {
          double* buf = (double*)fc.scratch;
          double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);
          const int frames = n / 2;
          const int piece = ramp ? gcap : 1024;
          for (int done = 0; done < frames; done += piece) {
            const int T = frames - done < piece ? frames - done : piece;
            __syncthreads();
            if (ramp) lay_out(T);
            for (int s = tid; s < 2 * T; s += nt) buf[REV_IDX(s >> 1, s & 1)] = (double)sig[2 * done + s];
            __syncthreads();
            dc_scan(g.dc, buf, T, xchg);
            __syncthreads();
            for (int s = tid; s < 2 * T; s += nt) sig[2 * done + s] = (float)buf[REV_IDX(s >> 1, s & 1)] * (ramp ? gseq[s >> 1] : v);
          }
          __syncthreads();
          return true;
        } else return false;
      }
      if (ramp) {
        const int frames = n / 2;
        for (int done = 0; done < frames; done += gcap) {
          const int T = frames - done < gcap ? frames - done : gcap;
          __syncthreads();
          lay_out(T);
          __syncthreads();
          for (int s = tid; s < 2 * T; s += nt) sig[2 * done + s] *= gseq[s >> 1];
        }
        __syncthreads();
        return true;
      }
      __syncthreads();
      for (int i = tid; i < n; i += nt) sig[i] = sig[i] * v;
      __syncthreads();
      return true;
    }
    case 1: {  // PanningEffect without ramps (pan.rs:105-158)
      PgPan& p = fx.u.pan;
      float inv_l = p.invert_l ? -1.0f : 1.0f, inv_r = p.invert_r ? -1.0f : 1.0f;
      if (sm_need_ramp(p.pan) || sm_need_ramp(p.width)) {
        // moving pan / width (pan.rs:122-156): the two smoothers' value sequences by two lanes (each only if it moves, as the serial loop decides
        // once per call), the per-frame mid / side and constant-power factors by all lanes
        if constexpr ((KMASK >> 10) & 1) {
          const bool pan_ramping = sm_need_ramp(p.pan), width_ramping = sm_need_ramp(p.width);
          const int cap = fc.tmp_floats / 2 < 1024 ? fc.tmp_floats / 2 : 1024;
          float* pseq = fc.tmp;
          float* wseq = fc.tmp + cap;
          const int frames = n / 2;
          for (int done = 0; done < frames; done += cap) {
            const int T = frames - done < cap ? frames - done : cap;
            __syncthreads();
            if (tid == 0) { if (pan_ramping) sm_sequence(p.pan, pseq, T); else { const float v = p.pan.target; for (int k = 0; k < T; ++k) pseq[k] = v; } }
            else if (tid == 64) { if (width_ramping) sm_sequence(p.width, wseq, T); else { const float v = p.width.target; for (int k = 0; k < T; ++k) wseq[k] = v; } }
            __syncthreads();
            for (int k = tid; k < T; k += nt) {
              const int f = 2 * (done + k);
              float l = sig[f] * inv_l, r = sig[f + 1] * inv_r;
              const float w = wseq[k];
              if (fabsf(w - 1.0f) > 1e-6f) { const float mid = (l + r) * 0.5f, side = (l - r) * 0.5f; l = mid + side * w; r = mid - side * w; }
              const float pv = pseq[k];
              if (fabsf(pv) > 1e-6f) { float pl, pr; panning_factors(pv, pl, pr); l *= pl; r *= pr; }
              sig[f] = l; sig[f + 1] = r;
            }
          }
          __syncthreads();
          return true;
        } else return false;
      }
      bool has_invert = inv_l < 0.0f || inv_r < 0.0f;
      float w = p.width.target, pv = p.pan.target;
      if (!has_invert && fabsf(pv) < 1e-6f && fabsf(w - 1.0f) < 1e-6f) return true;
      float pl = 1.0f, pr = 1.0f;
      bool do_pan = fabsf(pv) > 1e-6f, do_width = fabsf(w - 1.0f) > 1e-6f;
      if (do_pan) panning_factors(pv, pl, pr);
      __syncthreads();
      for (int f = tid * 2; f + 2 <= n; f += nt * 2) {
        float l = sig[f] * inv_l, r = sig[f + 1] * inv_r;
        if (do_width) { float mid = (l + r) * 0.5f, side = (l - r) * 0.5f; l = mid + side * w; r = mid - side * w; }
        if (do_pan) { l *= pl; r *= pr; }
        sig[f] = l; sig[f + 1] = r;
      }
      __syncthreads();
      return true;
    }
    case 2: if constexpr ((KMASK >> 2) & 1) {  // FilterEffect, no ramp (filter.rs:193-200)
      PgFilter& f = fx.u.filter;
      if (fx.call_ramp) {  // (the branch the call began in: fx_call_begin)
        // bit 10: kernel variants that carry the ramp paths (the fused wide kernel and the generic kernel; the staged kernels keep
        // their register allocation — measured +3.5 % on C5 with this code in them — and hand a ramping unit over as before)
        if constexpr ((KMASK >> 10) & 1) {
          if (fc.tmp_floats < 16) return false;
          filter_ramp_fast(fx, sig, n, fc);
          return true;
        } else return false;
      }
      biquad_chain_fast(sig, n, &f.coef, f.st, 1, 1, fc);
      return true;
    } else return false;
    case 3: if constexpr ((KMASK >> 3) & 1) {  // Eq5Effect, no ramp (eq5.rs:297-326)
      PgEq5& e = fx.u.eq5;
      if (fx.call_ramp) {
        if constexpr ((KMASK >> 10) & 1) {
          if (fc.tmp_floats < 24) return false;
          eq5_ramp_fast(fx, sig, n, fc);
          return true;
        } else return false;
      }
      biquad_chain_fast(sig, n, e.coef, &e.st[0][0], 5, 5, fc);
      return true;
    } else return false;
    case 4: if constexpr ((KMASK >> 4) & 1) {
      if (delay_fast(fx, sig, n, fc)) return true;
      if constexpr ((KMASK >> 10) & 1) return delay_ramp_fast(fx, sig, n, fc); else return false;
    } else return false;
    case 5: if constexpr ((KMASK >> 5) & 1) {
      if (reverb_fast<((KMASK >> 11) & 1) ? 4 : 1>(fx, sig, n, fc)) return true;  // (bit 11: the generic kernel — four sub-chunks per trip: measured 2 -> 4: C2 0.0355 -> 0.0341 ms per step, 8: 0.0350)
      if constexpr ((KMASK >> 10) & 1) return reverb_wet_ramp_fast(fx, sig, n, fc); else return false;
    } else return false;
    case 6: if constexpr ((KMASK >> 6) & 1) {
      if (chorus_fast(fx, sig, n, fc)) return true;
      if constexpr ((KMASK >> 10) & 1) return chorus_ramp_fast(fx, sig, n, fc); else return false;
    } else return false;
    case 7: if constexpr ((KMASK >> 7) & 1) return comp_fast(fx, sig, n, fc); else return false;
    case 8: if constexpr ((KMASK >> 8) & 1) return gate_fast(fx, sig, n, fc); else return false;
    case 9: if constexpr ((KMASK >> 9) & 1) {  // DistortionEffect, no ramps (distortion.rs:331-341)
      PgDist& d = fx.u.dist;
      if (!sm_need_ramp(d.mix) && d.mix.target == 0.0f) return true;   // (distortion.rs:331: nothing runs, no smoother moves)
      if (sm_need_ramp(d.mix) || sm_need_ramp(d.drive) || !(d.mix.target >= 1.0f)) {
        // a moving drive or mix, or a steady partial mix (distortion.rs:342-361: both smoothers step once per frame): two lanes lay the
        // sequences out, the waveshaper with its per-frame drive and compensation runs on all lanes
        if constexpr ((KMASK >> 10) & 1) {
          const bool full_wet = !sm_need_ramp(d.mix) && d.mix.target >= 1.0f;   // the branch that leaves the mix smoother alone
          const int cap = fc.tmp_floats / 2 < 1024 ? fc.tmp_floats / 2 : 1024;
          float* dseq = fc.tmp;
          float* mseq = fc.tmp + cap;
          const int ty = d.type;
          const int frames = n / 2;
          for (int done = 0; done < frames; done += cap) {
            const int T = frames - done < cap ? frames - done : cap;
            __syncthreads();
            if (tid == 0) sm_sequence(d.drive, dseq, T);
            else if (tid == 64 && !full_wet) sm_sequence(d.mix, mseq, T);
            __syncthreads();
            for (int s = tid; s < 2 * T; s += nt) {
              const float drive = dseq[s >> 1];
              const float comp = dist_compensation(d.luts, ty, drive);
              const float dry = sig[2 * done + s];
              const float wet = dist_shape(ty, dry, drive) * comp;
              if (full_wet) sig[2 * done + s] = wet;
              else { const float mix = mseq[s >> 1]; sig[2 * done + s] = (1.0f - mix) * dry + mix * wet; }
            }
          }
          __syncthreads();
          return true;
        } else return false;
      }
      float drive = d.drive.target;
      float comp = dist_compensation(d.luts, d.type, drive);
      int ty = d.type;
      __syncthreads();
      for (int i = tid; i < n; i += nt) sig[i] = dist_shape(ty, sig[i], drive) * comp;
      __syncthreads();
      return true;
    } else return false;
    default: return false;
  }
}

}  // namespace pgd
