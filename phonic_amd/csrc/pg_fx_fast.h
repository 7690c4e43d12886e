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
};

constexpr size_t FAST_SCRATCH_BYTES = (2 * 1024 + 128 + 8) * 8 + 16 * 40 + 16 * 8 + 13 * 16 + 4 * 8 + 9 * 16 * 3 * 8 + 8 * 129 * 2 * 8 + 64;  // reverb: f64 chunk buffer + phase records + epilogue gets

#include "pg_reverb_fast.inl"

// Must mirror the acceptance conditions of fx_fast_process exactly: the fast kernel has no serial code to fall back to.
DEVO bool fx_fast_eligible(const PgFx& fx) {
  switch (fx.kind) {
    case 0: return fx.u.gain.dc_mode == 0 && !sm_need_ramp(fx.u.gain.gain);
    case 1: return !sm_need_ramp(fx.u.pan.pan) && !sm_need_ramp(fx.u.pan.width);
    case 5: return reverb_fast_eligible(fx);
    case 9: return !sm_need_ramp(fx.u.dist.mix) && !sm_need_ramp(fx.u.dist.drive) && (fx.u.dist.mix.target == 0.0f || fx.u.dist.mix.target >= 1.0f);
    default: return false;
  }
}

// Memoryless / constant-gain cases: every sample is independent.
DEVO bool fx_fast_process(PgFx& fx, float* sig, int n, FastCtx& fc) {
  const int tid = threadIdx.x, nt = blockDim.x;
  switch (fx.kind) {
    case 0: {  // GainEffect without DC filter and without ramp: scale_buffer (gain.rs:162-165)
      const PgGain& g = fx.u.gain;
      if (g.dc_mode != 0 || sm_need_ramp(g.gain)) return false;
      float v = g.gain.target;
      __syncthreads();
      for (int i = tid; i < n; i += nt) sig[i] = sig[i] * v;
      __syncthreads();
      return true;
    }
    case 1: {  // PanningEffect without ramps (pan.rs:105-158)
      const PgPan& p = fx.u.pan;
      if (sm_need_ramp(p.pan) || sm_need_ramp(p.width)) return false;
      float inv_l = p.invert_l ? -1.0f : 1.0f, inv_r = p.invert_r ? -1.0f : 1.0f;
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
    case 5: return reverb_fast(fx, sig, n, fc);
    case 9: {  // DistortionEffect, no ramps (distortion.rs:331-341)
      const PgDist& d = fx.u.dist;
      if (sm_need_ramp(d.mix) || sm_need_ramp(d.drive)) return false;
      if (d.mix.target == 0.0f) return true;
      if (!(d.mix.target >= 1.0f)) return false;
      float drive = d.drive.target;
      float comp = dist_compensation(d.luts, d.type, drive);
      int ty = d.type;
      __syncthreads();
      for (int i = tid; i < n; i += nt) sig[i] = dist_shape(ty, sig[i], drive) * comp;
      __syncthreads();
      return true;
    }
    default: return false;
  }
}

}  // namespace pgd
