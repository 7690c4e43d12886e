// Voice rendering on gfx950: PreloadedFileSource (embedded cubic resampler, loop/repeat, VolumeFader) ->
// ChannelMappedSource -> AmplifiedSource -> PannedSource, i.e. the source chain `Player` builds for a
// file (reference src/player.rs:540-558), evaluated by one workgroup with the block in LDS.
//
// The resampler's f32 `sub_pos` schedule (which input frame feeds which output frame) is replayed
// sequentially by lane 0 exactly as CubicInterpolator::process does (src/utils/resampler/cubic.rs:72-111)
// — it is rounding dependent and must be bit exact — and only the 4-tap Hermite evaluation, the gathers
// from the PCM buffer in HBM and the gain/pan maths run on all lanes.
#pragma once
#include "pg_dsp_dev.h"

namespace pgd {

constexpr int SRC_OUT_CAP = 1024;  // output frames per resampling piece
#ifndef PG_SCHED_PAR_MIN
#define PG_SCHED_PAR_MIN 192       // pieces shorter than this take the one-lane walk of the resampler schedule instead of the time-parallel scan (ratio in [0.5, 1))
#endif
constexpr int SRC_WIN_CAP = 1040;  // consumed input frames per piece (+4 history)

struct SrcScratch {            // carved from the unit's LDS scratch arena
  uint16_t* sched_c;           // [SRC_OUT_CAP]   consumed-count at each output frame
  float* sched_f;              // [SRC_OUT_CAP]   interpolation fraction
  uint32_t* posmap;            // [SRC_WIN_CAP]   file frame index of each consumed frame
  float* win;                  // [2][SRC_WIN_CAP + 4] input window per channel (4 history + consumed)
  int32_t* ctl;                // [16] uniform control words written by lane 0
  unsigned long long* diag;
  const PgSchedEntry* sched_rd;  // this voice's class entry for the current launch (or nullptr)
};
constexpr size_t SRC_SCRATCH_BYTES = SRC_OUT_CAP * 2 + SRC_OUT_CAP * 4 + SRC_WIN_CAP * 4 + 2 * (SRC_WIN_CAP + 4) * 4 + 16 * 4;

DEVO void src_carve(char* base, SrcScratch& s) {
  s.sched_f = (float*)base; base += SRC_OUT_CAP * 4;
  s.posmap = (uint32_t*)base; base += SRC_WIN_CAP * 4;
  s.win = (float*)base; base += 2 * (SRC_WIN_CAP + 4) * 4;
  s.ctl = (int32_t*)base; base += 16 * 4;
  s.sched_c = (uint16_t*)base;
}

// Steady-state per-sample scalings of the source chain folded into the resampler's output loop: VolumeFader target
// (fader.rs:105-107), AmplifiedSource gain (smoothing.rs:67-70), PannedSource factors (smoothing.rs:100-108). Same
// multiplications in the same order as the separate passes; only used when nothing ramps and the file is stereo.
struct SrcPost { int on; int use_f, use_g, use_p; float fs, gain, pl, pr; };
DEVO float src_post(const SrcPost& P, float v, int ch) {
  if (P.use_f) v = v * P.fs;
  if (P.use_g) v = v * P.gain;
  if (P.use_p) v = v * (ch ? P.pr : P.pl);
  return v;
}

// Hermite x-form  src/utils/resampler/cubic.rs:125-142
DEVO float cubic_interp(float ym1, float y0, float y1, float y2, float fraction) {
  float c0 = y0;
  float c1 = (y1 - ym1) * 0.5f;
  float c2 = ym1 - y0 * 2.5f + y1 * 2.0f - y2 * 0.5f;
  float c3 = (y2 - ym1) * 0.5f + (y0 - y1) * 1.5f;
  return ((c3 * fraction + c2) * fraction + c1) * fraction + c0;
}

// ---- exact time-parallel evaluation of the resampler schedule (cubic.rs:72-90, ratio in [0.5, 1)) ----------------------------
// The reference advances an f32 accumulator once per output frame:  ge = sp >= 1; cc += ge; sp = ge ? sp - 1 : sp; emit;
// sp += ratio.  In units of u = 2^-24 every quantity is an integer (ratio in [0.5, 1) has ulp u; sp - 1 is exact; a sum below 1
// is exact; a sum A in [1, 2) is representable iff A is even, an odd A is a tie and rounds to the multiple of 4). So
//     U_k = U0_k + d_k,   U0_k = ((S + (k-1) R) mod 2^24) + R   (the recurrence without rounding, closed form),
//     d_k = d_{k-1} + rho(U0_k + d_{k-1}),   rho(A) = A >= 2^24 and A odd ? (A mod 4 == 3 ? +1 : -1) : 0,
// as long as the wrap decisions of the rounded and the unrounded sequence agree. d enters rho only through d mod 4, so the
// step k is a 4-entry increment table and the tables compose: one Kogge-Stone scan over the piece gives every d_k. Where a
// decision differs (a value within |d| of the wrap threshold, ~0.6 % of 1024-frame pieces) the closed form restarts from the
// exact state in front of it. Bit-identical to the serial recurrence (checked against it on the host model and by the parity tests).
// A rounding table: the increment of d for each value of d mod 4, four signed 16-bit fields in one 64-bit word (|d| <= 1024).
typedef unsigned long long SchedTab;
DEVO int schedtab_at(SchedTab a, int m) { return (int)(short)(a >> (16 * (m & 3))); }
DEVO SchedTab schedtab_pack(int t0, int t1, int t2, int t3) {
  return (unsigned long long)(unsigned short)t0 | ((unsigned long long)(unsigned short)t1 << 16) | ((unsigned long long)(unsigned short)t2 << 32) |
         ((unsigned long long)(unsigned short)t3 << 48);
}
DEVO SchedTab schedtab_compose(SchedTab a, SchedTab b) {  // a first, then b
  const int a0 = schedtab_at(a, 0), a1 = schedtab_at(a, 1), a2 = schedtab_at(a, 2), a3 = schedtab_at(a, 3);
  return schedtab_pack(a0 + schedtab_at(b, 0 + a0), a1 + schedtab_at(b, 1 + a1), a2 + schedtab_at(b, 2 + a2), a3 + schedtab_at(b, 3 + a3));
}
DEVO int sched_rho(int A) { return (A >= (1 << 24) && (A & 1)) ? ((A & 3) == 3 ? 1 : -1) : 0; }
// table of one step whose unrounded pre-wrap value is u0 (>= 2^24): rho(u0 + m) for m = 0..3 = the pattern {0, -1, 0, +1} rotated by u0 mod 4
DEVO SchedTab schedtab_step(int u0) {
  const int r = u0 & 3;
  int t[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { const int x = (r + m) & 3; t[m] = (x & 1) ? x - 2 : 0; }
  return schedtab_pack(t[0], t[1], t[2], t[3]);
}

// All lanes of the workgroup (256). `scr`: >= 32 ints of LDS. Returns false when it gave up (caller replays serially).
DEVO bool sched_parallel(float ratio, float sp0, int piece, uint16_t* oc, float* of, int* scr, int& c_total, float& sp_out, unsigned long long* diag = nullptr) {
  const int tid = pg_tid(), lane = tid & 63, wave = tid >> 6;
  const int TWO24 = 1 << 24;
  const int R = (int)(ratio * 16777216.0f);  // exact: ratio in [0.5, 1)
  int* s_viol = scr;          // first element whose wrap decision differs from the closed form
  int* s_start = scr + 1;     // restart: element, exact U of that element, wraps counted in front of it
  int* s_res = scr + 4;       // c_total, sp_out bits
  SchedTab* s_wave = (SchedTab*)(scr + 8);  // [4 waves] wave totals
  if (tid == 0) { s_start[0] = 0; s_start[1] = (int)(sp0 * 16777216.0f); s_start[2] = 0; }
  PG_STAMP(diag, 30);
  for (int iter = 0; iter < 16; ++iter) {
    __syncthreads();
    const int k0 = s_start[0], S = s_start[1], cc0 = s_start[2];
    if (tid == 0) *s_viol = 0x7fffffff;
    if (iter == 0) PG_STAMP(diag, 31);
    // closed form of this lane's four elements: U0 = ((S + (j-1) R) mod 2^24) + R — the low 24 bits of a 32-bit product suffice
    int U0[4]; int side[4];
    SchedTab T[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * tid + e, j = k - k0;
      int u0 = S;
      if (j >= 1) u0 = (int)(((unsigned)S + (unsigned)(j - 1) * (unsigned)R) & (unsigned)(TWO24 - 1)) + R;
      U0[e] = u0;
      side[e] = u0 >= TWO24;
      T[e] = (j >= 1 && k < piece && side[e]) ? schedtab_step(u0) : 0ull;
    }
    SchedTab incl = schedtab_compose(schedtab_compose(T[0], T[1]), schedtab_compose(T[2], T[3]));
    if (iter == 0) PG_STAMP(diag, 32);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const SchedTab b = __shfl_up(incl, off, 64);
      if (lane >= off) incl = schedtab_compose(b, incl);
    }
    if (lane == 63) s_wave[wave] = incl;
    SchedTab excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0ull;
    if (iter == 0) PG_STAMP(diag, 33);
    __syncthreads();
    int d = 0;  // d at the start element is 0; tables in front of it are identities
    for (int w = 0; w < wave; ++w) d += schedtab_at(s_wave[w], d);
    d += schedtab_at(excl, d);
    // walk the four elements with the true d
    int viol = 0x7fffffff;
    int Ue[4], cce[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * tid + e, j = k - k0;
      const int A = U0[e] + d;
      d += schedtab_at(T[e], d);
      const int U = U0[e] + d;
      Ue[e] = U;
      const int cc = cc0 + (int)(((unsigned long long)(unsigned)S + (unsigned long long)(unsigned)(j > 0 ? j : 0) * (unsigned)R) >> 24);
      cce[e] = cc;
      if (j >= 0 && k < piece) {
        if (j >= 1 && (((A >= TWO24) != (side[e] != 0)) || ((U >= TWO24) != (side[e] != 0))) && viol == 0x7fffffff) viol = k;
        const int P = U - (side[e] ? TWO24 : 0);
        oc[k] = (uint16_t)cc;
        of[k] = (float)P * 5.9604644775390625e-08f;
        if (k == piece - 1) { const int A1 = P + R; s_res[0] = cc; s_res[1] = (int)__float_as_uint((float)(A1 + sched_rho(A1)) * 5.9604644775390625e-08f); }
      }
    }
    if (iter == 0) PG_STAMP(diag, 34);
    if (viol != 0x7fffffff) atomicMin(s_viol, viol);
    __syncthreads();
    const int kv = *s_viol;
    if (kv == 0x7fffffff) { PG_STAMP(diag, 35); c_total = s_res[0]; sp_out = __uint_as_float((uint32_t)s_res[1]); return true; }
    // restart from the exact state of the element in front of the first differing decision
    const int kr = kv - 1;
    __syncthreads();
    if (kr >= 4 * tid && kr < 4 * tid + 4) {
      const int e = kr - 4 * tid;
      const int U = e == 0 ? Ue[0] : (e == 1 ? Ue[1] : (e == 2 ? Ue[2] : Ue[3]));
      const int cc = e == 0 ? cce[0] : (e == 1 ? cce[1] : (e == 2 ? cce[2] : cce[3]));
      s_start[0] = kr; s_start[1] = U; s_start[2] = cc - (U >= TWO24 ? 1 : 0);
    }
  }
  __syncthreads();
  return false;
}

// The same idea for ratio in [1, 4) (cubic.rs:92-111: voices played up to two octaves above the file's pitch); written out for [1, 2): Per output frame the reference
// runs  while sub_pos < ratio { push an input frame; sub_pos += 1 };  sub_pos -= ratio;  and interpolates at 1 - sub_pos. In units of
// u = 2^-23 (the ulp of the ratio and of every value in [1, 2)) the loop-top state S is an integer in [0, ONE], ONE = 2^23: the first push is
// exact; a second one (taken iff S + ONE < R, i.e. S + D < ONE with D = 2 ONE - R) lands in [2, 3) where the ulp is 2u — an odd value is a tie
// and goes to the multiple of four: rho of the schedule above, on S — and the subtraction is exact. So
//     S' = S + D + rho(S)   (two pushes)      or      S' = S + D - ONE   (one push),
// the unrounded sequence is X_j = (S_0 + j D) mod ONE in closed form, the true one X_j + d_j with d_{j+1} = d_j + rho(X_j + d_j) on the steps
// that push twice: the same 4-entry increment tables, the same scan, the same restart where a decision of the rounded sequence differs from
// the closed form's. Consumed frames after output j: 2 (j + 1) - floor((S_0 + (j + 1) D) / ONE). Model and check against the serial
// recurrence: tests/host/resampler_schedule_model.py (parallel_up), tests/test_host_models.py.
DEVO bool sched_parallel_up(float ratio, float sp0, int piece, uint16_t* oc, float* of, int* scr, int& c_total, float& sp_out) {
  const int tid = pg_tid(), lane = tid & 63, wave = tid >> 6;
  // ratio in [1, 2): unit 2^-23, at most K = 2 pushes, the second one rounds (it crosses 2.0). ratio in [2, 4): unit 2^-22 (the ulp of the ratio
  // and of [2, 4)), K = 3 or 4 pushes; pushes two and three are exact on that grid, a fourth crosses 4.0 and rounds the same way.
  const int SH = ratio < 2.0f ? 23 : 22;
  const int ONE = 1 << SH;
  const float unit = ratio < 2.0f ? 1.1920928955078125e-07f : 2.384185791015625e-07f;
  const int R = (int)(ratio * (float)ONE);  // exact
  const int K = R / ONE + 1;                // most pushes per output: 2, 3 or 4
  const bool rounds = K != 3;
  const int D = K * ONE - R;                // in (0, ONE]
  int* s_viol = scr;          // first output whose push count differs from the closed form's
  int* s_start = scr + 1;     // restart: output, its exact loop-top state, frames consumed in front of it (output < 0: give up)
  int* s_res = scr + 4;       // c_total, sp_out bits
  SchedTab* s_wave = (SchedTab*)(scr + 8);  // [4 waves] wave totals
  if (tid == 0) { s_start[0] = 0; s_start[1] = (int)(sp0 * (float)ONE); s_start[2] = 0; }
  for (int iter = 0; iter < 16; ++iter) {
    __syncthreads();
    const int k0 = s_start[0], S = s_start[1], cc0 = s_start[2];
    if (k0 < 0) break;
    if (tid == 0) *s_viol = 0x7fffffff;
    int X0[4]; int nowrap[4];
    SchedTab T[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * tid + e, j = k - k0;
      const int x0 = (int)(((unsigned)S + (unsigned)(j > 0 ? j : 0) * (unsigned)D) & (unsigned)(ONE - 1));  // the low 23 bits of a 32-bit product suffice
      X0[e] = x0;
      nowrap[e] = x0 + D < ONE;
      T[e] = (rounds && j >= 0 && k < piece && nowrap[e]) ? schedtab_step(x0) : 0ull;
    }
    SchedTab incl = schedtab_compose(schedtab_compose(T[0], T[1]), schedtab_compose(T[2], T[3]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const SchedTab b = __shfl_up(incl, off, 64);
      if (lane >= off) incl = schedtab_compose(b, incl);
    }
    if (lane == 63) s_wave[wave] = incl;
    SchedTab excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0ull;
    __syncthreads();
    int d = 0;  // d at the start output is 0; tables in front of it are identities
    for (int w = 0; w < wave; ++w) d += schedtab_at(s_wave[w], d);
    d += schedtab_at(excl, d);
    int viol = 0x7fffffff;
    int Se[4], ccb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * tid + e, j = k - k0;
      const int Sj = X0[e] + d;                       // exact loop-top state while every earlier decision agreed
      const int dn = d + schedtab_at(T[e], d);
      const int Sn = Sj + D - (nowrap[e] ? 0 : ONE) + (dn - d);
      const int jj = j > 0 ? j : 0;
      const int before = cc0 + K * jj - (int)(((unsigned long long)(unsigned)S + (unsigned long long)(unsigned)jj * (unsigned)D) >> SH);
      Se[e] = Sj; ccb[e] = before;
      d = dn;
      if (j >= 0 && k < piece) {
        if (((Sj + D < ONE) != (nowrap[e] != 0) || Sj < 0 || Sj > ONE) && viol == 0x7fffffff) viol = k;
        const int cc = before + (nowrap[e] ? K : K - 1);
        oc[k] = (uint16_t)cc;
        of[k] = (float)(ONE - Sn) * unit;
        if (k == piece - 1) { s_res[0] = cc; s_res[1] = (int)__float_as_uint((float)Sn * unit); }
      }
    }
    if (viol != 0x7fffffff) atomicMin(s_viol, viol);
    __syncthreads();
    const int kv = *s_viol;
    if (kv == 0x7fffffff) { c_total = s_res[0]; sp_out = __uint_as_float((uint32_t)s_res[1]); return true; }
    // restart AT the output whose decision differs: its loop-top state is exact (every earlier decision agreed)
    __syncthreads();
    if (kv >= 4 * tid && kv < 4 * tid + 4) {
      const int e = kv - 4 * tid;
      const int Sv = e == 0 ? Se[0] : (e == 1 ? Se[1] : (e == 2 ? Se[2] : Se[3]));
      const int cb = e == 0 ? ccb[0] : (e == 1 ? ccb[1] : (e == 2 ? ccb[2] : ccb[3]));
      if (Sv < 0 || Sv >= ONE || kv == k0) s_start[0] = -1;   // outside the closed form's range, or no progress: the serial walk takes the piece
      else { s_start[0] = kv; s_start[1] = Sv; s_start[2] = cb; }
    }
  }
  __syncthreads();
  return false;
}

// sample count -> frame count. A 64-bit division is a long software routine on this target; files are mono or stereo.
DEVO uint64_t div_channels(uint64_t x, int C) { return C == 2 ? (x >> 1) : (C == 1 ? x : x / (uint64_t)C); }

// PreloadedFileSource::write_buffer (src/source/file/preloaded.rs:270-332) for `out_frames` frames of the file's
// channel layout into `out` (LDS). `v` is the unit's LDS copy of the voice. Returns frames written (uniform).
// When `acc` is given (steady state), the finished samples are added straight into the mixer's block (add_buffers,
// src/source/mixed.rs:606-608) instead of going through the temporary mix buffer.
DEVO int src_write_buffer(PgVoice* v, float* out, int out_frames, const SrcScratch& S, const SrcPost& P, float* acc) {
  const int C = (int)v->channels;
  const int tid = pg_tid(), nt = blockDim.x;
  // loop range in samples (:273-280)
  uint64_t lr_start = 0, lr_end = v->n_samples;
  if (v->repeat > 0 && v->has_loop) { lr_start = v->loop_start * C; lr_end = v->loop_end * C; }
  int written = 0;  // frames
  PG_STAMP(S.diag, 37);
  const bool bypass = fabsf(v->ratio - 1.0f) < 0.000001f;  // cubic.rs:53-58
  while (written < out_frames) {
    __syncthreads();
    if (bypass) {
      // whole-buffer copy branch: (min, min) consumed/produced
      uint64_t pp = v->playback_pos;
      uint64_t remaining_in = lr_end > pp ? lr_end - pp : 0;
      uint64_t want = (uint64_t)(out_frames - written) * C;
      int nsm = (int)(remaining_in < want ? remaining_in : want);
      if (P.on && acc && C == 1) {  // a steady mono file at the mixer's rate: ChannelMappedSource's duplication (buffer.rs:209-217), gain and panning
        for (int i = tid; i < nsm; i += nt) {   // per output sample, added straight into the mixer's block — the same multiplications in the same order
          const float x = v->pcm[pp + i];      // as the separate passes (five LDS passes less per block)
          float* a2 = acc + 2 * (written + i);
          a2[0] = a2[0] + src_post(P, x, 0);
          a2[1] = a2[1] + src_post(P, x, 1);
        }
      } else if (P.on && acc) { for (int i = tid; i < nsm; i += nt) acc[written * C + i] = acc[written * C + i] + src_post(P, v->pcm[pp + i], i & 1); }
      else if (P.on) { for (int i = tid; i < nsm; i += nt) out[written * C + i] = src_post(P, v->pcm[pp + i], i & 1); }
      else for (int i = tid; i < nsm; i += nt) out[written * C + i] = v->pcm[pp + i];
      __syncthreads();
      if (tid == 0) {
        v->playback_pos = pp + nsm;
        if (v->playback_pos >= lr_end) {  // :317-326
          if (v->repeat_count > 0) { if (v->repeat_count != PG_USIZE_MAX) v->repeat_count -= 1; v->playback_pos = lr_start; }
          else v->pos_eof = 1;
        }
        S.ctl[0] = nsm / C;
      }
      __syncthreads();
      int produced = S.ctl[0];
      written += produced;
      if (v->pos_eof && produced == 0) break;
      continue;
    }
    // ---- one resampling piece: lane 0 replays the schedule ----
    int piece = out_frames - written;
    if (piece > SRC_OUT_CAP) piece = SRC_OUT_CAP;
    {
      int per_out = v->ratio < 1.0f ? 1 : (int)ceilf(v->ratio) + 1;  // most input frames one output frame can consume (+3 preload: SRC_WIN_CAP margin)
      int cap = (SRC_WIN_CAP - 8) / per_out;
      if (cap < 1) cap = 1;
      if (piece > cap) piece = cap;
    }
    // steady playback with ratio in [0.5, 1): every lane takes part in the exact time-parallel schedule
    bool par_ok = false;
    bool par_short = false;   // eligible for the time-parallel schedule, but the piece is short: the one-lane walk below is the cheaper way
    int par_c = 0;
    float par_sp = 0.0f;
    unsigned long long win_x[4] = {0ull, 0ull, 0ull, 0ull};  // prefetched stereo frames of the input window
    int win_pref = 0;
    {
      const float ratio = v->ratio, sp0 = v->sub_pos[0];
      const uint64_t pp0 = v->playback_pos;
      const uint64_t num_in0 = div_channels(lr_end > pp0 ? lr_end - pp0 : 0, C);
      const float t = sp0 * 16777216.0f;
      if (ratio >= 0.5f && ratio < 1.0f && v->initialized[0] && num_in0 > (uint64_t)piece && sp0 >= 0.0f && sp0 < 2.0f && t == floorf(t) && nt == 256) {
        if (C == 2) {  // the input window goes out to HBM before the schedule is known: an upper bound of the consumed frames is
          typedef __attribute__((address_space(1))) const unsigned long long gu64;
          gu64* src = (gu64*)(v->pcm) + (uint64_t)(pp0 / 2);
          int bound = (int)(sp0 + (float)piece * ratio) + 2;
          if ((uint64_t)bound > num_in0) bound = (int)num_in0;
          if (bound > 4 * nt) bound = 4 * nt;
          win_pref = bound;
#pragma unroll
          for (int k = 0; k < 4; ++k) { const int j = k * nt + tid; win_x[k] = j < bound ? src[j] : 0ull; }
        }
        // Short pieces (small real-time callbacks): the time-parallel schedule is a fixed ~12 K cycles of scan levels and barriers on a workgroup
        // whose block is a latency chain, the branch-free walk of one lane below ~15 cycles per output frame — the walk wins below a few hundred
        // frames (stamps per callback size: profiles/r05_headline_stamps_by_callback_size.txt). The window stays requested either way.
        if (piece >= PG_SCHED_PAR_MIN) par_ok = sched_parallel(ratio, sp0, piece, S.sched_c, S.sched_f, (int*)S.posmap, par_c, par_sp, S.diag);
        else par_short = true;
      } else {
        const float t23 = sp0 * (ratio < 2.0f ? 8388608.0f : 4194304.0f);
        if (ratio >= 1.0f && ratio < 4.0f && v->initialized[0] && num_in0 > 4ull * (uint64_t)piece + 4ull && sp0 >= 0.0f && sp0 < 1.0f && t23 == floorf(t23) && nt == 256) {
          if (C == 2) {  // (as above: at most ceil(ratio) input frames per output — the piece is capped for that)
            typedef __attribute__((address_space(1))) const unsigned long long gu64;
            gu64* src = (gu64*)(v->pcm) + (uint64_t)(pp0 / 2);
            int bound = (int)((float)piece * ratio) + 3;
            if ((uint64_t)bound > num_in0) bound = (int)num_in0;
            if (bound > 4 * nt) bound = 4 * nt;
            win_pref = bound;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int j = k * nt + tid; win_x[k] = j < bound ? src[j] : 0ull; }
          }
          par_ok = sched_parallel_up(ratio, sp0, piece, S.sched_c, S.sched_f, (int*)S.posmap, par_c, par_sp);
        }
      }
    }
    if (tid == 0) {
      float sub_pos = v->sub_pos[0];
      const float ratio = v->ratio;
      int initialized = v->initialized[0];
      uint64_t pp = v->playback_pos;
      uint64_t repeat_count = v->repeat_count;
      int eof = v->pos_eof;
      int c = 0;         // consumed frames in this piece
      int produced = 0;  // produced frames in this piece
      int linear = 0;
      {
        // Common case (steady playback, no loop wrap inside the piece): ONE resampler.process call covers the piece and
        // the input cannot run out (at most one push per output when ratio < 1), so the `consumed >= num_in` exits of
        // cubic.rs:75-77 are dead and the schedule is a branch-free f32 recurrence — the same operations in the same order.
        uint64_t remaining_in = lr_end > pp ? lr_end - pp : 0;
        uint64_t num_in = div_channels(remaining_in, C);
        const PgSchedEntry* se = S.sched_rd;
        if (se && !(se->valid && se->piece == piece && se->ratio_bits == __float_as_uint(ratio) && se->subpos_in_bits == __float_as_uint(sub_pos))) se = nullptr;
        // A voice's first process call pushes three input frames before anything else (cubic.rs:61-69): they are the first three of the run the
        // walks below consume, so the walks start their count at three. (Such a piece used to fall through to the general loop at the end — two
        // branches and two LDS stores per frame on this lane, ~100 K cycles for a block, and every note that starts makes its launch wait for it:
        // tools/exp_extra_voices.py.)
        const int c_pre = (!initialized && num_in >= 3) ? 3 : 0;
        if (par_ok) {
          S.ctl[3] = (int)(uint32_t)div_channels(pp, C);
          c = par_c;
          sub_pos = par_sp;
          pp += (uint64_t)c * C;
          produced = piece;
          linear = 1;
        } else if (ratio < 1.0f && initialized && num_in > (uint64_t)piece && se) {
          // schedule cache hit: the class representative replayed exactly this recurrence; lanes copy it below
          S.ctl[3] = (int)(uint32_t)div_channels(pp, C);
          c = se->c_total;
          sub_pos = __uint_as_float(se->subpos_out_bits);
          pp += (uint64_t)c * C;
          produced = piece;
          linear = 2;
        } else if (ratio < 1.0f && (initialized || c_pre) && num_in > (uint64_t)piece + (uint64_t)c_pre) {
          // Keep the recurrence entirely in the vector ALU (the values are wave-uniform, and the compiler would otherwise
          // bounce the compare result through the scalar unit every step): 3 dependent VALU ops per output frame.
          initialized = 1;
          float sp = sub_pos;
          int cc = c + c_pre;
          asm volatile("" : "+v"(sp), "+v"(cc));
          int k = 0;
          for (; k + 4 <= piece; k += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool ge = sp >= 1.0f;
              cc += ge ? 1 : 0;
              sp = ge ? sp - 1.0f : sp;
              S.sched_c[k + j] = (uint16_t)cc;
              S.sched_f[k + j] = sp;
              sp += ratio;
            }
          }
          for (; k < piece; ++k) {
            const bool ge = sp >= 1.0f;
            cc += ge ? 1 : 0;
            sp = ge ? sp - 1.0f : sp;
            S.sched_c[k] = (uint16_t)cc;
            S.sched_f[k] = sp;
            sp += ratio;
          }
          sub_pos = sp;
          c = cc;
          S.ctl[3] = (int)(uint32_t)div_channels(pp, C);
          pp += (uint64_t)c * C;  // num_in > piece >= c: the loop end cannot be reached
          produced = piece;
          linear = 1;
        } else if (ratio >= 1.0f && ratio <= 4.0f && (initialized || c_pre) && sub_pos >= 0.0f) {
          // The same for ratio >= 1 (cubic.rs:92-111; a sampler's notes above the file's pitch): an output frame pushes input frames while
          // sub_pos < ratio — at most K = ceil(ratio) of them, since sub_pos >= 0 — then steps back by ratio. With more than piece * K input
          // frames in front of the loop end the `consumed >= num_in` exit is dead and the K conditional pushes are straight-line code: the
          // same comparisons, additions of 1.0f and the subtraction in the same order (the general loop below pays two branches and an LDS
          // store per pushed frame).
          const int K = (int)ceilf(ratio);
          if (num_in > (uint64_t)piece * (uint64_t)K + (uint64_t)K + (uint64_t)c_pre) {
            float sp = sub_pos;
            int cc = c + c_pre;
            asm volatile("" : "+v"(sp), "+v"(cc));
            bool short_of = false;
            for (int k = 0; k < piece; ++k) {
              for (int j = 0; j < K; ++j) { const bool lt = sp < ratio; cc += lt ? 1 : 0; sp = lt ? sp + 1.0f : sp; }
              short_of |= sp < ratio;
              sp -= ratio;
              S.sched_c[k] = (uint16_t)cc;
              S.sched_f[k] = 1.0f - sp;
            }
            if (!short_of) {  // (always: K pushes reach ratio from any sub_pos >= 0; the general loop takes over otherwise)
              initialized = 1;
              sub_pos = sp;
              c = cc;
              S.ctl[3] = (int)(uint32_t)div_channels(pp, C);
              pp += (uint64_t)c * C;
              produced = piece;
              linear = 1;
            }
          }
        }
      }
      S.ctl[2] = linear;
      PG_STAMP_VAL(S.diag, 40, linear); PG_STAMP_VAL(S.diag, 41, S.sched_rd != nullptr);
      if (S.sched_rd) { PG_STAMP_VAL(S.diag, 42, S.sched_rd->valid); PG_STAMP_VAL(S.diag, 43, S.sched_rd->piece); PG_STAMP_VAL(S.diag, 44, piece); PG_STAMP_VAL(S.diag, 45, S.sched_rd->subpos_in_bits); PG_STAMP_VAL(S.diag, 46, __float_as_uint(v->sub_pos[0])); PG_STAMP_VAL(S.diag, 47, S.sched_rd->ratio_bits); PG_STAMP_VAL(S.diag, 48, __float_as_uint(ratio)); }
      while (produced < piece) {  // write_buffer loop :286-331; each iteration = one resampler.process call
        uint64_t remaining_in = lr_end > pp ? lr_end - pp : 0;
        const uint64_t num_in64 = div_channels(remaining_in, C);
        int num_in = (int)(num_in64 > 0x7fffffff ? 0x7fffffff : num_in64);
        int num_out = piece - produced;
        int consumed = 0, prod = 0;
        uint32_t base_frame = (uint32_t)div_channels(pp, C);
        if (!initialized && num_in >= 3) {  // cubic.rs:61-69
          initialized = 1;
          for (int f = 0; f < 3; ++f) { S.posmap[c] = base_frame + consumed; ++c; ++consumed; }
        }
        if (ratio < 1.0f) {  // cubic.rs:72-90
          while (prod < num_out) {
            if (sub_pos >= 1.0f) {
              if (consumed >= num_in) break;
              S.posmap[c] = base_frame + consumed; ++c; ++consumed;
              sub_pos -= 1.0f;
            }
            S.sched_c[produced + prod] = (uint16_t)c;
            S.sched_f[produced + prod] = sub_pos;
            ++prod;
            sub_pos += ratio;
          }
        } else {  // cubic.rs:92-111
          bool brk = false;
          while (prod < num_out) {
            while (sub_pos < ratio) {
              if (consumed >= num_in) { brk = true; break; }
              S.posmap[c] = base_frame + consumed; ++c; ++consumed;
              sub_pos += 1.0f;
            }
            if (brk) break;
            sub_pos -= ratio;
            S.sched_c[produced + prod] = (uint16_t)c;
            S.sched_f[produced + prod] = 1.0f - sub_pos;
            ++prod;
          }
        }
        pp += (uint64_t)consumed * C;
        produced += prod;
        if (pp >= lr_end) {  // :317-326
          if (repeat_count > 0) { if (repeat_count != PG_USIZE_MAX) repeat_count -= 1; pp = lr_start; }
          else eof = 1;
        }
        if (eof && prod == 0) break;  // :327-330
      }
      v->sub_pos[0] = sub_pos; v->sub_pos[1] = sub_pos;
      v->initialized[0] = initialized; v->initialized[1] = initialized;
      v->playback_pos = pp; v->repeat_count = repeat_count; v->pos_eof = eof;
      // 2 = the voice needs no published schedule: it took the time-parallel one, or walked a short piece by choice (a class representative
      // that published for such voices replayed the NEXT piece into global memory from one lane — 22 K cycles at the end of ITS workgroup's
      // 128-frame block, a quarter of the launch, for a schedule every voice of the class walks in 2 K: profiles/r05_headline_stamps_by_callback_size.txt)
      v->sched_hit = (par_ok || (par_short && linear == 1)) ? 2 : (linear == 2 ? 1 : 0);
      S.ctl[0] = produced; S.ctl[1] = c;
    }
    __syncthreads();
    PG_STAMP(S.diag, 17);
    const int produced = S.ctl[0], c_total = S.ctl[1];
    if (S.ctl[2] == 2) {  // cache hit: coalesced copy of the shared schedule (L2 resident) into LDS
      const PgSchedEntry* se = S.sched_rd;
      for (int k = tid; k < produced; k += nt) { S.sched_c[k] = se->sched_c[k]; S.sched_f[k] = se->sched_f[k]; }
      __syncthreads();
    }
    // ---- all lanes: gather the consumed frames (coalesced runs between loop wraps) and the history ----
    if (C == 2 && S.ctl[2]) {
      // stereo, one contiguous run of input frames: one 8-byte load per frame, four loads in flight per lane
      typedef __attribute__((address_space(1))) const unsigned long long gu64;  // one 8-byte global load = one stereo frame
      gu64* src = (gu64*)(v->pcm) + (uint64_t)(uint32_t)S.ctl[3];
      float* w0 = S.win;
      float* w1 = S.win + (SRC_WIN_CAP + 4);
      if (tid < 4) { w0[tid] = v->input[0][3 - tid]; w1[tid] = v->input[1][3 - tid]; }  // oldest first: input[3] .. input[0]
      if (win_pref > 0 && c_total <= win_pref) {   // (requested in front of the schedule, from the run's first frame: par_ok or the linear walk)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = k * nt + tid;
          if (j < c_total) { w0[4 + j] = __uint_as_float((uint32_t)win_x[k]); w1[4 + j] = __uint_as_float((uint32_t)(win_x[k] >> 32)); }
        }
      } else
      for (int jb = 0; jb < c_total; jb += 4 * nt) {
        unsigned long long x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = jb + k * nt + tid; x[k] = j < c_total ? src[j] : 0ull; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = jb + k * nt + tid;
          if (j < c_total) { w0[4 + j] = __uint_as_float((uint32_t)x[k]); w1[4 + j] = __uint_as_float((uint32_t)(x[k] >> 32)); }
        }
      }
    } else {
      for (int ch = 0; ch < C; ++ch) {
        float* w = S.win + ch * (SRC_WIN_CAP + 4);
        if (tid < 4) w[tid] = v->input[ch][3 - tid];  // oldest first: input[3], input[2], input[1], input[0]
        if (S.ctl[2]) { const uint64_t b = (uint64_t)(uint32_t)S.ctl[3]; for (int j = tid; j < c_total; j += nt) w[4 + j] = v->pcm[(b + j) * C + ch]; }
        else for (int j = tid; j < c_total; j += nt) w[4 + j] = v->pcm[(uint64_t)S.posmap[j] * C + ch];
      }
    }
    __syncthreads();
    PG_STAMP(S.diag, 18);
    // ---- all lanes: 4-tap Hermite per output frame and channel ----
    for (int i = tid; i < produced * C; i += nt) {
      int k = i / C, ch = i - k * C;
      const float* w = S.win + ch * (SRC_WIN_CAP + 4);
      int c = S.sched_c[k];
      const float y = cubic_interp(w[c], w[c + 1], w[c + 2], w[c + 3], S.sched_f[k]);
      if (P.on && acc) acc[(written + k) * C + ch] = acc[(written + k) * C + ch] + src_post(P, y, ch);
      else out[(written + k) * C + ch] = P.on ? src_post(P, y, ch) : y;
    }
    __syncthreads();
    PG_STAMP(S.diag, 19);
    if (tid == 0) {  // new history = the 4 newest window entries
      for (int ch = 0; ch < C; ++ch) {
        const float* w = S.win + ch * (SRC_WIN_CAP + 4);
        float h0 = w[c_total + 3], h1 = w[c_total + 2], h2 = w[c_total + 1], h3 = w[c_total];
        v->input[ch][0] = h0; v->input[ch][1] = h1; v->input[ch][2] = h2; v->input[ch][3] = h3;
      }
    }
    __syncthreads();
    written += produced;
    if (v->pos_eof && produced == 0) break;
    if (produced == 0 && c_total == 0) break;  // defensive: no progress possible
  }
  __syncthreads();
  return written;
}

// FileSourceImpl::update_speed  src/source/file/common.rs:141-169 (lane 0)
DEVO void voice_update_speed(PgVoice* v) {
  const double speed_diff = v->target_speed - v->current_speed;
  if (v->speed_glide_rate > 0.0f && fabs(speed_diff) > 0.0001) {
    const double semitone_diff = fabs(12.0 * log2(v->target_speed / v->current_speed));
    const float duration_secs = (float)semitone_diff / v->speed_glide_rate;
    if (duration_secs > 0.0f) {
      const float duration_frames = duration_secs * (float)v->out_rate;
      const double speed_step_per_frame = (v->target_speed - v->current_speed) / (double)duration_frames;
      const double speed_change_this_call = speed_step_per_frame * 64.0;  // SPEED_UPDATE_CHUNK_SIZE
      if (fabs(v->target_speed - v->current_speed) < fabs(speed_change_this_call)) v->current_speed = v->target_speed;
      else v->current_speed += speed_change_this_call;
    } else v->current_speed = v->target_speed;
  } else v->current_speed = v->target_speed;
  const uint32_t new_output_rate = (uint32_t)d2u64((double)v->out_rate / v->current_speed);
  v->ratio = (float)((double)v->src_rate / (double)new_output_rate);  // CubicResampler::update  cubic.rs:188-200
}
// PreloadedFileSource::set_speed  preloaded.rs:181-192 (lane 0)
DEVO void voice_set_speed(PgVoice* v, double speed, float glide) {
  if (v->finished || v->stream_on) return;  // (a host-fed source has no resampler of its own; the host refuses the call)
  v->samples_to_next_speed_update = 0;
  v->target_speed = speed;
  v->speed_glide_rate = glide > 0.0f ? glide : 0.0f;
  if (v->speed_glide_rate == 0.0f) { v->current_speed = speed; voice_update_speed(v); }
}
// PreloadedFileSource::seek  preloaded.rs:139-147 (lane 0)
DEVO void voice_seek(PgVoice* v, double seconds) {
  if (v->finished || v->stream_on) return;  // (playback_pos of a host-fed source counts the ring frames read)
  const double buffer_pos = seconds * (double)v->src_rate * (double)v->channels;
  uint64_t p = d2u64(buffer_pos);
  v->playback_pos = p < v->n_samples ? p : v->n_samples;
  for (int c = 0; c < 2; ++c) { for (int k = 0; k < 4; ++k) v->input[c][k] = 0.0f; v->sub_pos[c] = 0.0f; v->initialized[c] = 0; }  // resampler.reset()
}

// PreloadedFileSource::write (preloaded.rs:396-475): `frames` frames of the FILE's channel layout into `out` (LDS or global memory),
// VolumeFader applied, `finished` kept. Returns frames written. `post_on` (out): the fader / gain / pan of a steady stereo voice were
// fused into the resampler's output loop (and, with `acc`, added straight into the mixer's block): nothing is left to do for the caller.
// GLIDE = false: the kernel variant never sees a gliding voice (the fast kernel defers such units), so the loop is left out.
// `ask_ends`: the mixer's write call ends with these frames. A call that fills a piece of its chunk goes on in the chunk's next piece (the
// mixer asks a source once per chunk, mixed.rs:595-600): what the reference does when a call returns — the fader's "have we arrived" test
// (fader.rs:118-121) and with it `playback_finished` — waits for the piece the call ends in.
template <bool GLIDE>
DEVO int file_source_write(PgVoice* v, float* out, int frames, int pending_stop, const SrcScratch& S, float* acc, bool allow_post, int* post_on, bool ask_ends = true) {
  *post_on = 0;
  const int tid = pg_tid(), nt = blockDim.x;
  const int C = (int)v->channels;
  PG_STAMP(S.diag, 36);
  // process_messages: Stop (preloaded.rs:195-208)
  if (tid == 0 && pending_stop && !v->finished) {
    if (v->fade_out_seconds > 0.0f) {  // VolumeFader::start_fade_out  fader.rs:67-91
      float from = (v->fader_state == 1) ? v->fader_current : 1.0f;
      v->fader_state = 1; v->fader_current = from; v->fader_target = 0.0f;
      const float LN100 = 4.605f;
      float samples_duration = (float)v->out_rate * v->fade_out_seconds / LN100;
      v->fader_inertia = 1.0f - pg_expf_glibc(-1.0f / samples_duration);
    } else {
      v->finished = 1;
    }
  }
  __syncthreads();
  if (v->finished) return 0;
  SrcPost P;
  // (mono: only the whole-buffer copy branch of a file at the mixer's rate folds the mapping in, and only straight into the mixer's block)
  const bool mono_fused = C == 1 && acc != nullptr && fabsf(v->ratio - 1.0f) < 0.000001f && !(GLIDE && v->current_speed != v->target_speed);
  P.on = (allow_post && (C == 2 || mono_fused) && v->fader_state != 1 && !sm_need_ramp(v->volume) && !sm_need_ramp(v->panning)) ? 1 : 0;
  P.fs = v->fader_target; P.use_f = P.fs != 1.0f;
  P.gain = v->volume.target; P.use_g = fabsf(1.0f - P.gain) > 0.000001f;
  P.use_p = fabsf(v->panning.target) > 0.000001f; P.pl = 1.0f; P.pr = 1.0f;
  if (P.use_p) panning_factors(v->panning.target, P.pl, P.pr);
  int wf;
  if (GLIDE && v->current_speed != v->target_speed) {
    // pitch glide: the resampler is re-targeted every 64 frames (preloaded.rs:421-446, common.rs:56)
    wf = 0;
    while (wf < frames) {
      __syncthreads();
      if (tid == 0 && v->samples_to_next_speed_update == 0) {
        if (v->current_speed != v->target_speed) voice_update_speed(v);
        v->samples_to_next_speed_update = 64u * (uint32_t)C;
      }
      __syncthreads();
      int chunk = frames - wf;
      const int until = (int)(v->samples_to_next_speed_update / (uint32_t)C);
      if (chunk > until) chunk = until;
      const int w = src_write_buffer(v, out + wf * C, chunk, S, P, (P.on && acc) ? acc + wf * C : nullptr);
      __syncthreads();
      if (tid == 0) v->samples_to_next_speed_update -= (uint32_t)(w * C);
      wf += w;
      if (w < chunk) break;  // input exhausted
    }
    __syncthreads();
  } else {
    if (tid == 0) v->samples_to_next_speed_update = 0;
    wf = src_write_buffer(v, out, frames, S, P, P.on ? acc : nullptr);  // frames of the file layout
  }
  *post_on = P.on;
  const int total = wf * C;
  if (!P.on) {  // VolumeFader::process  fader.rs:103-122 (a steady voice had it applied in the resampler's output loop)
    if (v->fader_state != 1) {
      float tv = v->fader_target;
      if (tv != 1.0f) for (int i = tid; i < total; i += nt) out[i] = out[i] * tv;
    } else {
      // a fade in progress (fader.rs:109-117): one lane walks the f32 recurrence and lays the per-frame factors out in the source scratch (free
      // now: the resampler has written), all lanes multiply — the same products as the reference's loop
      float* const seq = S.sched_f;
      constexpr int SEQ_CAP = SRC_OUT_CAP + SRC_WIN_CAP + 2 * (SRC_WIN_CAP + 4);
      __syncthreads();
      for (int base = 0; base < wf; base += SEQ_CAP) {
        const int n = wf - base < SEQ_CAP ? wf - base : SEQ_CAP;
        if (tid == 0) {
          float cur = v->fader_current;
          const float tgt = v->fader_target, inertia = v->fader_inertia;
          // (eight values per trip, written out: inside the render kernels the compiler leaves such a walk rolled — see sm_sequence, pg_dsp_dev.h)
          int f = 0;
          for (; f + 8 <= n; f += 8) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { cur += (tgt - cur) * inertia; o[k] = cur; }
#pragma unroll
            for (int k = 0; k < 8; ++k) seq[f + k] = o[k];
          }
          for (; f < n; ++f) { cur += (tgt - cur) * inertia; seq[f] = cur; }
          v->fader_current = cur;
        }
        __syncthreads();
        for (int i = tid; i < n * C; i += nt) out[base * C + i] *= seq[C == 2 ? (i >> 1) : (C == 1 ? i : i / C)];
        __syncthreads();
      }
      if (tid == 0 && (ask_ends || wf < frames) && fabsf(v->fader_current - v->fader_target) < 0.0001f) v->fader_state = 2;
    }
  }
  __syncthreads();
  if (tid == 0 && (ask_ends || wf < frames)) {  // preloaded.rs:465-472, when the write call returns (at the end of the file the resampler may still
    bool fade_out_completed = v->fader_state == 2 && v->fader_target == 0.0f;   // hold a frame or two of output for the call's next piece)
    if (v->pos_eof || fade_out_completed) v->finished = 1;
  }
  __syncthreads();
  return wf;
}

// Source::write of a host-fed source (any `dyn Source` the host pulls itself — a synth, a streamed file — src/source.rs:80-110,
// src/source/synth/common.rs:194-263): the next frames of the device ring the host keeps filled, as many as are there (a short read is a
// source that delivered less); exhausted once the host ended the stream and everything fed has been read. A stop ends it at once (the host's
// own source fades before it ends its stream). `frames` frames of the source's channel layout into `out`; returns frames written.
DEVO int stream_source_write(PgVoice* v, float* out, int frames, int pending_stop) {
  const int tid = pg_tid(), nt = blockDim.x;
  const int C = (int)v->channels;
  __syncthreads();
  if (tid == 0 && pending_stop) v->finished = 1;
  __syncthreads();
  if (v->finished) return 0;
  const uint64_t fed = v->stream_fed & 0x7fffffffffffffffull, rd = v->playback_pos;
  const int ended = (int)(v->stream_fed >> 63);
  const uint64_t avail = fed > rd ? fed - rd : 0;
  const int n = (uint64_t)frames < avail ? frames : (int)avail;
  const uint32_t cap = v->stream_cap, r0 = (uint32_t)(rd % (uint64_t)cap);
  for (int i = tid; i < n * C; i += nt) {
    const int f = C == 2 ? (i >> 1) : i, c = C == 2 ? (i & 1) : 0;
    uint32_t p = r0 + (uint32_t)f;
    if (p >= cap) p -= cap;
    out[i] = v->pcm[(size_t)p * C + c];
  }
  __syncthreads();
  if (tid == 0) { v->playback_pos = rd + (uint64_t)n; if (ended && rd + (uint64_t)n >= fed) v->finished = 1; }
  __syncthreads();
  return n;
}

// CubicInterpolator::process for ONE channel of interleaved staging buffers (cubic.rs:36-114): the exact serial recurrence, one lane.
DEVO void outer_cubic_channel(PgVoice* v, int ch, int C, const float* in, int in_samples, float* out, int out_samples, int* consumed_samples, int* produced_samples) {
  const int num_in = in_samples / C, num_out = out_samples / C;
  int num_consumed = 0, num_produced = 0;
  const float ratio = v->outer_ratio;
  float i0 = v->outer_input[ch][0], i1 = v->outer_input[ch][1], i2 = v->outer_input[ch][2], i3 = v->outer_input[ch][3];
  float sub_pos = v->outer_sub_pos[ch];
  int initialized = v->outer_init[ch];
#define OUTER_PUSH(x) do { i3 = i2; i2 = i1; i1 = i0; i0 = (x); } while (0)
  if (!initialized && num_in >= 3) {  // cubic.rs:61-69
    initialized = 1;
    for (int f = 0; f < 3; ++f) { OUTER_PUSH(in[f * C + ch]); num_consumed += 1; }
  }
  if (ratio < 1.0f) {  // cubic.rs:72-90
    while (num_produced < num_out) {
      if (sub_pos >= 1.0f) {
        if (num_consumed >= num_in) break;
        OUTER_PUSH(in[num_consumed * C + ch]);
        num_consumed += 1;
        sub_pos -= 1.0f;
      }
      out[num_produced * C + ch] = cubic_interp(i3, i2, i1, i0, sub_pos);
      num_produced += 1;
      sub_pos += ratio;
    }
  } else {  // cubic.rs:92-111
    bool brk = false;
    while (num_produced < num_out && !brk) {
      while (sub_pos < ratio) {
        if (num_consumed >= num_in) { brk = true; break; }
        OUTER_PUSH(in[num_consumed * C + ch]);
        num_consumed += 1;
        sub_pos += 1.0f;
      }
      if (brk) break;
      sub_pos -= ratio;
      out[num_produced * C + ch] = cubic_interp(i3, i2, i1, i0, 1.0f - sub_pos);
      num_produced += 1;
    }
  }
#undef OUTER_PUSH
  v->outer_input[ch][0] = i0; v->outer_input[ch][1] = i1; v->outer_input[ch][2] = i2; v->outer_input[ch][3] = i3;
  v->outer_sub_pos[ch] = sub_pos; v->outer_init[ch] = initialized;
  *consumed_samples = num_consumed * C; *produced_samples = num_produced * C;
}

// The same resampler call for all channels at once. The channels' accumulators are equal bit for bit (one ratio, the same input and output
// counts per call), so lane 0 replays the schedule once — the f32 `sub_pos` recurrence and the consumed count of every output frame, no
// sample touched — and every lane evaluates Hermite outputs from the staged input (window position c = the history's four frames followed
// by the consumed input frames, as in src_write_buffer). The one-lane-per-channel walk above pays a dependent global read per input frame:
// 0.55 ms per 1024-frame block of a ResampledSource-backed voice. Returns false (nothing done) when the channels' states differ.
DEVO bool outer_cubic_parallel(PgVoice* v, int C, const float* in, int in_samples, float* out, int out_samples, const SrcScratch& S) {
  const int tid = pg_tid(), nt = blockDim.x;
  const int num_in = in_samples / C, num_out = out_samples / C;
  if (C > 2 || num_out > SRC_OUT_CAP) return false;
  if (C == 2 && (__float_as_uint(v->outer_sub_pos[0]) != __float_as_uint(v->outer_sub_pos[1]) || v->outer_init[0] != v->outer_init[1])) return false;
  __syncthreads();
  if (tid == 0) {
    int num_consumed = 0, num_produced = 0;
    const float ratio = v->outer_ratio;
    float sub_pos = v->outer_sub_pos[0];
    int initialized = v->outer_init[0];
    if (!initialized && num_in >= 3) { initialized = 1; num_consumed = 3; }  // cubic.rs:61-69
    if (ratio < 1.0f) {  // cubic.rs:72-90
      while (num_produced < num_out) {
        if (sub_pos >= 1.0f) {
          if (num_consumed >= num_in) break;
          num_consumed += 1;
          sub_pos -= 1.0f;
        }
        S.sched_c[num_produced] = (uint16_t)num_consumed;
        S.sched_f[num_produced] = sub_pos;
        num_produced += 1;
        sub_pos += ratio;
      }
    } else {  // cubic.rs:92-111
      bool brk = false;
      while (num_produced < num_out && !brk) {
        while (sub_pos < ratio) {
          if (num_consumed >= num_in) { brk = true; break; }
          num_consumed += 1;
          sub_pos += 1.0f;
        }
        if (brk) break;
        sub_pos -= ratio;
        S.sched_c[num_produced] = (uint16_t)num_consumed;
        S.sched_f[num_produced] = 1.0f - sub_pos;
        num_produced += 1;
      }
    }
    S.ctl[0] = num_consumed * C; S.ctl[1] = num_produced * C;
    S.ctl[2] = initialized; S.ctl[3] = (int)__float_as_uint(sub_pos);
  }
  __syncthreads();
  const int consumed = S.ctl[0] / C, produced = S.ctl[1] / C;
  // window element at position p of channel ch: the history (oldest first) in front of the consumed input frames
  auto W = [&](int p, int ch) -> float { return p < 4 ? v->outer_input[ch][3 - p] : in[(p - 4) * C + ch]; };
  for (int i = tid; i < produced * C; i += nt) {
    const int k = C == 2 ? (i >> 1) : i, ch = C == 2 ? (i & 1) : 0;
    const int c = S.sched_c[k];
    out[i] = cubic_interp(W(c, ch), W(c + 1, ch), W(c + 2, ch), W(c + 3, ch), S.sched_f[k]);
  }
  float h[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (tid < C) for (int j = 0; j < 4; ++j) h[j] = W(consumed + 3 - j, tid);   // newest first
  __syncthreads();
  if (tid < C) {
    for (int j = 0; j < 4; ++j) v->outer_input[tid][j] = h[j];
    v->outer_sub_pos[tid] = __uint_as_float((uint32_t)S.ctl[3]);
    v->outer_init[tid] = S.ctl[2];
  }
  return true;
}

// ResampledSource::write (src/source/resampled.rs:101-152) around the file source: the voice's PreloadedFileSource runs at
// `out_rate` != the mixer's rate (pg_voice_options::source_rate), ConvertedSource puts a cubic ResampledSource behind it
// (converted.rs:15-45). Two TempBuffers of 512 frames (buffer.rs:499-610) in device memory, ranges in samples; the input range is NOT
// shrunk to what the source delivered (only resamplers with a required input size pad, :120-127), so an exhausted source leaves a
// stale tail that is resampled like the reference does. `frames` frames of the file layout into `out`; returns frames written.
template <bool GLIDE>
DEVO int resampled_source_write(PgVoice* v, float* out, int frames, int pending_stop, const SrcScratch& S) {
  const int tid = pg_tid(), nt = blockDim.x;
  const int C = (int)v->channels;
  const int cap = 512 * C;  // DEFAULT_CHUNK_SIZE * channel_count
  const int out_len = frames * C;
  int total_written = 0;
  __syncthreads();
  if (tid == 0 && pending_stop) v->outer_pending_stop = 1;  // FilePlaybackMessage::Stop waits in the file source's queue for its next write
  __syncthreads();
  while (total_written < out_len) {
    if (v->out_start >= v->out_end) {  // output_buffer.is_empty()
      if (v->in_start >= v->in_end) {  // input_buffer.is_empty(): fetch new input from the source
        const int stop = v->outer_pending_stop;
        __syncthreads();
        if (tid == 0) { v->in_start = 0; v->in_end = (uint32_t)cap; v->outer_pending_stop = 0; }
        __syncthreads();
        int post_on;
        if (v->stream_on) (void)stream_source_write(v, v->stage_in, 512, stop);
        else (void)file_source_write<GLIDE>(v, v->stage_in, 512, stop, S, nullptr, false, &post_on);
        __threadfence_block();
        __syncthreads();
      }
      // resampler.process(input_buffer.get(), output_buffer.get_mut()): channels are independent recurrences, one lane each;
      // (consumed, written) of the LAST channel count (cubic.rs:179-186)
#ifdef PG_NO_OUTER_PARALLEL   // (diagnostic builds: the one-lane-per-channel walk only)
      if (tid < C) {
#else
      if (!outer_cubic_parallel(v, C, v->stage_in + v->in_start, (int)(v->in_end - v->in_start), v->stage_out, cap, S) && tid < C) {
#endif
        int consumed, produced;
        outer_cubic_channel(v, tid, C, v->stage_in + v->in_start, (int)(v->in_end - v->in_start), v->stage_out, cap, &consumed, &produced);
        if (tid == C - 1) { S.ctl[0] = consumed; S.ctl[1] = produced; }
      }
      __threadfence_block();
      __syncthreads();
      const int consumed = S.ctl[0], produced = S.ctl[1];
      __syncthreads();
      if (tid == 0) { v->in_start += (uint32_t)consumed; v->out_start = 0; v->out_end = (uint32_t)produced; }
      __syncthreads();
      if (v->finished && produced == 0) break;  // source and resampler produced no more output
    }
    const int avail = (int)(v->out_end - v->out_start);
    const int n = out_len - total_written < avail ? out_len - total_written : avail;
    for (int i = tid; i < n; i += nt) out[total_written + i] = v->stage_out[v->out_start + i];
    __syncthreads();
    if (tid == 0) v->out_start += (uint32_t)n;
    __syncthreads();
    total_written += n;
  }
  return total_written / C;
}

// PreloadedFileSource::write [+ ResampledSource::write] + ChannelMappedSource::write (mapped.rs:61-99) +
// AmplifiedSource::write (amplified.rs:93-104) + PannedSource::write (panned.rs:93-104).
// Renders `frames` stereo output frames into `out` (LDS, 2*frames floats); returns stereo samples written.
// ADAPTERS: 0 = the kernel variant never sees a ResampledSource-backed or host-fed voice (the staged kernels: the host keeps such units out);
// 1 = host-fed voices only (the four-per-CU fast kernel: a graph with a ResampledSource takes the wide kernel instead); 2 = both.
template <bool GLIDE, int ADAPTERS>
DEVO int voice_write(PgVoice* v, float* out, int frames, int pending_stop, const SrcScratch& S, float* acc, int* added, bool ask_ends = true) {
  *added = 0;
  const int tid = pg_tid(), nt = blockDim.x;
  const int C = (int)v->channels;
  int wf;
  if (ADAPTERS >= 1 && (v->stream_on || (ADAPTERS == 2 && v->outer_on))) {
    // ChannelMappedSource::write (mapped.rs:61-99) asks its source again, inside the same call, for what a short write left open, until
    // the block is full or a write returns nothing; only then does the mixer look at is_exhausted. A ResampledSource that broke off because
    // its source is finished and its resampler had no output for the last input frames starts over on the second request — it refills its
    // input range (stale) and plays on. With equal channel counts the mapper passes the one call through (:62-64).
    const bool outer = ADAPTERS == 2 && v->outer_on;
    wf = 0;
    int stop = pending_stop;  // (the Stop message is queued once, in front of the call)
    for (;;) {
      const int r = outer ? resampled_source_write<GLIDE>(v, out + wf * C, frames - wf, stop, S) : stream_source_write(v, out + wf * C, frames - wf, stop);
      stop = 0;
      wf += r;
      if (C == 2 || r == 0 || wf >= frames) break;
    }
  } else {
    int post_on;
    wf = file_source_write<GLIDE>(v, out, frames, pending_stop, S, acc, true, &post_on, ask_ends);
    if (post_on) { *added = acc ? 1 : 0; return wf * 2; }  // (stereo samples: a fused mono voice has been mapped to both channels on the way)
  }
  __syncthreads();
  // ChannelMappedSource: mono -> stereo (buffer.rs:209-217)
  if (C == 1) {
    // in-place expansion runs back to front, one tile of blockDim.x frames at a time, staged through registers
    for (int base = wf > 0 ? ((wf - 1) / nt) * nt : -1; base >= 0; base -= nt) {
      int f = base + tid;
      float x = (f < wf) ? out[f] : 0.0f;
      __syncthreads();
      if (f < wf) { out[2 * f] = x; out[2 * f + 1] = x; }
      __syncthreads();
    }
  }
  int written = wf * 2;
  __syncthreads();
  // apply_smoothed_gain  smoothing.rs:60-71  (ramp path multiplies per SAMPLE)
  // While a smoother moves, one lane lays out its value sequence — the serial loop's own sm_next, an f32 recurrence that must be walked — into
  // the source stage's scratch (free once the source has written: fraction table, position map and input window are contiguous), tile by tile,
  // and all lanes apply it: the reference's per-sample / per-frame multiplications, same values, same order per sample. (Until round 5 one lane
  // also did the multiplications, a read-modify-write of LDS per sample with two square roots per frame for the panning: 0.1-0.35 ms per block
  // for every voice whose volume or panning had been touched in the last 1500 frames — and every unit's block waits for the slowest unit.)
  float* const seq = S.sched_f;
  constexpr int SEQ_CAP = SRC_OUT_CAP + SRC_WIN_CAP + 2 * (SRC_WIN_CAP + 4);   // floats from sched_f to the end of the input window
  if (sm_need_ramp(v->volume)) {
    for (int base = 0; base < written; base += SEQ_CAP) {
      const int n = written - base < SEQ_CAP ? written - base : SEQ_CAP;
      if (tid == 0) { PgSmooth s = v->volume; sm_sequence(s, seq, n); v->volume = s; }
      __syncthreads();
      for (int i = tid; i < n; i += nt) out[base + i] *= seq[i];
      __syncthreads();
    }
  } else {
    float gain = v->volume.target;
    if (fabsf(1.0f - gain) > 0.000001f) for (int i = tid; i < written; i += nt) out[i] = out[i] * gain;
  }
  __syncthreads();
  // apply_smoothed_panning  smoothing.rs:74-122
  if (sm_need_ramp(v->panning)) {
    const int frames_w = written / 2;
    for (int base = 0; base < frames_w; base += SEQ_CAP) {
      const int n = frames_w - base < SEQ_CAP ? frames_w - base : SEQ_CAP;
      if (tid == 0) { PgSmooth s = v->panning; sm_sequence(s, seq, n); v->panning = s; }
      __syncthreads();
      for (int f = tid; f < n; f += nt) { float l, r; panning_factors(seq[f], l, r); out[2 * (base + f)] *= l; out[2 * (base + f) + 1] *= r; }
      __syncthreads();
    }
  } else {
    float pan = v->panning.target;
    if (fabsf(pan) > 0.000001f) {
      float l, r;
      panning_factors(pan, l, r);
      for (int i = tid; i < written; i += nt) out[i] *= (i & 1) ? r : l;
    }
  }
  __syncthreads();
  return written;
}

// MixedSource::process_sources for ONE playing source (src/source/mixed.rs:558-624): renders into `tmp` and adds
// into `sig`. One call of process_sources = one chunk of the mixer, rendered as pieces: `chunk_first` names the chunk's first piece,
// `chunk_end` the position at which the chunk ends. Returns bit 0: the source produced output in this piece; bit 1 (first piece only): it
// has not started yet but will inside this chunk — `audible_input` of the chunk (mixed.rs:696-706) is decided at the first piece.
template <bool GLIDE, int ADAPTERS = 2>
DEVO int voice_process(PgVoice* gv, PgVoice* lv /*LDS*/, float* sig, float* tmp, int frames, uint64_t pos, const SrcScratch& S0,
                       const PgSchedEntry* sched, int sched_bank, bool have_word = false, uint32_t word = 0, uint64_t call_end = 0, bool chunk_first = true,
                       uint64_t chunk_end = 0, bool in_lds = false) {
  SrcScratch S = S0;
  const int tid = pg_tid(), nt = blockDim.x;
  static_assert(sizeof(PgVoice) / 4 <= 256, "one dword per lane");
  __syncthreads();
  if (!in_lds) {  // stage the voice state into LDS (uniform reads, lane-0 writes); `word` = this lane's dword when the caller prefetched it
    const uint32_t* src = (const uint32_t*)gv;      // (in_lds: the copy is what the launch's previous block left there)
    uint32_t* dst = (uint32_t*)lv;
    if (have_word) { if (tid < (int)(sizeof(PgVoice) / 4)) dst[tid] = word; }
    else for (int i = tid; i < (int)(sizeof(PgVoice) / 4); i += nt) dst[i] = src[i];
  }
  __syncthreads();
  S.sched_rd = (sched && lv->sched_class >= 0) ? sched + (size_t)lv->sched_class * 2 + sched_bank : nullptr;
  PG_STAMP(S.diag, 16);
  // A source that broke out of its chunk's loop (nothing written, or exhausted: mixed.rs:612-620) is left alone until the mixer's next chunk
  const bool skip = lv->chunk_skip != 0 && !chunk_first;
  if (lv->chunk_skip != 0 && chunk_first) { __syncthreads(); if (tid == 0) { lv->chunk_skip = 0; gv->chunk_skip = 0; } __syncthreads(); }
  // (an exhausted ResampledSource-backed voice is still asked for the rest of the write in which it ran out: PgVoice::zombie_end)
  if (!lv->active && !(ADAPTERS == 2 && pos < lv->zombie_end)) return 0;
  if (skip) return 0;
  const int out_len = frames * 2;
  int total_written = 0;
  if (lv->start_time > pos) {
    uint64_t fu = lv->start_time - pos;
    if (fu >= (uint64_t)frames) {
      // not in this piece — but a source that starts in a later piece of the chunk will write there (a fresh file source always has frames; a
      // host-fed one only if its ring holds some)
      const bool will = chunk_first && lv->start_time < chunk_end && lv->active && !lv->finished &&
                        (!lv->stream_on || (lv->stream_fed & 0x7fffffffffffffffull) > lv->playback_pos);
      return will ? 2 : 0;
    }
    total_written = (int)fu * 2;
  }
  bool produced_output = false;
  while (total_written < out_len) {
    uint64_t source_time = pos + (uint64_t)(total_written / 2);
    uint64_t samples_until_stop = PG_USIZE_MAX;
    if (lv->has_stop) {
      uint64_t d = lv->stop_time > source_time ? lv->stop_time - source_time : 0;
      samples_until_stop = d * 2;
    }
    int pending_stop = 0;
    if (samples_until_stop == 0) {
      pending_stop = 1;
      __syncthreads();
      if (tid == 0) lv->has_stop = 0;
      samples_until_stop = PG_USIZE_MAX;
    }
    uint64_t remaining = (uint64_t)(out_len - total_written);
    if (samples_until_stop < remaining) remaining = samples_until_stop;
    int to_write = (int)(remaining < 8192 ? remaining : 8192);
    // (a call that reaches the end of this piece without reaching a stop time or the end of the chunk goes on in the next piece)
    const bool ask_ends = pos + (uint64_t)frames >= chunk_end || samples_until_stop <= (uint64_t)(out_len - total_written);
    int added;
    int written = voice_write<GLIDE, ADAPTERS>(lv, tmp, to_write / 2, pending_stop, S, sig + total_written, &added, ask_ends);
    if (!added) for (int i = tid; i < written; i += nt) sig[total_written + i] = sig[total_written + i] + tmp[i];  // add_buffers
    __syncthreads();
    total_written += written;
    produced_output |= written > 0;
    // is_transient && is_exhausted (ResampledSource: the source is exhausted AND both staging buffers are empty, resampled.rs:162-164)
    const bool exhausted = lv->outer_on ? (lv->finished && lv->in_start >= lv->in_end && lv->out_start >= lv->out_end) : lv->finished != 0;
    if (!ask_ends && written == to_write) {
      // the call filled this piece and goes on in the chunk's next one: the mixer looks at the source (is_exhausted, written == 0) when the
      // call returns, not here — a ResampledSource whose staging buffers happen to be empty at this frame refills them inside the same call
    } else if (exhausted && !lv->persistent) {
      if (tid == 0) { if (ADAPTERS == 2 && lv->active && lv->outer_on) lv->zombie_end = call_end; lv->active = 0; lv->chunk_skip = 1; }
      break;
    } else if (written == 0) { if (tid == 0) lv->chunk_skip = 1; break; }
  }
  __syncthreads();
  {  // write the voice state back
    uint32_t* dst = (uint32_t*)gv;
    const uint32_t* src = (const uint32_t*)lv;
    for (int i = tid; i < (int)(sizeof(PgVoice) / 4); i += nt) dst[i] = src[i];
  }
  __syncthreads();
  return produced_output ? 1 : 0;
}

// The f32 schedule recurrence of one piece (cubic.rs:72-90 with ratio < 1): identical operations to the replay in
// src_write_buffer. One lane; results go to `oc` / `of` (LDS or global).
template <typename C16, typename F32>
DEVO void sched_replay(float ratio, float& sp, int& cc, int piece, C16* oc, F32* of) {
  asm volatile("" : "+v"(sp), "+v"(cc));
  for (int k = 0; k < piece; ++k) {
    const bool ge = sp >= 1.0f;
    cc += ge ? 1 : 0;
    sp = ge ? sp - 1.0f : sp;
    oc[k] = (uint16_t)cc;
    of[k] = sp;
    sp += ratio;
  }
}

// Schedule cache for the ratios the time-parallel schedule does not cover (ratio < 0.5): the class representative publishes the
// schedule of the NEXT block (same piece length assumed) into the other bank, from its own state after this block; the other
// voices of the class (same f32 ratio, same sub_pos) copy it instead of replaying it. Runs on lane 0 after the unit's audio has
// been written.
DEVO void sched_publish(const PgVoice* gv, PgSchedEntry* sched, int sched_bank, int piece) {
  if (!sched || gv->sched_class < 0 || !gv->sched_rep) return;
  if (gv->sched_hit == 2) return;  // the class runs the time-parallel schedule: nobody reads the cache
  PgSchedEntry* e = sched + (size_t)gv->sched_class * 2 + (sched_bank ^ 1);
  const float ratio = gv->ratio;
  const bool ok = gv->active && !gv->finished && gv->initialized[0] && ratio < 1.0f && fabsf(ratio - 1.0f) >= 0.000001f && piece >= 1 && piece <= PG_SCHED_CAP;
  if (!ok) { e->valid = 0; return; }
  float sp = gv->sub_pos[0];
  int cc = 0;
  e->ratio_bits = __float_as_uint(ratio);
  e->subpos_in_bits = __float_as_uint(sp);
  e->piece = piece;
  sched_replay(ratio, sp, cc, piece, e->sched_c, e->sched_f);
  e->subpos_out_bits = __float_as_uint(sp);
  e->c_total = cc;
  e->valid = 1;
}

}  // namespace pgd
