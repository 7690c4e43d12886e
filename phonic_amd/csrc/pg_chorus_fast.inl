// Time-parallel steady-state ChorusEffect (reference src/effect/chorus.rs:311-394) for one workgroup.
//
// With no parameter ramping the per-frame control values are block constants; the pre-filter (SVF) sees only the input; the
// two LFOs only depend on their f32 phase accumulators; and the feedback runs through the delay line, which frame n reads
// `2 + delay + (1 + lfo) * depth` frames behind the write head. Inside a chunk of T <= floor(2 + delay_in_samples) - 2 frames
// every line read therefore hits pre-chunk data. Per piece (<= 1024 frames):
//   0. LFO phases of every frame: the f32 accumulation  p += inc; if (p >= 1) p -= 1  is piecewise an exact arithmetic
//      progression (per binade, see f32_phase_advance); one lane per LFO lists the pieces, all lanes fill the phase arrays
//   1. SVF over the piece's input by the blocked scan
//   2. per chunk: all (frame, channel) items evaluate the LFO (parabolic sine), read and interpolate their two taps; barrier;
//      all items write  filtered + out * feedback  into the line and mix the output.
// Same arithmetic as the serial loop up to the f64 rounding of the SVF scan.

struct ChorusPiece { int k0; float p0; double du; };  // phases of frames k0 .. next piece: p0 + (k - k0) * du (exact in f64)
constexpr int CHORUS_PIECE_CAP = 56;

// One lane: lists the exact pieces of up to `steps` phase updates starting at p, as many as fit the piece table. Returns the number of
// pieces; *covered = frames described by them (== steps unless the table filled up); p = the phase after `*covered` updates.
DEVO int chorus_phase_pieces(float& p, float d, int steps, ChorusPiece* rec, int* covered) {
  int n_rec = 0, k = 0;
  while (k < steps && n_rec < CHORUS_PIECE_CAP) {
    const uint32_t bits = __float_as_uint(p);
    const int e = (int)((bits >> 23) & 0xff);
    int m = 0;
    double du = 0.0;
    if (e > 24 && e < 0xff && p > 0.0f && d > 0.0f && p < 1.0f) {
      const double u = __longlong_as_double((long long)((unsigned long long)(e - 127 - 23 + 1023) << 52));
      const double D = (double)d / u;
      const double Dr = rint(D);
      if (D < 16777216.0 && Dr >= 1.0 && D - floor(D) != 0.5) {
        const double top = __longlong_as_double((long long)((unsigned long long)(e - 127 + 1 + 1023) << 52));
        const double q = floor(((top - u) - (double)p) / (Dr * u));
        if (q >= 1.0) { m = q > (double)(steps - k) ? steps - k : (int)q; du = Dr * u; }
      }
    }
    rec[n_rec].k0 = k; rec[n_rec].p0 = p; rec[n_rec].du = du;
    ++n_rec;
    if (m > 0) { p = (float)((double)p + (double)m * du); k += m; }   // frames k .. k+m-1 read p0 + j*du; the next piece starts at the value after m steps
    else { p += d; if (p >= 1.0f) p -= 1.0f; k += 1; }                // the plain hardware step: this piece covers one frame
  }
  *covered = k;
  return n_rec;
}
// the phase frame k reads (any lane): its piece is found by a walk over the (few, LDS-resident) piece records
DEVO float chorus_phase_at(const ChorusPiece* r, int nr, int k) {
  int i = 0;
  while (i + 1 < nr && r[i + 1].k0 <= k) ++i;
  return (float)((double)r[i].p0 + (double)(k - r[i].k0) * r[i].du);
}

DEVO bool chorus_fast_eligible(const PgFx& fx) {
  const PgChorus& c = fx.u.chorus;
  if (sm_need_ramp(c.rate) || sm_need_ramp(c.phase) || sm_need_ramp(c.depth) || sm_need_ramp(c.feedback) || sm_need_ramp(c.delay) ||
      sm_need_ramp(c.wet) || sm_need_ramp(c.freq) || sm_need_ramp(c.res))
    return false;
  const float delay_in_samples = c.delay.target * (float)fx.sample_rate * 0.001f;
  const float depth_in_samples = c.lfo_range * c.depth.target;
  if (!(depth_in_samples >= 0.0f)) return false;
  return floorf(2.0f + delay_in_samples) >= 66.0f && (2.0f + delay_in_samples + 2.0f * depth_in_samples) < (float)(c.mask - 8);
}

// The same test made by a whole wave (every wave of the workgroup calls it and gets the same answer): lane l looks at smoother l mod 8 — the
// eight sit side by side in PgChorus — instead of every lane walking all eight one after the other (1.5 K cycles at the head of a chorus whose
// unit's block is a latency chain).
DEVO bool chorus_fast_eligible_wave(const PgFx& fx) {
  const PgChorus& c = fx.u.chorus;
  const PgSmooth* sm = &c.rate;
  static_assert(offsetof(PgChorus, res) - offsetof(PgChorus, rate) == 7 * sizeof(PgSmooth), "the eight smoothers of PgChorus are contiguous");
  if (__ballot(sm_need_ramp(sm[pg_tid() & 7]) ? 1 : 0) != 0ull) return false;
  const float delay_in_samples = c.delay.target * (float)fx.sample_rate * 0.001f;
  const float depth_in_samples = c.lfo_range * c.depth.target;
  if (!(depth_in_samples >= 0.0f)) return false;
  return floorf(2.0f + delay_in_samples) >= 66.0f && (2.0f + delay_in_samples + 2.0f * depth_in_samples) < (float)(c.mask - 8);
}

// ---- ChorusEffect while parameters ramp (chorus.rs:311-394 with any of its eight smoothers moving) ------------------------------------------
// As for the Delay: single lanes lay out the per-frame value sequences of a piece with the serial loop's own calls — delay (spring), depth,
// feedback, wet; the two LFO values incl. update_lfos while rate / phase ramp (chorus.rs:223-231: the oscillators are re-seated on
// `current_phase` every frame for as long as either ramps); the pre-filter's cutoff / resonance while either ramps — then the piece is
// rendered with per-frame values: time-varying SVF scan of the input, taps at the frame's delay position, line writes with the frame's
// feedback, dry / wet with the frame's mix. Chunks are cut for the shortest delay of the frames they hold.
constexpr int CHORUS_RAMP_SEQS = 8;  // delay, depth, feedback, wet, left LFO, right LFO, filter cutoff, filter resonance
DEVO bool chorus_ramp_eligible(const PgFx& fx) {
  const PgChorus& c = fx.u.chorus;
  const float srf = (float)fx.sample_rate;
  const PgSmooth& t = c.delay;
  const float travel = t.kind == SM_SPRING && t.a > 0.0f ? fabsf(t.b) / (t.a * t.comp) : 0.0f;
  const float lo = fmaxf(fminf(t.current, t.target) - travel, 0.0f), hi = fmaxf(t.current, t.target) + travel;
  const float depth_hi = c.lfo_range * fmaxf(c.depth.current, c.depth.target);
  if (!(depth_hi >= 0.0f)) return false;
  return floorf(2.0f + lo * 0.999f * srf * 0.001f) >= 66.0f && (2.0f + hi * 1.001f * srf * 0.001f + 2.0f * depth_hi) < (float)(c.mask - 8);
}
DEVO bool chorus_ramp_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  if (!chorus_ramp_eligible(fx)) return false;
  PgChorus& c = fx.u.chorus;
  const int tid = pg_tid(), nt = blockDim.x;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  int cap = fc.tmp_floats / CHORUS_RAMP_SEQS;
  if (cap > nt) cap = nt;  // two (frame, channel) items per lane at most: the interpolated outputs wait in two registers across the barrier
  if (cap < 8 || nt != 256) return false;
  double* buf = (double*)fc.scratch;
  double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);
  int* red = (int*)(xchg + 4);
  float* seq = fc.tmp;
  float* a_dly = seq, *a_depth = seq + cap, *a_fb = seq + 2 * cap, *a_wet = seq + 3 * cap, *a_ll = seq + 4 * cap, *a_lr = seq + 5 * cap, *a_cut = seq + 6 * cap, *a_res = seq + 7 * cap;
  const float srf = (float)fx.sample_rate, nyq = srf / 2.0f;
  const double srd = (double)fx.sample_rate;
  const uint32_t mask = c.mask;
  const int svf_type = delay_to_svf(c.filter_type);
  for (int p0 = 0; p0 < frames; p0 += cap) {
    const int P = frames - p0 < cap ? frames - p0 : cap;
    float* sp = sig + 2 * p0;
    __syncthreads();
    // 0. the sequences of the piece
    if (tid < 4) {
      PgSmooth& g = tid == 0 ? c.delay : (tid == 1 ? c.depth : (tid == 2 ? c.feedback : c.wet));
      float* dst = seq + tid * cap;
      sm_sequence(g, dst, P);
      if (tid == 2) for (int k = 0; k < P; ++k) dst[k] = clampf(dst[k], -0.999f, 0.999f);
    } else if (tid == 64) {  // update_lfos while rate / phase ramp, then lfo.run() of both oscillators (chorus.rs:327-329,353-354)
      PgSmooth rate = c.rate, phase = c.phase;
      PgLfo o0 = c.osc[0], o1 = c.osc[1];
      const double current_phase = c.current_phase;
      for (int k = 0; k < P; ++k) {
        if (sm_need_ramp(rate) || sm_need_ramp(phase)) {
          const double r = (double)sm_next(rate);
          lfo_set_rate(o0, fx.sample_rate, r); lfo_set_rate(o1, fx.sample_rate, r);
          const double phase_offset = (double)sm_next(phase);
          lfo_set_phase_degrees(o0, (float)current_phase);
          lfo_set_phase_degrees(o1, (float)(current_phase + phase_offset));
        }
        a_ll[k] = lfo_run(o0);
        a_lr[k] = lfo_run(o1);
      }
      c.rate = rate; c.phase = phase; c.osc[0] = o0; c.osc[1] = o1;
    } else if (tid == 128) {  // the pre-filter's parameters: next() of both while either ramps (chorus.rs:332-345), else the set in place
      PgSmooth fr = c.freq, rs = c.res;
      float cut = c.coef.cutoff, res = c.coef.resonance;
      for (int k = 0; k < P; ++k) {
        if (sm_need_ramp(fr) || sm_need_ramp(rs)) { cut = clampf(sm_next(fr), 20.0f, nyq); res = sm_next(rs); }
        a_cut[k] = cut; a_res[k] = res;
      }
      c.freq = fr; c.res = rs;
    }
    // 1. pre-filter over the piece with the frame's coefficients (svf.rs:137-168,211-222)
    for (int s = tid; s < 2 * P; s += nt) buf[REV_IDX(s >> 1, s & 1)] = (double)sp[s];
    __syncthreads();
    {
      auto coef = [&](int n, double& a1, double& a2, double& a3, double& m0, double& m1, double& m2) {
        const double g = tan(F64_PI * (double)a_cut[n] / srd);
        const double kq = fmax(2.0 * (1.0 - (double)a_res[n] * 0.97), 0.03);
        if (svf_type == 0) { m0 = 0.0; m1 = 0.0; m2 = 1.0; } else if (svf_type == 2) { m0 = 0.0; m1 = 1.0; m2 = 0.0; } else { m0 = 1.0; m1 = -kq; m2 = -1.0; }
        a1 = 1.0 / (1.0 + g * (g + kq));
        a2 = g * a1;
        a3 = g * a2;
      };
      svf_scan_time_varying<false>(coef, c.flt, buf, P, xchg);
      if (tid == 0) svf_set(c.coef, svf_type, fx.sample_rate, a_cut[P - 1], a_res[P - 1]);
    }
    __syncthreads();
    // 2. chunks
    int done = 0;
    while (done < P) {
      int T = P - done;
      for (;;) {  // shorter than the shortest delay position of the frames it holds (chorus.rs:356-357: 2 + delay + (1 + lfo) * depth, lfo >= -1 up to the parabola's overshoot)
        __syncthreads();
        if (tid == 0) red[0] = 0x7fffffff;
        __syncthreads();
        int m = 0x7fffffff;
        for (int k = tid; k < T; k += nt) { const int f = (int)floorf(2.0f + a_dly[done + k] * srf * 0.001f); m = m < f ? m : f; }
        if (m != 0x7fffffff) atomicMin(&red[0], m);
        __syncthreads();
        const int t_max = red[0] - 4;
        if (T <= t_max) break;
        T = t_max < 1 ? 1 : t_max;
      }
      const uint32_t wp0[2] = {c.write_pos[0], c.write_pos[1]};
      float outv[2] = {0.0f, 0.0f};
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s = tid + it * 256;
        if (s < 2 * T) {
          const int nn = done + (s >> 1), ch = s & 1;
          const float delay_in_samples = a_dly[nn] * srf * 0.001f;
          const float depth_in_samples = c.lfo_range * a_depth[nn];
          const float lfo = ch == 0 ? a_ll[nn] : a_lr[nn];
          const float delay_pos = 2.0f + delay_in_samples + (1.0f + lfo) * depth_in_samples;
          const gdouble* line = (const gdouble*)c.line[ch];
          const uint32_t wp = (wp0[ch] + (uint32_t)(s >> 1)) & mask;
          const double read_pos = (double)wp - (double)delay_pos;
          const double read_pos_floor = floor(read_pos);
          const double fraction = read_pos - read_pos_floor;
          const long long index1 = (long long)read_pos_floor;
          const uint32_t i1 = (uint32_t)((unsigned long long)index1 & (unsigned long long)mask);
          const uint32_t i2 = (uint32_t)((unsigned long long)(index1 + 1) & (unsigned long long)mask);
          if (fc.idx_log) fc.idx_log[(p0 + nn) * 2 + ch] = (int32_t)i1;
          const double v1 = line[i1], v2 = line[i2];
          const float out = (float)(v1 + (v2 - v1) * fraction);
          outv[it] = out;
          const int bi = REV_IDX(nn, ch);
          buf[bi] = (double)(float)buf[bi] + (double)out * (double)a_fb[nn];   // what this frame writes into the line
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s = tid + it * 256;
        if (s < 2 * T) {
          const int nn = done + (s >> 1), ch = s & 1;
          ((gdouble*)c.line[ch])[(wp0[ch] + (uint32_t)(s >> 1)) & mask] = buf[REV_IDX(nn, ch)];
          const float wet = a_wet[nn];
          sp[2 * nn + ch] = sp[2 * nn + ch] * (1.0f - wet) + outv[it] * wet;
        }
      }
      __syncthreads();
      if (tid == 0) { c.write_pos[0] = (wp0[0] + (uint32_t)T) & mask; c.write_pos[1] = (wp0[1] + (uint32_t)T) & mask; }
      __syncthreads();
      done += T;
    }
  }
  // (the call-end phase bookkeeping of chorus.rs:388-393 runs in the caller, once per process call: chorus_call_end)
  __syncthreads();
  return true;
}

DEVO bool chorus_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  if (!chorus_fast_eligible_wave(fx)) return false;
  PgChorus& c = fx.u.chorus;
  const int tid = pg_tid(), nt = blockDim.x;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  double* buf = (double*)fc.scratch;                       // [T][2] f64, skewed (REV_IDX)
  char* lp = fc.scratch + REV_BUF_DOUBLES * 8;
  double* xchg = (double*)lp;                 lp += 4 * 8;
  PgBiquadCoef* lco = (PgBiquadCoef*)lp;      lp += sizeof(PgBiquadCoef);
  ChorusPiece* rec = (ChorusPiece*)lp;        lp += 2 * CHORUS_PIECE_CAP * sizeof(ChorusPiece);
  int* pctl = (int*)lp;                       lp += 32;   // [osc]: pieces, covered frames; then the phases after the piece (f32 bits)
  // [frame][2] interpolated line output: in the free `tmp` rows of the unit (2 floats per frame of the block) — 8 KB less arena, which
  // takes a Filter -> Chorus unit from 55 KB to 46 KB of LDS: three workgroups per CU instead of two
  float* o32 = fc.tmp_floats >= 2 * (frames < 1024 ? frames : 1024) ? fc.tmp : nullptr;
  if (!o32) return false;
  static_assert(REV_BUF_DOUBLES * 8 + 32 + sizeof(PgBiquadCoef) + 2 * CHORUS_PIECE_CAP * sizeof(ChorusPiece) + 32 <= FAST_SCRATCH_CHORUS_BYTES,
                "chorus fast path: LDS arena too small");
  const float srf = (float)fx.sample_rate;
  const float delay_ms = c.delay.target, depth = c.depth.target;
  const float feedback = clampf(c.feedback.target, -0.999f, 0.999f);
  const float wet_amount = c.wet.target, dry_amount = 1.0f - c.wet.target;
  const float delay_in_samples = delay_ms * srf * 0.001f;
  const float depth_in_samples = c.lfo_range * depth;
  const int t_max = (int)floorf(2.0f + delay_in_samples) - 4;  // margin: the parabolic sine may overshoot -1 by a hair
  const uint32_t mask = c.mask;
  __syncthreads();
  if (tid == 0) {
    PgBiquadCoef b;
    b.a1 = c.coef.a1; b.a2 = c.coef.a2; b.a3 = c.coef.a3;
    if (c.coef.type == 0) { b.m0 = 0.0; b.m1 = 0.0; b.m2 = 1.0; }
    else if (c.coef.type == 2) { b.m0 = 0.0; b.m1 = 1.0; b.m2 = 0.0; }
    else { b.m0 = 1.0; b.m1 = -c.coef.k; b.m2 = -1.0; }
    *lco = b;
  }
  for (int p0 = 0; p0 < frames;) {
    int P = frames - p0 < 1024 ? frames - p0 : 1024;
    float* sp = sig + 2 * p0;
    __syncthreads();
    PG_STAMP(fc.diag, 24);
    // 0. LFO phases (lfo.run() once per frame and oscillator, chorus.rs:353-354): one lane per oscillator lists the exact pieces; the piece
    // of the signal ends where the shorter of the two lists ends (a full table: only at phase increments far above the 10 Hz the rate allows)
    for (int pass = 0; pass < 2; ++pass) {
      if (tid == 0 || tid == 64) {
        const int o = tid >> 6;
        float p = c.osc[o].phase;
        pctl[2 * o] = chorus_phase_pieces(p, c.osc[o].phase_inc, P, rec + o * CHORUS_PIECE_CAP, &pctl[2 * o + 1]);
        pctl[4 + o] = (int)__float_as_uint(p);
      }
      __syncthreads();
      const int covered = pctl[1] < pctl[3] ? pctl[1] : pctl[3];
      __syncthreads();
      if (covered >= P) break;
      P = covered;  // (second pass: both lists describe exactly P frames)
    }
    if (tid == 0 || tid == 64) c.osc[tid >> 6].phase = __uint_as_float((uint32_t)pctl[4 + (tid >> 6)]);
    PG_STAMP(fc.diag, 25);
    // 1. pre-filter over the piece (svf.rs:211-222)
    for (int s = tid; s < 2 * P; s += nt) buf[REV_IDX(s >> 1, s & 1)] = (double)sp[s];
    __syncthreads();
    rev_biquad_scan(*lco, c.flt, buf, P, xchg);
    __syncthreads();
    PG_STAMP(fc.diag, 26);
    // 2. chunks
    int done = 0;
    while (done < P) {
      int T = P - done;
      if (T > t_max) T = t_max;
      const uint32_t wp0[2] = {c.write_pos[0], c.write_pos[1]};
      __syncthreads();
      // two (frame, channel) items per lane and trip: the four line taps of both are in flight together (a trip's loads used to wait for the
      // previous trip's LDS stores: 4.5 dependent HBM round trips per chunk on a kernel that is a latency chain, profiles/r03_c3_stamps.txt)
      for (int s0 = tid; s0 < 2 * T; s0 += 2 * nt) {
        double v1[2], v2[2], fraction[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int s = s0 + k * nt;
          v1[k] = 0.0; v2[k] = 0.0; fraction[k] = 0.0;
          if (s < 2 * T) {
            const int nn = done + (s >> 1), ch = s & 1;
            PgLfo l; l.phase = chorus_phase_at(rec + ch * CHORUS_PIECE_CAP, pctl[2 * ch], nn); l.phase_inc = 0.0f; l.waveform = c.osc[ch].waveform;
            const float lfo = lfo_value(l);
            const float delay_pos = 2.0f + delay_in_samples + (1.0f + lfo) * depth_in_samples;
            const gdouble* line = (const gdouble*)c.line[ch];
            const uint32_t wp = (wp0[ch] + (uint32_t)(s >> 1)) & mask;
            const double read_pos = (double)wp - (double)delay_pos;
            const double read_pos_floor = floor(read_pos);
            fraction[k] = read_pos - read_pos_floor;
            const long long index1 = (long long)read_pos_floor;
            const uint32_t i1 = (uint32_t)((unsigned long long)index1 & (unsigned long long)mask);
            const uint32_t i2 = (uint32_t)((unsigned long long)(index1 + 1) & (unsigned long long)mask);
            if (fc.idx_log) fc.idx_log[(p0 + nn) * 2 + ch] = (int32_t)i1;  // test hook: read_idx1 (dsp/delay.rs:120-133)
            v1[k] = line[i1]; v2[k] = line[i2];
          }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int s = s0 + k * nt;
          if (s < 2 * T) {
            const int nn = done + (s >> 1), ch = s & 1;
            const float out = (float)(v1[k] + (v2[k] - v1[k]) * fraction[k]);
            o32[2 * nn + ch] = out;
            const int bi = REV_IDX(nn, ch);
            buf[bi] = (double)(float)buf[bi] + (double)out * (double)feedback;   // what this frame writes into the line
          }
        }
      }
      __syncthreads();
      PG_STAMP(fc.diag, 27 + (done > 0 ? 2 : 0));
      for (int s = tid; s < 2 * T; s += nt) {
        const int nn = done + (s >> 1), ch = s & 1;
        ((gdouble*)c.line[ch])[(wp0[ch] + (uint32_t)(s >> 1)) & mask] = buf[REV_IDX(nn, ch)];
        sp[2 * nn + ch] = sp[2 * nn + ch] * dry_amount + o32[2 * nn + ch] * wet_amount;
      }
      __syncthreads();
      if (tid == 0) { c.write_pos[0] = (wp0[0] + (uint32_t)T) & mask; c.write_pos[1] = (wp0[1] + (uint32_t)T) & mask; }
      __syncthreads();
      PG_STAMP(fc.diag, 28 + (done > 0 ? 2 : 0));
      done += T;
    }
    p0 += P;
  }
  // (the call-end phase bookkeeping of chorus.rs:388-393 runs in the caller, once per process call: chorus_call_end)
  __syncthreads();
  return true;
}
