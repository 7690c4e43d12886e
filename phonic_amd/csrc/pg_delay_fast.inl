// Time-parallel steady-state DelayEffect (reference src/effect/delay.rs:334-454) for one workgroup.
//
// Why this is legal: with the three LFO modulation depths at zero and no parameter ramping, every per-frame control value
// of the reference loop is a block constant (lfo_val * 0 = +-0, powf(2, +-0) = 1), and the only feedback path runs through the
// delay line: frame n reads the line `delay_samples` behind the write head, so inside a chunk of T <= floor(delay_samples) - 1
// frames every line READ hits data written before the chunk. Per chunk:
//   1. all (frame, channel) items read their two taps and interpolate           (InterpolatedDelayLine::process, delay.rs:107-155)
//   2. the wet path  SVF -> saturate -> DC filter -> clamp  over the chunk: the SVF (linear, 2 states) and the DC filter (linear,
//      1 state) by blocked scans, the rational tanh and the clamp element-wise                     (process_feedback, delay.rs:226-237)
//   3. all items write  input + previous frame's wet value * feedback  into the line and mix the output (dry/wet law, M/S width)
// The LFO only advances its f32 phase; f32_phase_advance reproduces T accumulation steps exactly in closed form.
// Same arithmetic as the serial loop up to the f64 rounding of the two scans (the wet value is rounded to f32 as in the reference).

// DcFilter::process_sample (src/utils/dsp/filters/dc.rs:84-88) over the chunk, both channels, in place on the skewed LDS buffer:
//   y_n = (x_n - x_{n-1}) + r * y_{n-1}.  Blocked like the biquad scan: 128 segments of 8 frames per channel.
DEVO void dc_scan(PgDc* dc /*[2]*/, double* buf, int T, double* xchg /* LDS [4] */) {
  const int tid = pg_tid();
  const int wave = tid >> 6, lane = tid & 63;
  const int ch = wave & 1, half = wave >> 1;
  const int seg = half * 64 + lane;
  const int n0 = seg * 8;
  const int len = n0 >= T ? 0 : (T - n0 < 8 ? T - n0 : 8);
  const double r = dc[ch].r;
  double x[8];
  const double x_in = (n0 == 0) ? dc[ch].x1 : (len > 0 ? buf[REV_IDX(n0 - 1, ch)] : 0.0);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = k < len ? buf[REV_IDX(n0 + k, ch)] : 0.0;
  // pass 1: zero-state response of the segment (segment 0 starts from the carried state)
  double y = (seg == 0) ? dc[ch].y1 : 0.0;
  {
    double xp = x_in;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < len) { y = (x[k] - xp) + r * y; xp = x[k]; }
  }
  // the scan of rev_biquad_scan_t with a scalar transition: rows of 16 lanes by DPP shifts, row totals by readlane, the row's entry state
  // through r^(8 m) for lane m of the row
  const double y_carried = (seg == 0) ? dc[ch].y1 : 0.0;
  const double q1 = (r * r) * (r * r) * ((r * r) * (r * r)), q2 = q1 * q1, q4 = q2 * q2, q8 = q4 * q4, q16 = q8 * q8;
  y = y + q1 * dpp_zero_f64<0x111>(y);
  y = y + q2 * dpp_zero_f64<0x112>(y);
  y = y + q4 * dpp_zero_f64<0x114>(y);
  y = y + q8 * dpp_zero_f64<0x118>(y);
  const int row = lane >> 4;
  const double t[4] = {readlane_f64(y, 15), readlane_f64(y, 31), readlane_f64(y, 47), readlane_f64(y, 63)};
  double rs = 0.0;  // the state entering this lane's row
  auto walk = [&](double w) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (row == k) rs = w; w = t[k] + q16 * w; }
    if (half == 0 && lane == 0) xchg[ch] = w;
  };
  if (half == 0) walk(0.0);
  __syncthreads();  // also: every lane has read its inputs before pass 2 overwrites the buffer
  if (half == 1) walk(xchg[ch]);
  double ys = dpp_zero_f64<0x111>(y);  // start state of the segment = end state of the previous one
  {
    const int m = lane & 15;
    if (m & 1) rs = rs * q1;
    if (m & 2) rs = rs * q2;
    if (m & 4) rs = rs * q4;
    if (m & 8) rs = rs * q8;
    ys = ys + rs;
  }
  if (seg == 0) ys = y_carried;
  // pass 2 (the caller puts a barrier behind the call: the carried state is rewritten here)
  {
    double xp = x_in;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < len) { ys = (x[k] - xp) + r * ys; xp = x[k]; buf[REV_IDX(n0 + k, ch)] = ys; }
    if (len > 0 && n0 + len == T) { dc[ch].y1 = ys; dc[ch].x1 = xp; }
  }
}

// The TPT-SVF with coefficients that change every frame (LFO -> filter cutoff), over the chunk, both channels, in place:
//   s_{n+1} = A_n s_n + B_n x_n   — still linear in the state, so the blocked scan of rev_biquad_scan_t carries over with one change:
// a segment's transition is the product of its eight A_n (accumulated in pass 1) instead of a power of one matrix, and the
// Kogge-Stone scan combines (matrix, offset) pairs:  (Mc, zc) after (Mp, zp)  =  (Mc Mp,  zc + Mc zp).
// `coef(n, a1, a2, a3, m0, m1, m2)` recomputes frame n's coefficients (the same expressions as svf_apply / biquad_apply, so the same
// values as the serial loop's per-frame set); output = m0 v0 + m1 v1 + m2 v2 (for the SVF taps: (0,0,1), (0,1,0), (1,-k,-1)),
// rounded through f32 when the reference stores the filtered sample as f32 (FilterEffect).
template <bool ROUND_F32, typename CoefFn>
DEVO void svf_scan_time_varying(CoefFn coef, PgState2* st, double* buf, int T, double* xchg /* LDS [2][2] */) {
  const int tid = pg_tid();
  const int wave = tid >> 6, lane = tid & 63;
  const int ch = wave & 1, half = wave >> 1;
  const int seg = half * 64 + lane;
  const int n0 = seg * 8;
  const int len = n0 >= T ? 0 : (T - n0 < 8 ? T - n0 : 8);
  // pass 1: zero-state response (segment 0: from the carried state) and the segment's transition matrix
  double s1 = 0.0, s2 = 0.0;
  if (seg == 0) { s1 = st[ch].ic1eq; s2 = st[ch].ic2eq; }
  Mat2 M{1.0, 0.0, 0.0, 1.0};
  for (int k = 0; k < len; ++k) {
    double a1, a2, a3, m0, m1, m2;
    coef(n0 + k, a1, a2, a3, m0, m1, m2);
    const double v0 = buf[REV_IDX(n0 + k, ch)];
    const double v3 = v0 - s2;
    const double v1 = a1 * s1 + a2 * v3;
    const double v2 = s2 + a2 * s1 + a3 * v3;
    s1 = 2.0 * v1 - s1;
    s2 = 2.0 * v2 - s2;
    M = mat2_mul(Mat2{2.0 * a1 - 1.0, -2.0 * a2, 2.0 * a2, 1.0 - 2.0 * a3}, M);
  }
  const Mat2 Mseg = M;
  auto wave_scan = [&](Mat2& Ma, double& z1, double& z2) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double y1 = __shfl_up(z1, off, 64), y2 = __shfl_up(z2, off, 64);
      const Mat2 Y{__shfl_up(Ma.a, off, 64), __shfl_up(Ma.b, off, 64), __shfl_up(Ma.c, off, 64), __shfl_up(Ma.d, off, 64)};
      if (lane >= off) {
        z1 = z1 + (Ma.a * y1 + Ma.b * y2);
        const double z2n = z2 + (Ma.c * y1 + Ma.d * y2);
        z2 = z2n;
        Ma = mat2_mul(Ma, Y);
      }
    }
  };
  if (half == 0) {
    wave_scan(M, s1, s2);
    if (lane == 63) { xchg[ch * 2] = s1; xchg[ch * 2 + 1] = s2; }
  }
  __syncthreads();
  double e1 = s1, e2 = s2;
  if (half == 1) {
    if (lane == 0) { const double x1 = xchg[ch * 2], x2 = xchg[ch * 2 + 1]; const double t1 = s1 + (Mseg.a * x1 + Mseg.b * x2); s2 = s2 + (Mseg.c * x1 + Mseg.d * x2); s1 = t1; }
    wave_scan(M, s1, s2);
    e1 = s1; e2 = s2;
  }
  double b1 = __shfl_up(e1, 1, 64), b2 = __shfl_up(e2, 1, 64);
  if (lane == 0) {
    if (half == 0) { b1 = st[ch].ic1eq; b2 = st[ch].ic2eq; }
    else { b1 = xchg[ch * 2]; b2 = xchg[ch * 2 + 1]; }
  }
  // pass 2: every segment again from its true start state, writing the outputs (svf_tick, svf.rs:211-222)
  for (int k = 0; k < len; ++k) {
    double a1, a2, a3, m0, m1, m2;
    coef(n0 + k, a1, a2, a3, m0, m1, m2);
    const double v0 = buf[REV_IDX(n0 + k, ch)];
    const double v3 = v0 - b2;
    const double v1 = a1 * b1 + a2 * v3;
    const double v2 = b2 + a2 * b1 + a3 * v3;
    b1 = 2.0 * v1 - b1;
    b2 = 2.0 * v2 - b2;
    const double y = m0 * v0 + m1 * v1 + m2 * v2;
    buf[REV_IDX(n0 + k, ch)] = ROUND_F32 ? (double)(float)y : y;
  }
  __syncthreads();
  if (len > 0 && n0 + len == T) { st[ch].ic1eq = b1; st[ch].ic2eq = b2; }
}

DEVO bool delay_fast_eligible(const PgFx& fx) {
  const PgDelay& d = fx.u.delay;
  if (d.lfo.waveform >= 5) return false;  // Random / Smooth Random: the generator is drawn from on phase wraps — the exact serial lane renders those
  if (sm_need_ramp(d.delay_time) || sm_need_ramp(d.feedback) || sm_need_ramp(d.cutoff) || sm_need_ramp(d.drive) || sm_need_ramp(d.wet) ||
      sm_need_ramp(d.width) || sm_need_ramp(d.lfo_rate) || sm_need_ramp(d.d_time) || sm_need_ramp(d.d_feedback) || sm_need_ramp(d.d_filter))
    return false;
  // LFO -> time moves the tap by up to +-|depth| * 50 ms (delay.rs:349-352; |lfo| <= 1 up to the parabolic sine's overshoot, covered
  // by the margin): the chunk length follows the shortest delay the block can see
  const float dev_ms = fabsf(d.d_time.target) * 50.0f * 1.01f + 0.01f;
  const float min_samples = fmaxf(d.delay_time.target - (d.d_time.target != 0.0f ? dev_ms : 0.0f), 1.0f) * 0.001f * (float)fx.sample_rate;
  const float max_samples = fmaxf(d.delay_time.target + (d.d_time.target != 0.0f ? dev_ms : 0.0f), 1.0f) * 0.001f * (float)fx.sample_rate;
  return min_samples >= 66.0f && max_samples < (float)(d.mask - 8);
}

// ---- DelayEffect while parameters ramp (delay.rs:334-454 with any of its ten smoothers moving) ---------------------------------------------
// Every per-frame control value of the reference loop is the next() of a smoother (or derived from the LFO, whose rate may ramp): ten
// lanes of one wave lay out the nine value sequences and the LFO values of a piece — the same sm_next / lfo_run / set_rate calls as the
// serial loop, each sequence on its own: they do not interact — all lanes turn them into the per-frame delay, cutoff and feedback, and the
// piece is rendered like the steady state with per-frame values: taps, the time-varying SVF scan (svf_scan_time_varying), rational tanh
// with the frame's drive, DC scan, line writes and the dry / wet / width law with the frame's values. Chunks are cut for the shortest
// delay of the piece, so all line reads still hit pre-chunk data.
constexpr int DELAY_RAMP_SEQS = 10;  // delay_time, d_time, d_filter, cutoff, feedback, d_feedback, drive, wet, width, LFO value
DEVO PgSmooth& delay_ramp_smoother(PgDelay& d, int j) {
  switch (j) { case 0: return d.delay_time; case 1: return d.d_time; case 2: return d.d_filter; case 3: return d.cutoff; case 4: return d.feedback;
               case 5: return d.d_feedback; case 6: return d.drive; case 7: return d.wet; default: return d.width; }
}
// Shortest / longest delay (in samples) the coming frames can ask for: the smoothers move between their current value and their target (the
// critically damped spring of the delay time may travel on by |velocity| / omega when it was redirected in flight), the LFO stays in
// [-1, 1] up to the parabolic sine's overshoot.
DEVO bool delay_ramp_eligible(const PgFx& fx) {
  const PgDelay& d = fx.u.delay;
  if (d.lfo.waveform >= 5) return false;
  const float srf = (float)fx.sample_rate;
  const PgSmooth& t = d.delay_time;
  const float travel = t.kind == SM_SPRING && t.a > 0.0f ? fabsf(t.b) / (t.a * t.comp) : 0.0f;
  const float lo = fminf(t.current, t.target) - travel, hi = fmaxf(t.current, t.target) + travel;
  const float dev = fmaxf(fabsf(d.d_time.current), fabsf(d.d_time.target)) * 50.0f * 1.01f + 0.01f;
  const float min_samples = fmaxf(lo * 0.999f - dev, 1.0f) * 0.001f * srf, max_samples = fmaxf(hi * 1.001f + dev, 1.0f) * 0.001f * srf;
  return min_samples >= 66.0f && max_samples < (float)(d.mask - 8);
}
DEVO bool delay_ramp_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  if (!delay_ramp_eligible(fx)) return false;
  PgDelay& d = fx.u.delay;
  const int tid = pg_tid(), nt = blockDim.x;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  const int cap = fc.tmp_floats / DELAY_RAMP_SEQS < 1024 ? fc.tmp_floats / DELAY_RAMP_SEQS : 1024;
  if (cap < 8) return false;
  double* buf = (double*)fc.scratch;
  double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);
  int* red = (int*)(xchg + 4);  // [2] reductions
  float* seq = fc.tmp;          // [DELAY_RAMP_SEQS][cap]
  float* a_dly = seq, *a_dtime = seq + cap, *a_dflt = seq + 2 * cap, *a_cut = seq + 3 * cap, *a_fb = seq + 4 * cap, *a_dfb = seq + 5 * cap;
  float* a_drive = seq + 6 * cap, *a_wet = seq + 7 * cap, *a_width = seq + 8 * cap, *a_lfo = seq + 9 * cap;
  const float srf = (float)fx.sample_rate, nyq = srf / 2.0f;
  const double srd = (double)fx.sample_rate;
  const double kq = fmax(2.0 * (1.0 - (double)0.302f * 0.97), 0.03);
  const int svf_type = delay_to_svf(d.filter_type);
  const uint32_t mask = d.mask;
  const int mode = d.mode;
  for (int p0 = 0; p0 < frames; p0 += cap) {
    const int P = frames - p0 < cap ? frames - p0 : cap;
    __syncthreads();
    // 0. the ten sequences of the piece
    if (tid < DELAY_RAMP_SEQS - 1) {
      float* dst = seq + tid * cap;
      sm_sequence(delay_ramp_smoother(d, tid), dst, P);
    } else if (tid == 64) {  // lfo.run(), then the rate update while it ramps (delay.rs:343-347)
      PgLfo l = d.lfo;
      PgSmooth rate = d.lfo_rate;
      for (int k = 0; k < P; ++k) {
        a_lfo[k] = lfo_run(l);
        if (sm_need_ramp(rate)) lfo_set_rate(l, fx.sample_rate, (double)sm_next(rate));
      }
      d.lfo = l; d.lfo_rate = rate;
    }
    if (tid == 0) red[0] = 0x7fffffff;
    __syncthreads();
    // per-frame delay in samples, cutoff and feedback (delay.rs:349-372), in place
    for (int k = tid; k < P; k += nt) {
      const float lv = a_lfo[k];
      const float delay_ms = fmaxf(a_dly[k] + lv * a_dtime[k] * 50.0f, 1.0f);
      a_dly[k] = delay_ms * 0.001f * srf;
      a_cut[k] = clampf(a_cut[k] * powf(2.0f, lv * a_dflt[k] * 2.0f), 20.0f, nyq);
      const float bfb = a_fb[k];
      a_fb[k] = clampf(bfb + lv * a_dfb[k] * (1.0f - fabsf(bfb)), 0.0f, 0.999f);
    }
    __syncthreads();
    int done = 0;
    while (done < P) {
      // chunk length: every read of the chunk must hit pre-chunk data -> shorter than the shortest delay of the frames it holds
      int T = P - done;
      for (;;) {
        __syncthreads();
        if (tid == 0) red[0] = 0x7fffffff;
        __syncthreads();
        int m = 0x7fffffff;
        for (int k = tid; k < T; k += nt) { const int f = (int)floorf(a_dly[done + k]); m = m < f ? m : f; }
        if (m != 0x7fffffff) atomicMin(&red[0], m);
        __syncthreads();
        const int t_max = red[0] - 2;
        if (T <= t_max) break;
        T = t_max < 1 ? 1 : t_max;  // (eligibility keeps the delay >= 66 samples: t_max >= 64)
      }
      float* s0 = sig + 2 * (p0 + done);
      const uint32_t wp0[2] = {d.write_pos[0], d.write_pos[1]};
      const float fb_in[2] = {d.fb[0], d.fb[1]};
      __syncthreads();
      // 1. taps + interpolation (dsp/delay.rs:118-134)
      for (int s = tid; s < 2 * T; s += nt) {
        const int nn = s >> 1, ch = s & 1;
        const gdouble* line = (const gdouble*)d.line[ch];
        const uint32_t wp = (wp0[ch] + (uint32_t)nn) & mask;
        const double read_pos = (double)wp - (double)a_dly[done + nn];
        const double read_pos_floor = floor(read_pos);
        const double fraction = read_pos - read_pos_floor;
        const long long index1 = (long long)read_pos_floor;
        const uint32_t i1 = (uint32_t)((unsigned long long)index1 & (unsigned long long)mask);
        const uint32_t i2 = (uint32_t)((unsigned long long)(index1 + 1) & (unsigned long long)mask);
        if (fc.idx_log) fc.idx_log[(p0 + done + nn) * 2 + ch] = (int32_t)i1;
        const double v1 = line[i1], v2 = line[i2];
        buf[REV_IDX(nn, ch)] = (double)(float)(v1 + (v2 - v1) * fraction);
      }
      __syncthreads();
      // 2. wet path: SVF with the frame's cutoff -> saturate with the frame's drive -> DC filter
      {
        const float* cutp = a_cut + done;
        auto coef = [&](int n, double& a1, double& a2, double& a3, double& m0, double& m1, double& m2) {
          const double g = tan(F64_PI * (double)cutp[n] / srd);
          if (svf_type == 0) { m0 = 0.0; m1 = 0.0; m2 = 1.0; } else if (svf_type == 2) { m0 = 0.0; m1 = 1.0; m2 = 0.0; } else { m0 = 1.0; m1 = -kq; m2 = -1.0; }
          a1 = 1.0 / (1.0 + g * (g + kq));
          a2 = g * a1;
          a3 = g * a2;
        };
        svf_scan_time_varying<false>(coef, d.flt, buf, T, xchg);
        if (tid == 0) svf_set(d.coef, svf_type, fx.sample_rate, cutp[T - 1], 0.302f);  // the coefficient cache as the serial loop leaves it
      }
      __syncthreads();
      for (int s = tid; s < 2 * T; s += nt) { const int bi = REV_IDX(s >> 1, s & 1); buf[bi] = delay_saturate(buf[bi], a_drive[done + (s >> 1)]); }
      __syncthreads();
      dc_scan(d.dc, buf, T, xchg);
      __syncthreads();
      // 3. line writes and output with the frame's feedback, wet and width (delay.rs:374-452)
      for (int s = tid; s < 2 * T; s += nt) {
        const int nn = s >> 1, ch = s & 1;
        const float left_input = s0[2 * nn], right_input = s0[2 * nn + 1];
        const float clean_l = clampf((float)buf[REV_IDX(nn, 0)], -4.0f, 4.0f), clean_r = clampf((float)buf[REV_IDX(nn, 1)], -4.0f, 4.0f);
        float prev_l = fb_in[0], prev_r = fb_in[1];
        if (nn > 0) { prev_l = clampf((float)buf[REV_IDX(nn - 1, 0)], -4.0f, 4.0f); prev_r = clampf((float)buf[REV_IDX(nn - 1, 1)], -4.0f, 4.0f); }
        const float fb_n = a_fb[done + nn], wet = a_wet[done + nn], width = a_width[done + nn];
        float line_in;
        if (mode == 0) line_in = (ch == 0 ? left_input + prev_l * fb_n : right_input + prev_r * fb_n);
        else line_in = (ch == 0 ? (left_input + right_input) * 0.5f + prev_r * fb_n : prev_l * fb_n);
        ((gdouble*)d.line[ch])[(wp0[ch] + (uint32_t)nn) & mask] = (double)line_in;
        const float dry_gain = fminf((1.0f - wet) * 2.0f, 1.0f);
        const float wet_gain = fminf(wet * 2.0f, 1.0f);
        const float out_l = left_input * dry_gain + clean_l * wet_gain;
        const float out_r = right_input * dry_gain + clean_r * wet_gain;
        const float mid = (out_l + out_r) * 0.5f;
        const float side = (out_l - out_r) * 0.5f;
        const float res = ch == 0 ? mid + side * width : mid - side * width;
        __builtin_amdgcn_wave_barrier();
        s0[2 * nn + ch] = res;
      }
      __syncthreads();
      if (tid == 0) {
        d.fb[0] = clampf((float)buf[REV_IDX(T - 1, 0)], -4.0f, 4.0f);
        d.fb[1] = clampf((float)buf[REV_IDX(T - 1, 1)], -4.0f, 4.0f);
        d.write_pos[0] = (wp0[0] + (uint32_t)T) & mask;
        d.write_pos[1] = (wp0[1] + (uint32_t)T) & mask;
      }
      __syncthreads();
      done += T;
    }
  }
  __syncthreads();
  return true;
}

DEVO bool delay_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  if (!delay_fast_eligible(fx)) return false;
  PgDelay& d = fx.u.delay;
  const int tid = pg_tid(), nt = blockDim.x;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  double* buf = (double*)fc.scratch;                       // [T][2] f64, skewed (REV_IDX)
  double* xchg = (double*)(fc.scratch + REV_BUF_DOUBLES * 8);
  PgBiquadCoef* lco = (PgBiquadCoef*)(xchg + 4);           // the SVF as (a1, a2, a3, m0, m1, m2) for the blocked scan
  const float srf = (float)fx.sample_rate;
  // block constants: the reference's per-frame expressions with lfo_val * 0 folded (delay.rs:346-376)
  const float delay_ms = fmaxf(d.delay_time.target + 0.0f, 1.0f);
  const float delay_samples = delay_ms * 0.001f * srf;
  const float fb = clampf(d.feedback.target + 0.0f, 0.0f, 0.999f);
  const float drive = d.drive.target, wet = d.wet.target, width = d.width.target;
  // LFO -> time / feedback: the LFO's value differs per frame. Its f32 phase sequence is laid out per chunk by one lane (the plain
  // accumulation, exact by construction) in the first floats of the free `tmp` rows; everything that depends on it is element-wise.
  const float time_depth = d.d_time.target, fb_depth = d.d_feedback.target, filter_depth = d.d_filter.target;
  const bool lfo_mod = time_depth != 0.0f || fb_depth != 0.0f || filter_depth != 0.0f;
  float* ph = fc.tmp;
  __syncthreads();
  if (tid == 0) {
    const float cutoff = clampf(d.cutoff.target * 1.0f, 20.0f, (float)fx.sample_rate / 2.0f);
    svf_set(d.coef, delay_to_svf(d.filter_type), fx.sample_rate, cutoff, 0.302f);
    PgBiquadCoef c;
    c.a1 = d.coef.a1; c.a2 = d.coef.a2; c.a3 = d.coef.a3;
    if (d.coef.type == 0) { c.m0 = 0.0; c.m1 = 0.0; c.m2 = 1.0; }            // Lowpass: v2
    else if (d.coef.type == 2) { c.m0 = 0.0; c.m1 = 1.0; c.m2 = 0.0; }       // Bandpass: v1
    else { c.m0 = 1.0; c.m1 = -d.coef.k; c.m2 = -1.0; }                      // Highpass: v0 - k v1 - v2
    *lco = c;
  }
  __syncthreads();
  int t_max = (int)floorf(delay_samples) - 1;
  if (time_depth != 0.0f) t_max = (int)floorf(fmaxf(d.delay_time.target - (fabsf(time_depth) * 50.0f * 1.01f + 0.01f), 1.0f) * 0.001f * srf) - 2;
  if (lfo_mod && t_max > fc.tmp_floats) t_max = fc.tmp_floats;
  const uint32_t mask = d.mask;
  const float dry_gain = fminf((1.0f - wet) * 2.0f, 1.0f);
  const float wet_gain = fminf(wet * 2.0f, 1.0f);
  const int mode = d.mode;
  int done = 0;
  while (done < frames) {
    int T = frames - done;
    if (T > t_max) T = t_max;
    if (T > 1024) T = 1024;
    float* s0 = sig + 2 * done;
    const uint32_t wp0[2] = {d.write_pos[0], d.write_pos[1]};
    const float fb_in[2] = {d.fb[0], d.fb[1]};
    __syncthreads();
    if (lfo_mod) {
      if (tid == 0) {  // lfo.run() once per frame (delay.rs:343): the phase each frame reads, then the update
        float p = d.lfo.phase;
        const float inc = d.lfo.phase_inc;
        for (int k = 0; k < T; ++k) { ph[k] = p; p += inc; if (p >= 1.0f) p -= 1.0f; }
        d.lfo.phase = p;
      }
      __syncthreads();
    }
    // 1. taps + interpolation (delay.rs:118-134). Four items per lane and trip: their eight line reads are in flight together — one item per
    // trip made the block's eight trips eight consecutive round trips through the loaded memory system (the staged kernels' stage 1 is a latency chain)
    auto tap = [&](int s, uint32_t& i1, uint32_t& i2) -> double {   // read position of item s: line index of the two taps, returns the fraction
      const int nn = s >> 1, ch = s & 1;
      const uint32_t wp = (wp0[ch] + (uint32_t)nn) & mask;
      float delay_samples_n = delay_samples;
      if (time_depth != 0.0f) {  // delay.rs:349-352
        PgLfo l; l.phase = ph[nn]; l.phase_inc = 0.0f; l.waveform = d.lfo.waveform;
        const float time_mod_ms = lfo_value(l) * time_depth * 50.0f;
        delay_samples_n = fmaxf(d.delay_time.target + time_mod_ms, 1.0f) * 0.001f * srf;
      }
      const double read_pos = (double)wp - (double)delay_samples_n;
      const double read_pos_floor = floor(read_pos);
      const long long index1 = (long long)read_pos_floor;
      i1 = (uint32_t)((unsigned long long)index1 & (unsigned long long)mask);
      i2 = (uint32_t)((unsigned long long)(index1 + 1) & (unsigned long long)mask);
      return read_pos - read_pos_floor;
    };
    for (int sb = tid; sb < 2 * T; sb += 4 * nt) {
      double v1[4], v2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = sb + q * nt;
        v1[q] = 0.0; v2[q] = 0.0;
        if (s < 2 * T) {
          uint32_t i1, i2;
          (void)tap(s, i1, i2);
          const gdouble* line = (const gdouble*)d.line[s & 1];
          if (fc.idx_log) fc.idx_log[(done + (s >> 1)) * 2 + (s & 1)] = (int32_t)i1;  // test hook: read_idx1 of InterpolatedDelayLine::process (dsp/delay.rs:120-133)
          v1[q] = line[i1]; v2[q] = line[i2];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = sb + q * nt;
        if (s < 2 * T) {
          uint32_t i1, i2;
          const double fraction = tap(s, i1, i2);   // (recomputed: cheaper than four f64 fractions kept live under the loads)
          buf[REV_IDX(s >> 1, s & 1)] = (double)(float)(v1[q] + (v2[q] - v1[q]) * fraction);
        }
      }
    }
    __syncthreads();
    // 2. wet path
    if (filter_depth != 0.0f) {
      // LFO -> filter (delay.rs:354-363): frame n's cutoff = clamp(cutoff * 2^(lfo_n * depth * 2)), its SVF coefficients as svf_apply
      const float cutoff_base = d.cutoff.target, nyq = (float)fx.sample_rate / 2.0f;
      const double srd = (double)fx.sample_rate;
      const double kq = fmax(2.0 * (1.0 - (double)0.302f * 0.97), 0.03);
      const int wf = d.lfo.waveform;
      const int svf_type = delay_to_svf(d.filter_type);
      auto coef = [&](int n, double& a1, double& a2, double& a3, double& m0, double& m1, double& m2) {
        PgLfo l; l.phase = ph[n]; l.phase_inc = 0.0f; l.waveform = wf;
        const float filter_mod = powf(2.0f, lfo_value(l) * filter_depth * 2.0f);
        const float cutoff = clampf(cutoff_base * filter_mod, 20.0f, nyq);
        const double g = tan(F64_PI * (double)cutoff / srd);
        if (svf_type == 0) { m0 = 0.0; m1 = 0.0; m2 = 1.0; } else if (svf_type == 2) { m0 = 0.0; m1 = 1.0; m2 = 0.0; } else { m0 = 1.0; m1 = -kq; m2 = -1.0; }
        a1 = 1.0 / (1.0 + g * (g + kq));
        a2 = g * a1;
        a3 = g * a2;
      };
      svf_scan_time_varying<false>(coef, d.flt, buf, T, xchg);
      if (tid == 0) {  // the coefficient cache as the serial loop leaves it: set for the chunk's last frame
        PgLfo l; l.phase = ph[T - 1]; l.phase_inc = 0.0f; l.waveform = wf;
        const float cutoff = clampf(cutoff_base * powf(2.0f, lfo_value(l) * filter_depth * 2.0f), 20.0f, nyq);
        svf_set(d.coef, delay_to_svf(d.filter_type), fx.sample_rate, cutoff, 0.302f);
      }
    } else rev_biquad_scan(*lco, d.flt, buf, T, xchg);
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) { const int bi = REV_IDX(s >> 1, s & 1); buf[bi] = delay_saturate(buf[bi], drive); }
    __syncthreads();
    dc_scan(d.dc, buf, T, xchg);
    __syncthreads();
    // 3. line writes and output. clean(n) = clamp((f32) y_n); the line of frame n takes the previous frame's clean value
    for (int s = tid; s < 2 * T; s += nt) {
      const int nn = s >> 1, ch = s & 1;
      const float left_input = s0[2 * nn], right_input = s0[2 * nn + 1];
      const float clean_l = clampf((float)buf[REV_IDX(nn, 0)], -4.0f, 4.0f), clean_r = clampf((float)buf[REV_IDX(nn, 1)], -4.0f, 4.0f);
      float prev_l = fb_in[0], prev_r = fb_in[1];
      if (nn > 0) { prev_l = clampf((float)buf[REV_IDX(nn - 1, 0)], -4.0f, 4.0f); prev_r = clampf((float)buf[REV_IDX(nn - 1, 1)], -4.0f, 4.0f); }
      float fb_n = fb;
      if (fb_depth != 0.0f) {  // delay.rs:366-372
        PgLfo l; l.phase = ph[nn]; l.phase_inc = 0.0f; l.waveform = d.lfo.waveform;
        const float base_feedback = d.feedback.target;
        fb_n = clampf(base_feedback + lfo_value(l) * fb_depth * (1.0f - fabsf(base_feedback)), 0.0f, 0.999f);
      }
      float line_in;
      if (mode == 0) line_in = (ch == 0 ? left_input + prev_l * fb_n : right_input + prev_r * fb_n);       // stereo  :386-399
      else line_in = (ch == 0 ? (left_input + right_input) * 0.5f + prev_r * fb_n : prev_l * fb_n);         // ping-pong :400-418
      ((gdouble*)d.line[ch])[(wp0[ch] + (uint32_t)nn) & mask] = (double)line_in;
      // dry/wet law and M/S width (delay.rs:424-452); the lane of channel `ch` writes its own output sample
      const float out_l = left_input * dry_gain + clean_l * wet_gain;
      const float out_r = right_input * dry_gain + clean_r * wet_gain;
      const float mid = (out_l + out_r) * 0.5f;
      const float side = (out_l - out_r) * 0.5f;
      // both lanes of a frame read both inputs before either writes: the inputs were loaded above
      const float res = ch == 0 ? mid + side * width : mid - side * width;
      __builtin_amdgcn_wave_barrier();
      s0[2 * nn + ch] = res;
    }
    __syncthreads();
    if (tid == 0) {
      d.fb[0] = clampf((float)buf[REV_IDX(T - 1, 0)], -4.0f, 4.0f);
      d.fb[1] = clampf((float)buf[REV_IDX(T - 1, 1)], -4.0f, 4.0f);
      d.write_pos[0] = (wp0[0] + (uint32_t)T) & mask;
      d.write_pos[1] = (wp0[1] + (uint32_t)T) & mask;
    }
    __syncthreads();
    done += T;
  }
  if (tid == 0 && !lfo_mod) f32_phase_advance(d.lfo.phase, d.lfo.phase_inc, frames);  // lfo.run() once per frame (delay.rs:343)
  __syncthreads();
  return true;
}
