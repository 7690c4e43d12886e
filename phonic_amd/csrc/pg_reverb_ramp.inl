// ReverbEffect while its `wet` smoother moves and the room size stands still (reference src/effect/reverb.rs:409-427: the ramp branch calls
// update_parameters per frame). What a wet ramp changes per frame: the gain in front of the `sin` of the allpass input, the dry share of the
// output, and the cutoff of the three low-pass biquads (10000 - room * wet * 3000 Hz); the delay lengths, blend and regen follow the room size
// alone. So the time-parallel reverb carries over with two changes: one lane lays out the smoother's f32 value sequence of the piece with the
// serial loop's own sm_next calls (and the clamped f32 cutoff next to it, in the block's temporary row), and the three biquads run as the
// time-varying blocked scan (svf_scan_time_varying, pg_delay_fast.inl) with every frame's coefficient set recomputed as biquad_set does.
// A moving room size changes the ring lengths per frame (a linear ramp of at most 109 frames): that block stays on the exact serial lane.
// The three low-pass biquads of a piece whose `wet` moves share their cutoff sequence and differ in Q: g = tan(pi * cutoff / sample_rate) is a
// frame's most expensive term by far (an f64 tangent) and the same for all three. The lane that scans frames n0 .. n0 + 7 of a channel keeps
// their eight g in registers — laid out once for the front's scan, once for the two scans of the tail — and a scan derives a1 / a2 / a3 from g
// and its own k = 1 / Q with biquad_apply's expressions (biquad.rs:153-271; m = (0, 0, 1): low-pass). The scan itself is
// svf_scan_time_varying's (pg_delay_fast.inl), unrolled over a segment's eight frames so that g[k] is a register. Through the generic scan
// every frame's tangent was evaluated twelve times (two passes, two channels, three filters): 75 K of a commanded reverb's 330 K cycles.
DEVO void rev_ramp_g(const float* cut, uint32_t sr, int T, double (&g)[8]) {
  const int tid = pg_tid();
  const int seg = ((tid >> 6) >> 1) * 64 + (tid & 63);
  const int n0 = seg * 8;
  const int len = n0 >= T ? 0 : (T - n0 < 8 ? T - n0 : 8);
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = 0.0;
#pragma nounroll
  for (int k = 0; k < len; ++k) {   // (rolled: one copy of the tangent; the selects keep g in registers)
    const double v = tan(F64_PI * (double)cut[n0 + k] / (double)sr);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = (j == k) ? v : g[j];
  }
}
DEVO void svf_scan_lowpass_g(const double (&g)[8], float q, PgState2* st, double* buf, int T, double* xchg /* LDS [2][2] */) {
  const int tid = pg_tid();
  const int wave = tid >> 6, lane = tid & 63;
  const int ch = wave & 1, half = wave >> 1;
  const int seg = half * 64 + lane;
  const int n0 = seg * 8;
  const int len = n0 >= T ? 0 : (T - n0 < 8 ? T - n0 : 8);
  const double kq = 1.0 / (double)q;
  const double m0 = 0.0, m1 = 0.0, m2 = 1.0;
  // pass 1: zero-state response (segment 0: from the carried state) and the segment's transition matrix
  double s1 = 0.0, s2 = 0.0;
  if (seg == 0) { s1 = st[ch].ic1eq; s2 = st[ch].ic2eq; }
  Mat2 M{1.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int k = 0; k < 8; ++k) if (k < len) {
    const double a1 = 1.0 / (1.0 + g[k] * (g[k] + kq)), a2 = g[k] * a1, a3 = g[k] * a2;
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
#pragma unroll
  for (int k = 0; k < 8; ++k) if (k < len) {
    const double a1 = 1.0 / (1.0 + g[k] * (g[k] + kq)), a2 = g[k] * a1, a3 = g[k] * a2;
    const double v0 = buf[REV_IDX(n0 + k, ch)];
    const double v3 = v0 - b2;
    const double v1 = a1 * b1 + a2 * v3;
    const double v2 = b2 + a2 * b1 + a3 * v3;
    b1 = 2.0 * v1 - b1;
    b2 = 2.0 * v2 - b2;
    buf[REV_IDX(n0 + k, ch)] = m0 * v0 + m1 * v1 + m2 * v2;
  }
  __syncthreads();
  if (len > 0 && n0 + len == T) { st[ch].ic1eq = b1; st[ch].ic2eq = b2; }
}
DEVO bool reverb_wet_ramp_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  PgReverb& r = fx.u.reverb;
  if (!reverb_wet_ramp_eligible(fx)) return false;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  if (fc.tmp_floats < 2 * frames && fc.tmp_floats < 2 * REV_T_CAP) return false;  // two f32 per frame of a piece
  const int tid = pg_tid(), nt = blockDim.x;
  const RevLds m = rev_lds(fc.scratch);
  rev_load_vtab(r, m);
  RevBlock b;
  if (!rev_block_params(fx, m, fc.ctl, b)) return false;  // ring lengths, blend, regen, predelay: functions of the room size
  const double rs = (double)r.room.target;                // sm_next() of a resting smoother
  const uint32_t sr = fx.sample_rate;
  const float nyq = (float)sr / 2.0f;
  for (int done = 0; done < frames; done += REV_T_CAP) {
    const int T = frames - done < REV_T_CAP ? frames - done : REV_T_CAP;
    float* s0 = sig + 2 * done;
    float* wv = fc.tmp;       // [T] wet per frame
    float* cut = fc.tmp + T;  // [T] low-pass cutoff per frame (reverb.rs:413, clamped as update_filter_coefs does)
    __syncthreads();
    // the smoother's value sequence on one lane (sm_sequence: the exponential smoother as a tight register loop — the generic sm_next loop
    // through a local PgSmooth cost ~500 cycles per frame here, 0.2 ms per 1024-frame block of ONE commanded unit: tools/diag_cmd.py), the
    // cutoff that follows from it on all lanes
    if (tid == 0) sm_sequence(r.wet, wv, T);
    __syncthreads();
    for (int k = tid; k < T; k += nt) cut[k] = clampf((float)(10000.0 - (rs * (double)wv[k] * 3000.0)), 20.0f, nyq);
    __syncthreads();
    // front: predelay, then biquad A with moving coefficients
    rev_front<false>(r, s0, T, m, b, fc.diag);
    {
      double g[8];
      rev_ramp_g(cut, sr, T, g);
      svf_scan_lowpass_g(g, 1.618034f, r.sa, m.bufA, T, m.xchg);
    }
    __syncthreads();
    // mid: allpasses and vibrato lines, the wet gain per frame
    rev_mid(r, T, m, b, fc.ctl, fc.diag, fc.idx_log ? fc.idx_log + (size_t)done * 16 : nullptr, wv);
    // tail: biquad B -> clamp / asin -> biquad C -> dry mix (reverb.rs:340-368), wet per frame
    double* bufA = m.bufA;
    double g[8];
    rev_ramp_g(cut, sr, T, g);
    svf_scan_lowpass_g(g, 0.618034f, r.sb, bufA, T, m.xchg);
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) { const int bi = REV_IDX(s >> 1, s & 1); bufA[bi] = rev_asin(clampd(bufA[bi], -1.0, 1.0)); }
    __syncthreads();
    svf_scan_lowpass_g(g, 0.5f, r.sc, bufA, T, m.xchg);
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) {
      double y = bufA[REV_IDX(s >> 1, s & 1)];
      const double w = (double)wv[s >> 1];
      if (w != 1.0) y += rev_guard(s0[s], (s & 1) ? r.fpd_r : r.fpd_l) * (1.0 - w);
      s0[s] = (float)y;
    }
    __syncthreads();
    if (tid == 0) {  // the coefficient sets the serial loop ends the piece with
      if (biquad_set(r.ca, 0, sr, cut[T - 1], 1.618034f, 0.0f))
        if (biquad_set(r.cb, 0, sr, cut[T - 1], 0.618034f, 0.0f)) (void)biquad_set(r.cc, 0, sr, cut[T - 1], 0.5f, 0.0f);
    }
  }
  __syncthreads();
  if (tid == 0) r.cache_valid = 0;  // the block parameters cached above belong to the target values
  __syncthreads();
  return true;
}
