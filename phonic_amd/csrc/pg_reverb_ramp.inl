// ReverbEffect while its `wet` smoother moves and the room size stands still (reference src/effect/reverb.rs:409-427: the ramp branch calls
// update_parameters per frame). What a wet ramp changes per frame: the gain in front of the `sin` of the allpass input, the dry share of the
// output, and the cutoff of the three low-pass biquads (10000 - room * wet * 3000 Hz); the delay lengths, blend and regen follow the room size
// alone. So the time-parallel reverb carries over with two changes: one lane lays out the smoother's f32 value sequence of the piece with the
// serial loop's own sm_next calls (and the clamped f32 cutoff next to it, in the block's temporary row), and the three biquads run as the
// time-varying blocked scan (svf_scan_time_varying, pg_delay_fast.inl) with every frame's coefficient set recomputed as biquad_set does.
// A moving room size changes the ring lengths per frame (a linear ramp of at most 109 frames): that block stays on the exact serial lane.
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
    auto coef_q = [&](float q) {
      return [=](int n, double& a1, double& a2, double& a3, double& m0, double& m1, double& m2) {
        PgBiquadCoef c;
        c.type = 0; c.sample_rate = sr; c.cutoff = cut[n]; c.q = q; c.gain = 0.0f;
        c.a1 = 0.0; c.a2 = 0.0; c.a3 = 0.0; c.m0 = 0.0; c.m1 = 0.0; c.m2 = 0.0;
        (void)biquad_apply(c);
        a1 = c.a1; a2 = c.a2; a3 = c.a3; m0 = c.m0; m1 = c.m1; m2 = c.m2;
      };
    };
    // front: predelay, then biquad A with moving coefficients
    rev_front<false>(r, s0, T, m, b, fc.diag);
    svf_scan_time_varying<false>(coef_q(1.618034f), r.sa, m.bufA, T, m.xchg);
    __syncthreads();
    // mid: allpasses and vibrato lines, the wet gain per frame
    rev_mid(r, T, m, b, fc.ctl, fc.diag, fc.idx_log ? fc.idx_log + (size_t)done * 16 : nullptr, wv);
    // tail: biquad B -> clamp / asin -> biquad C -> dry mix (reverb.rs:340-368), wet per frame
    double* bufA = m.bufA;
    svf_scan_time_varying<false>(coef_q(0.618034f), r.sb, bufA, T, m.xchg);
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) { const int bi = REV_IDX(s >> 1, s & 1); bufA[bi] = rev_asin(clampd(bufA[bi], -1.0, 1.0)); }
    __syncthreads();
    svf_scan_time_varying<false>(coef_q(0.5f), r.sc, bufA, T, m.xchg);
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
