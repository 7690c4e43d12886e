// Time-parallel CompressorEffect / limiter (reference src/effect/compressor.rs:230-294) for one workgroup (the makeup gain may ramp: its
// smoother's per-frame values are laid out by the lane that walks the envelope).
//
// The serial loop pays one dependent HBM read per frame (the look-ahead line): 1.2 ms per 1024-frame block on one lane. Here
//   * the delayed frame is  line[wp + n - delay]  for n < delay and the block's own input for n >= delay: read in parallel;
//   * the look-ahead peak (LookupDelayLine, delay.rs:206-265) is by construction the maximum of the frame peaks over the last
//     `delay_frames` frames (a new peak >= the tracked one replaces it; when the tracked one leaves the window the window is
//     rescanned), i.e. a sliding-window maximum: log-step doubling over [history | block] in LDS, exact (max is order-free);
//   * dB conversion, knee / slope law, makeup and the multiplication are element-wise;
//   * only the envelope follower (attack / release switch on the sign of input - envelope, envelope.rs:51-60) stays a serial
//     recurrence: one lane walks the block in LDS, no memory access.
// The tracked peak's position is left on the NEWEST frame holding the maximum; any frame holding it keeps the reference's expiry
// logic exact (the value sequence is the window maximum either way).

DEVO float comp_gain_reduction_db(const PgComp& c, float envelope) {  // compressor.rs:262-279
  const float t = c.threshold, w = c.knee;
  const float slope = (c.ratio >= 20.0f) ? 1.0f : 1.0f - 1.0f / c.ratio;
  if (w > 0.0f && envelope > (t - w / 2.0f) && envelope < (t + w / 2.0f)) {
    const float knee_lower = t - w / 2.0f;
    const float x = (envelope - knee_lower) / w;
    return x * x * slope * w / 2.0f;
  }
  if (envelope > (t + w / 2.0f)) return (envelope - t) * slope;
  return 0.0f;
}

constexpr int COMP_FAST_CAP = 4096;  // history + block frames held in LDS

DEVO bool comp_fast_eligible(const PgFx& fx) {
  const PgComp& c = fx.u.comp;
  return c.delay_frames >= 1 && c.delay_frames <= c.mask && (int)c.delay_frames - 1 + 1024 <= COMP_FAST_CAP;
}

DEVO bool comp_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  if (!comp_fast_eligible(fx)) return false;
  PgComp& c = fx.u.comp;
  const int tid = pg_tid(), nt = blockDim.x;
  float* a0 = (float*)fc.scratch;              // [COMP_FAST_CAP] frame peaks: history (W - 1 frames) then the block
  float* a1 = a0 + COMP_FAST_CAP;              // ping-pong partner of the doubling; later the copy of the block's input
  float* env = a1 + COMP_FAST_CAP;             // [1024] input dB, then envelope
  int* red = (int*)(env + 1024);               // [8] reductions
  static_assert((2 * COMP_FAST_CAP + 1024 + 8) * 4 <= FAST_SCRATCH_COMP_BYTES, "compressor fast path: LDS arena too small");
  const int W = (int)c.delay_frames, H = W - 1;
  const uint32_t mask = c.mask;
  gdouble* line = (gdouble*)c.line;
  const bool limiter = c.ratio >= 20.0f;
  const float makeup = c.makeup.target;
  const bool makeup_ramps = sm_need_ramp(c.makeup);  // decided once per process call, like `need_ramp` in the serial loop's next_value()
  const int total = n_samples / 2;
  for (int f0 = 0; f0 < total; f0 += 1024) {   // pieces of <= 1024 frames
    const int N = total - f0 < 1024 ? total - f0 : 1024, TOT = H + N;
    float* sp = sig + 2 * f0;
    const uint32_t wp0 = c.write_pos;
    __syncthreads();
    PG_STAMP(fc.diag, 24);
    // 1. frame peaks (delay.rs:245-247): history from the line, the block from the signal
    for (int i = tid; i < TOT; i += nt) {
      float p;
      if (i < H) {
        const uint32_t fi = (wp0 + (uint32_t)(i - H)) & mask;
        p = (float)fmax(fmax(0.0, fabs(line[fi * 2])), fabs(line[fi * 2 + 1]));   // the stored samples are f32 values: exact
      } else {
        const int n = i - H;
        p = fmaxf(fmaxf(0.0f, fabsf(sp[2 * n])), fabsf(sp[2 * n + 1]));
      }
      a0[i] = p;
    }
    if (tid == 0) { red[0] = 0; red[1] = -0x7fffffff; }
    __syncthreads();
    PG_STAMP(fc.diag, 25);
    // 2. tracked peak after the piece: maximum of the last window and its newest holder (peaks are >= 0: their bits order like ints).
    // (Per-lane maxima, a wave reduction, ONE LDS atomic per wave: 960 atomics on one LDS word serialise — 20 K cycles of a lone workgroup.)
    {
      int mx = 0;
      for (int i = TOT - W + tid; i < TOT; i += nt) if (i >= 0) mx = max(mx, (int)__float_as_uint(a0[i]));
      for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
      if ((tid & 63) == 0) atomicMax(&red[0], mx);
      __syncthreads();
      const float m = __uint_as_float((uint32_t)red[0]);
      int at = -0x7fffffff;
      for (int i = TOT - W + tid; i < TOT; i += nt) if (i >= 0 && a0[i] == m) at = max(at, i - H);
      for (int off = 32; off > 0; off >>= 1) at = max(at, __shfl_xor(at, off, 64));
      if ((tid & 63) == 0) atomicMax(&red[1], at);
    }
    PG_STAMP(fc.diag, 26);
    // 3. input level in dB (compressor.rs:241-255)
    if (limiter) {
      float* src = a0; float* dst = a1;
      int span = 1;
      while (span * 2 <= W) {  // after a pass: src[i] = max over the `span` entries ending at i. Up to three doubling levels per pass (eight taps
        const int lv = span * 8 <= W ? 3 : (span * 4 <= W ? 2 : 1);   // `span` apart): a look-ahead of 960 frames takes 3 passes and barriers, not 9
        const int taps = 1 << lv;
        __syncthreads();
        for (int i = tid; i < TOT; i += nt) {   // (an index below 0 is clamped to entry 0, which such a window holds anyway; a tap beyond `taps` re-reads entry i)
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int k = i - j * span; v[j] = src[j < taps ? (k > 0 ? k : 0) : i]; }
          dst[i] = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
        }
        float* t = src; src = dst; dst = t;
        span <<= lv;
      }
      __syncthreads();
      for (int n = tid; n < N; n += nt) {   // window [i - W + 1, i] = [i - span + 1, i] U [i - W + 1, i - W + span]
        const int i = n + H;
        const float m = fmaxf(src[i], src[i - W + span]);
        env[n] = (m > 1e-6f) ? 20.0f * pg_log10f(m) : -120.0f;
      }
    } else {
      __syncthreads();
      for (int n = tid; n < N; n += nt) { const float fp = a0[n + H]; env[n] = (fp > 1e-6f) ? 20.0f * pg_log10f(fp) : -120.0f; }
    }
    __syncthreads();
    PG_STAMP(fc.diag, 27);
    // 4. envelope follower: the one serial recurrence, walked by wave 0 as a systolic chain, 64 frames per pass. Lane j holds the level of
    // frame j; every step each lane takes the envelope of the lane below it (a DPP wavefront shift: lane 0 takes the envelope carried in
    // from the previous pass) and applies EnvelopeFollower::run (envelope.rs:51-60) with its own level — the same f32 operations in the same
    // order: compare, subtract, multiply, add. Lane 0 is right from the first step on, lane 1 from the second ... and a lane that is right
    // stays right (its input no longer changes): after 64 steps every lane holds the envelope of its frame. Five dependent VALU
    // instructions per frame and no memory access inside the pass (a lone lane walking the block in LDS paid a round trip per frame:
    // 133 K of the 200 K cycles a bus limiter spent on a block). Meanwhile the other waves stage the input (the output pass overwrites it).
    if (tid < 64) {
      float carry = c.env_current;
      const float att = c.env_attack, rel = c.env_release;
      for (int base = 0; base < N; base += 64) {
        const int cnt = N - base < 64 ? N - base : 64;
        const float x = tid < cnt ? env[base + tid] : 0.0f;
        float cur = carry;
#pragma unroll 8
        for (int t = 0; t < 64; ++t) {
          const float below = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(carry), (int)__float_as_uint(cur), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
          const float coef = x > below ? att : rel;
          cur = x + coef * (below - x);
        }
        if (tid < cnt) env[base + tid] = cur;
        carry = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cur), cnt - 1));
      }
      const float cur = carry;
      if (tid == 0) {
        c.env_current = cur;
        if (makeup_ramps) sm_sequence(c.makeup, a0, N);  // makeup_gain.next_value() per frame (a0 is free by now)
        c.peak_value = (double)__uint_as_float((uint32_t)red[0]);
        c.peak_pos = (wp0 + (uint32_t)red[1]) & mask;
        c.write_pos = (wp0 + (uint32_t)N) & mask;
      }
      if (nt <= 64) for (int i = tid; i < 2 * N; i += nt) a1[i] = sp[i];
    } else {
      for (int i = tid - 64; i < 2 * N; i += nt - 64) a1[i] = sp[i];
    }
    __syncthreads();
    PG_STAMP(fc.diag, 28);
    // 5. gain and output (compressor.rs:257-292); 6. the block's input goes into the line
    for (int s = tid; s < 2 * N; s += nt) {
      const int n = s >> 1, ch = s & 1;
      const float total_gain = db_to_linear((makeup_ramps ? a0[n] : makeup) - comp_gain_reduction_db(c, env[n]));
      const float delayed = n < W ? (float)line[((wp0 + (uint32_t)(n - W)) & mask) * 2 + ch] : a1[2 * (n - W) + ch];
      sp[s] = delayed * total_gain;
    }
    __syncthreads();  // every read of the line's history is done before its slots are rewritten
    PG_STAMP(fc.diag, 29);
    // (the ring is only next_pow2(delay) frames long: of the frames that share a slot, the last one stays — write only those)
    for (int s = tid; s < 2 * N; s += nt) if ((s >> 1) >= N - (int)(mask + 1)) line[((wp0 + (uint32_t)(s >> 1)) & mask) * 2 + (s & 1)] = (double)a1[s];
    __syncthreads();
    PG_STAMP(fc.diag, 30);
  }
  return true;
}
