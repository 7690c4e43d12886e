// Time-parallel steady-state ReverbEffect (reference src/effect/reverb.rs:217-369, 409-447) for one workgroup.
//
// Why this is legal: in steady state (room size and wet not ramping) every feedback path of the reverb passes
// through a delay of at least `predelay = floor(29*size)` >= 725 frames (reverb.rs:196-213, size >= 25), and the
// vibrato read-ahead is <= 15 frames (reverb.rs:563). Within a chunk of T <= predelay frames all delay-line READS
// therefore hit samples written before the chunk, so the frames of a chunk are independent once reads are ordered
// before writes; the only true recurrences left are the three 2-state biquads (A, B, C) and the 16 vibrato phase
// accumulators.
//
//   phase 1   predelay ring: all reads (coalesced f64) -> barrier -> all writes
//   phase A   biquad A over the chunk (2 channel lanes)
//   phase 3   per 128-frame sub-chunk, one (frame, channel) item per lane:
//                reads : sin, 4 allpass taps, 8 vibrato line reads (2 taps each) of the PREVIOUS frame's `get`
//                barrier
//                writes: 4 allpass writes, 8 line `set`s (feedback handed over inside the lane)
//   phase B/C biquad B -> clamp, asin (all lanes) -> biquad C (2 channel lanes) -> dry mix (all lanes)
//
// Vibrato phases: the reference accumulates `phase += depth*speed` in f64 once per frame (reverb.rs:601-603). While
// the accumulator stays inside one binade, fl(p + d) = p + d_u with d_u = d rounded to a multiple of ulp(p), so
// p_n = p_0 + n*d_u EXACTLY (all terms are multiples of the ulp and < 2^53 ulps). Each chunk is cut so that no
// accumulator crosses a power of two inside it (a 1-frame chunk uses the plain hardware add), which makes the phase
// stream bit-identical to the serial recurrence — no index flips from phase drift.

struct RevRec {       // per (line, channel) vibrato phase record of the current chunk
  double p0, du;      // phase at chunk start; per-frame increment valid inside the chunk (or the raw increment when T == 1)
};

struct RevRing {      // generalized ring walk: position of frame i given the position at chunk start
  uint32_t p0, delay;
  __device__ __forceinline__ uint32_t at(uint32_t i) const {  // write/set position of frame i (i >= 0)
    if (p0 <= delay) { uint32_t v = p0 + i; uint32_t m = delay + 1; return v >= m ? v % m : v; }
    return i == 0 ? p0 : (i - 1) % (delay + 1);  // stale position above a shrunk ring: first write lands there, then 0,1,2..
  }
};

DEVO double rev_guard(float x, uint32_t fpd) {  // reverb.rs:231-236
  double v = (double)x;
  if (fabs(v) < 1.18e-23) v = (double)fpd * 1.18e-17;
  return v;
}

// one ReverbDelayLine::get for one channel at ring count `cnt` and vibrato phase `ph`  (reverb.rs:554-586)
DEVO double rev_get(const double* __restrict__ buf, uint32_t cnt, uint32_t delay, int ch, double ph, double blend) {
  double offset = (sin(ph) + 1.0) * 7.0;
  double working = (double)cnt + offset;
  double w_floor = floor(working);
  double w_frac = working - w_floor;
  uint32_t w_int = (uint32_t)w_floor;
  uint32_t read_1 = w_int;
  if (read_1 > delay) read_1 -= delay + 1;
  uint32_t read_2 = w_int + 1;
  if (read_2 > delay) read_2 -= delay + 1;
  double val1 = buf[read_1 * 2 + ch];
  double val2 = buf[read_2 * 2 + ch];
  double interpol = val1 * (1.0 - w_frac) + val2 * w_frac;
  return (1.0 - blend) * interpol + (val1 * blend);
}

// serial biquad over an LDS f64 buffer laid out [frame][2]; lanes 0 and 1 take one channel each
DEVO void rev_biquad_lanes(const PgBiquadCoef& c, PgState2* st, double* buf, int T) {
  if (threadIdx.x < 2) {
    const int ch = threadIdx.x;
    double a = st[ch].ic1eq, b = st[ch].ic2eq;
    const double a1 = c.a1, a2 = c.a2, a3 = c.a3, m0 = c.m0, m1 = c.m1, m2 = c.m2;
    for (int n = 0; n < T; ++n) {
      double v0 = buf[n * 2 + ch];
      double v3 = v0 - b;
      double v1 = a1 * a + a2 * v3;
      double v2 = b + a2 * a + a3 * v3;
      a = 2.0 * v1 - a;
      b = 2.0 * v2 - b;
      buf[n * 2 + ch] = m0 * v0 + m1 * v1 + m2 * v2;
    }
    st[ch].ic1eq = a; st[ch].ic2eq = b;
  }
}

DEVO bool reverb_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  PgReverb& r = fx.u.reverb;
  if (sm_need_ramp(r.room) || sm_need_ramp(r.wet)) return false;  // per-frame delay sizes / coefficients: exact serial path
  const int tid = threadIdx.x, nt = blockDim.x;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  double* bufA = (double*)fc.scratch;                       // [T_CAP][2] f64
  RevRec* rec = (RevRec*)(fc.scratch + 16384);              // [16]
  double* gl = (double*)(fc.scratch + 16384 + 16 * sizeof(RevRec));  // [16] epilogue gets
  int* ctl = fc.ctl;
  constexpr int T_CAP = 1024;

  // ---- block parameters (reverb.rs:429-440): delay lengths, blend/regen, the three low-pass coefficient sets ----
  __syncthreads();
  if (tid == 0) {
    ReverbBlock rb;
    reverb_params(fx, (double)r.room.target, (double)r.wet.target, rb);
    ((double*)gl)[0] = rb.blend; ((double*)gl)[1] = rb.regen; ((double*)gl)[2] = rb.wet;
    ctl[4] = (int)rb.predelay;
  }
  __syncthreads();
  const double blend = gl[0], regen = gl[1], wet = gl[2];
  const uint32_t predelay = (uint32_t)ctl[4];
  __syncthreads();
  // chunk length bound: all reads must hit pre-chunk data
  uint32_t t_max = predelay;
  for (int i = 0; i < 4; ++i) t_max = t_max < r.ap[i].delay ? t_max : r.ap[i].delay;
  for (int i = 0; i < 8; ++i) { uint32_t d = r.line[i].delay > 17 ? r.line[i].delay - 17 : 0; t_max = t_max < d ? t_max : d; }
  if (t_max < 32) return false;  // degenerate geometry: serial path
  if (t_max > (uint32_t)T_CAP) t_max = T_CAP;

  int done = 0;
  while (done < frames) {
    // ---- chunk set-up: vibrato phase records + chunk length (16 lanes) ----
    if (tid < 16) {
      const PgReverbLine& l = r.line[tid >> 1];
      const double p0 = l.vib_phase[tid & 1];
      const double d = l.depth * 0.1;
      unsigned long long bits = (unsigned long long)__double_as_longlong(p0);
      int e = (int)((bits >> 52) & 0x7ff);
      unsigned long long m_valid = 1;
      double du = d;
      if (e > 0 && e < 0x7ff && p0 > 0.0) {
        // u = ulp(p0) = 2^(e-1075+...); scale d by 1/u exactly
        const double inv_u = __longlong_as_double((long long)((unsigned long long)(1023 + 52 - (e - 1023)) << 52));  // 2^(52-(e-1023))
        const double u = __longlong_as_double((long long)((unsigned long long)(e - 52) << 52));                        // 2^((e-1023)-52)
        if (e - 52 > 0 && (1023 + 52 - (e - 1023)) > 0 && (1023 + 52 - (e - 1023)) < 0x7ff) {
          const double D = d * inv_u;  // exact (power-of-two scaling)
          if (D < 4503599627370496.0 && D >= 1.0) {
            const double Dr = rint(D);
            const double fr = D - floor(D);
            if (fr != 0.5) {
              const unsigned long long S = (bits & 0xFFFFFFFFFFFFFull) | 0x10000000000000ull;  // p0 / u
              const unsigned long long Di = (unsigned long long)Dr;
              const unsigned long long room = 0x20000000000000ull - 1ull - S;                  // 2^53 - 1 - S
              unsigned long long mv = Di > 0 ? room / Di : 1;
              if (mv >= 2) { m_valid = mv; du = Dr * u; }
            }
          }
        }
      }
      rec[tid].p0 = p0;
      rec[tid].du = du;
      ctl[8 + tid] = (int)(m_valid > 100000ull ? 100000ull : m_valid);
    }
    __syncthreads();
    int T = frames - done;
    if ((uint32_t)T > t_max) T = (int)t_max;
    for (int i = 0; i < 16; ++i) T = T < ctl[8 + i] ? T : ctl[8 + i];
    // when T == 1 `du` may be the raw increment of a lane whose closed form is invalid: p_1 = p0 + d is the plain add
    const bool single = (T == 1);
    float* s0 = sig + 2 * done;

    // ---- phase 1: predelay (DelayLine<2>::process, delay.rs:47-66) ----
    RevRing pr{r.pre_write_pos & r.pre_mask, predelay};
    for (int s = tid; s < 2 * T; s += nt) {
      int n = s >> 1, ch = s & 1;
      bufA[s] = r.pre[(size_t)pr.at(n + 1) * 2 + ch];
    }
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) {
      int n = s >> 1, ch = s & 1;
      r.pre[(size_t)pr.at(n) * 2 + ch] = rev_guard(s0[s], ch ? r.fpd_r : r.fpd_l);
    }
    // ---- phase A: biquad A ----
    rev_biquad_lanes(r.ca, r.sa, bufA, T);
    __syncthreads();

    // ---- phase 3: allpasses + vibrato lines, sub-chunks of nt/2 frames ----
    RevRing apr[4], lr[8];
    for (int i = 0; i < 4; ++i) apr[i] = RevRing{r.ap[i].write_pos, r.ap[i].delay};
    for (int i = 0; i < 8; ++i) lr[i] = RevRing{r.line[i].count, r.line[i].delay};
    const int src[8] = {3, 2, 1, 0, 0, 1, 2, 3};  // set(): a<-l, b<-k, c<-j, d<-i, e<-i, f<-j, g<-k, h<-l  (reverb.rs:275-282)
    for (int base = 0; base < T; base += nt / 2) {
      const int n = base + (tid >> 1), ch = tid & 1;
      const bool active = n < T;
      double x = 0.0, dl[4] = {0, 0, 0, 0}, F[8], o_prev = 0.0;
      if (active) {
        x = sin(bufA[n * 2 + ch] * wet);  // reverb.rs:253-257
        for (int i = 0; i < 4; ++i) dl[i] = r.ap[i].buf[(size_t)apr[i].at(n + 1) * 2 + ch];  // `delayed` (== new_delayed for delay >= 1)
        if (n >= 1) {
          // gets after the step of frame n-1: count = position of frame n, phase after n steps
          double g[8];
          for (int i = 0; i < 8; ++i) {
            const RevRec rc = rec[i * 2 + ch];
            const double ph = single ? rc.p0 + rc.du : rc.p0 + (double)n * rc.du;
            g[i] = rev_get(r.line[i].buf, lr[i].at(n), lr[i].delay, ch, ph, blend);
          }
          F[0] = (g[0] - (g[1] + g[2] + g[3])) * regen; F[1] = (g[1] - (g[0] + g[2] + g[3])) * regen;   // reverb.rs:303-319
          F[2] = (g[2] - (g[0] + g[1] + g[3])) * regen; F[3] = (g[3] - (g[0] + g[1] + g[2])) * regen;
          F[4] = (g[4] - (g[5] + g[6] + g[7])) * regen; F[5] = (g[5] - (g[4] + g[6] + g[7])) * regen;
          F[6] = (g[6] - (g[4] + g[5] + g[7])) * regen; F[7] = (g[7] - (g[4] + g[5] + g[6])) * regen;
          o_prev = (g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7]) / 8.0;                        // reverb.rs:321-338
        } else {
          for (int i = 0; i < 8; ++i) F[i] = r.line[i].feedback[ch];  // handed over from the previous chunk
        }
      }
      __syncthreads();  // every read of this sub-chunk has been issued and consumed
      if (active) {
        if (n >= 1) bufA[(n - 1) * 2 + ch] = o_prev;
        double apo[4];
        double v = x;
        for (int i = 0; i < 4; ++i) {  // AllpassDelayLine::process (delay.rs:314-350)
          double b = v - (dl[i] * 0.5);
          r.ap[i].buf[(size_t)apr[i].at(n) * 2 + ch] = b;
          v = b * 0.5 + dl[i];
          apo[i] = v;
        }
        for (int i = 0; i < 8; ++i) r.line[i].buf[(size_t)lr[i].at(n) * 2 + ch] = apo[src[i]] + F[i];  // set (reverb.rs:588-594)
      }
      __syncthreads();
    }
    // ---- epilogue: gets after the step of the chunk's last frame (16 lanes: one (line, channel) each) ----
    if (tid < 16) {
      const int i = tid >> 1, ch = tid & 1;
      const RevRec rc = rec[tid];
      const double ph = single ? rc.p0 + rc.du : rc.p0 + (double)T * rc.du;
      gl[tid] = rev_get(r.line[i].buf, lr[i].at(T), lr[i].delay, ch, ph, blend);
      r.line[i].vib_phase[ch] = ph;
    }
    __syncthreads();
    if (tid < 2) {
      const int ch = tid;
      double g[8];
      for (int i = 0; i < 8; ++i) g[i] = gl[i * 2 + ch];
      r.line[0].feedback[ch] = (g[0] - (g[1] + g[2] + g[3])) * regen; r.line[1].feedback[ch] = (g[1] - (g[0] + g[2] + g[3])) * regen;
      r.line[2].feedback[ch] = (g[2] - (g[0] + g[1] + g[3])) * regen; r.line[3].feedback[ch] = (g[3] - (g[0] + g[1] + g[2])) * regen;
      r.line[4].feedback[ch] = (g[4] - (g[5] + g[6] + g[7])) * regen; r.line[5].feedback[ch] = (g[5] - (g[4] + g[6] + g[7])) * regen;
      r.line[6].feedback[ch] = (g[6] - (g[4] + g[5] + g[7])) * regen; r.line[7].feedback[ch] = (g[7] - (g[4] + g[5] + g[6])) * regen;
      bufA[(T - 1) * 2 + ch] = (g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7]) / 8.0;
    }
    if (tid == 0) {  // advance ring positions (uniform integer bookkeeping)
      r.pre_write_pos = pr.at(T);
      for (int i = 0; i < 4; ++i) r.ap[i].write_pos = apr[i].at(T);
      for (int i = 0; i < 8; ++i) r.line[i].count = lr[i].at(T);
    }
    __syncthreads();
    // ---- biquad B -> clamp -> asin -> biquad C -> dry mix (reverb.rs:340-368) ----
    rev_biquad_lanes(r.cb, r.sb, bufA, T);
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) bufA[s] = asin(clampd(bufA[s], -1.0, 1.0));
    __syncthreads();
    rev_biquad_lanes(r.cc, r.sc, bufA, T);
    __syncthreads();
    for (int s = tid; s < 2 * T; s += nt) {
      double y = bufA[s];
      if (wet != 1.0) y += rev_guard(s0[s], (s & 1) ? r.fpd_r : r.fpd_l) * (1.0 - wet);
      s0[s] = (float)y;
    }
    __syncthreads();
    done += T;
  }
  return true;
}
