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

struct RevRec {       // per (line, channel) vibrato phase record of the current chunk: two closed-form pieces
  double p0, du0;     // phase after n steps = p0 + n*du0            for n <= m0
  double p1, du1;     //                     = p1 + (n-m0-1)*du1     for n >  m0   (p1 = fl(p_m0 + d): the binade-crossing step)
  uint32_t m0, m1;    // steps each piece is valid for; the chunk is cut at m0 + 1 + m1
};

// Closed form of the f64 accumulator p += d around p: while p stays inside its binade, fl(p + d) = p + du exactly, where
// du = d rounded to a multiple of ulp(p). Returns the number of steps for which that is provably true (0: not at all).
DEVO uint32_t rev_phase_piece(double p, double d, double& du) {
  du = d;
  unsigned long long bits = (unsigned long long)__double_as_longlong(p);
  int e = (int)((bits >> 52) & 0x7ff);
  if (!(e > 52 && e < 0x7ff && p > 0.0)) return 0;
  const int ie = 1023 + 52 - (e - 1023);
  if (!(ie > 0 && ie < 0x7ff)) return 0;
  const double inv_u = __longlong_as_double((long long)((unsigned long long)ie << 52));        // 1 / ulp(p)
  const double u = __longlong_as_double((long long)((unsigned long long)(e - 52) << 52));      // ulp(p)
  const double D = d * inv_u;  // exact (power-of-two scaling)
  if (!(D < 4503599627370496.0 && D >= 1.0)) return 0;
  const double Dr = rint(D);
  if (D - floor(D) == 0.5) return 0;  // a tie would round to even: depends on the accumulator's parity
  const unsigned long long S = (bits & 0xFFFFFFFFFFFFFull) | 0x10000000000000ull;  // p / ulp(p)
  const double room = (double)(0x20000000000000ull - 1ull - S);                     // ulps left in the binade (exact)
  // exactly floor(room / Dr) steps stay inside the binade. One f64 division plus an exact fma remainder correction
  // (a 64-bit integer division is a ~150k-cycle software loop on this target).
  double q = floor(room / Dr);
  const double rem = fma(-q, Dr, room);  // exact: |room - q*Dr| < 2^53
  if (rem < 0.0) q -= 1.0;
  else if (rem >= Dr) q += 1.0;
  if (!(q >= 1.0)) return 0;
  du = Dr * u;
  return q > 1048576.0 ? 1048576u : (uint32_t)q;
}
DEVO double rev_phase_at(const RevRec& rc, uint32_t n) {
  return n <= rc.m0 ? rc.p0 + (double)n * rc.du0 : rc.p1 + (double)(n - rc.m0 - 1) * rc.du1;
}

typedef __attribute__((address_space(1))) double gdouble;  // global (HBM) address space: global_load/store, not flat

struct RevDesc {      // one delay line for the current chunk: base pointer, ring position at chunk start, ring length - 1
  double* buf;
  uint32_t p0, delay;
};
// generalized ring walk: position of frame i (i <= T + 1 <= delay + 1, so one conditional subtract replaces the modulo)
DEVO uint32_t rev_at(const RevDesc& d, uint32_t i) {
  if (d.p0 <= d.delay) { uint32_t v = d.p0 + i; uint32_t m = d.delay + 1; return v >= m ? v - m : v; }
  return i == 0 ? d.p0 : i - 1;  // stale position above a shrunk ring: the first write lands there, then 0, 1, 2 ...
}

// values every lane of the workgroup agrees on, moved to scalar registers (loop control and ring arithmetic go to the SALU)
DEVO uint32_t uni_u32(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
DEVO double uni_f64(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Workgroup-uniform ring descriptor held in scalar registers: ring arithmetic on it costs no LDS round trip.
//   position of frame i = (pe + i) mod m, except the first write above a shrunk ring (`stale`), which lands on the raw p0.
struct RevRing {
  double* buf;
  uint32_t p0, m, pe, stale;
};
DEVO RevRing rev_ring_uniform(const RevDesc& d) {
  const unsigned long long b = (unsigned long long)d.buf;
  const uint32_t lo = uni_u32((uint32_t)b), hi = uni_u32((uint32_t)(b >> 32));
  RevRing u;
  u.buf = (double*)(((unsigned long long)hi << 32) | lo);
  u.p0 = uni_u32(d.p0);
  const uint32_t delay = uni_u32(d.delay);
  u.m = delay + 1;
  u.stale = u.p0 > delay ? 1u : 0u;
  u.pe = u.stale ? delay : u.p0;
  return u;
}
DEVO uint32_t ring_wrap(uint32_t v, uint32_t m) { const uint32_t w = v - m; return v < w ? v : w; }  // v < 2m: one conditional subtract
DEVO uint32_t ring_at(const RevRing& R, uint32_t i) {   // == rev_at(desc, i) for i <= m
  uint32_t v = ring_wrap(R.pe + i, R.m);
  if (R.stale) { if (i == 0) v = R.p0; }  // uniform (scalar) branch, practically never taken
  return v;
}
// The same for a ring whose position is inside the ring (p0 <= delay; reverb_fast_eligible checks it for the twelve rings of the mid stage):
// three scalars per ring instead of five — the mid stage holds twelve of them — and no special case in the address arithmetic.
struct RevRingN {
  double* buf;
  uint32_t pe, m;
};
DEVO RevRingN rev_ringn_uniform(const RevDesc& d) {
  const unsigned long long b = (unsigned long long)d.buf;
  const uint32_t lo = uni_u32((uint32_t)b), hi = uni_u32((uint32_t)(b >> 32));
  RevRingN u;
  u.buf = (double*)(((unsigned long long)hi << 32) | lo);
  u.pe = uni_u32(d.p0);
  u.m = uni_u32(d.delay) + 1;
  return u;
}
DEVO uint32_t ring_at(const RevRingN& R, uint32_t i) { return ring_wrap(R.pe + i, R.m); }
DEVO void ring_advance(RevRing& R, uint32_t T) {  // T in [1, m]
  const uint32_t np = ring_wrap(R.pe + T, R.m);
  R.p0 = np; R.pe = np; R.stale = 0;
}
// element address of (ring position, channel) in the [pos][2] f64 ring
#ifndef REV_TFR_FIXED
#define REV_TFR_FIXED 0
#endif
#if REV_TFR_FIXED
#define REV_TFR_T uint32_t
#define REV_TFR_SET(dst, frac) dst = (uint32_t)((frac) * 4294967296.0)
#define REV_TFR_GET(x) ((double)(x) * 2.3283064365386963e-10)
#else
#define REV_TFR_T double
#define REV_TFR_SET(dst, frac) dst = (frac)
#define REV_TFR_GET(x) (x)
#endif
// (a 32-bit byte offset on the uniform base: the access is `global_load/store v, v_off, s[base]`, no 64-bit address arithmetic per lane)
typedef __attribute__((address_space(1))) char gchar;
DEVO uint32_t ring_off(uint32_t pos, int ch) { return (pos << 4) | ((uint32_t)ch << 3); }
DEVO gdouble* ring_ptr_off(const RevRing& R, uint32_t off) { return (gdouble*)((gchar*)R.buf + (size_t)off); }
DEVO gdouble* ring_ptr(const RevRing& R, uint32_t pos, int ch) { return ring_ptr_off(R, ring_off(pos, ch)); }
DEVO gdouble* ring_ptr_off(const RevRingN& R, uint32_t off) { return (gdouble*)((gchar*)R.buf + (size_t)off); }
DEVO gdouble* ring_ptr(const RevRingN& R, uint32_t pos, int ch) { return ring_ptr_off(R, ring_off(pos, ch)); }

DEVO double rev_guard(float x, uint32_t fpd) {  // reverb.rs:231-236
  double v = (double)x;
  if (fabs(v) < 1.18e-23) v = (double)fpd * 1.18e-17;
  return v;
}

// one ReverbDelayLine::get for one channel at ring count `cnt` and vibrato phase `ph`  (reverb.rs:554-586)
// `sn` = sin(vib_phase)
DEVO double rev_get(const double* buf_generic, uint32_t cnt, uint32_t delay, int ch, double sn, double blend, int32_t* idx_out = nullptr) {
  const gdouble* buf = (const gdouble*)buf_generic;
  double offset = (sn + 1.0) * 7.0;
  double working = (double)cnt + offset;
  double w_floor = floor(working);
  double w_frac = working - w_floor;
  uint32_t w_int = (uint32_t)w_floor;
  uint32_t read_1 = w_int;
  if (read_1 > delay) read_1 -= delay + 1;
  uint32_t read_2 = w_int + 1;
  if (read_2 > delay) read_2 -= delay + 1;
  if (idx_out) *idx_out = (int32_t)read_1;
  double val1 = buf[read_1 * 2 + ch];
  double val2 = buf[read_2 * 2 + ch];
  double interpol = val1 * (1.0 - w_frac) + val2 * w_frac;
  return (1.0 - blend) * interpol + (val1 * blend);
}

// sin / asin of the reverb's two shaping points (reverb.rs:256-257,351-352) on the arguments audio actually produces: |x| <= pi/4 resp.
// |x| <= 1/2 take a short odd polynomial — Taylor to x^15 for sin (truncation < 5e-17 at pi/4), to x^33 for asin (< 3e-13 at 1/2, far
// less below) — instead of the library routine (argument reduction, two polynomials and a select; a rational with a division for asin);
// anything larger goes to the library. The parity gate is 1e-5 RMS on f32 output: both are ~8 orders of magnitude inside it.
#ifndef PG_FAST_SIN
#define PG_FAST_SIN 1
#endif
#ifndef PG_FAST_ASIN
#define PG_FAST_ASIN 0   // measured (three interleaved repetitions on one box): the short sin −1 % kernel time, the 16-term asin chain +0.7 % (a
#endif                   // longer dependent chain than the library's rational): sin on, asin off

DEVO double rev_sin(double x) {
#if PG_FAST_SIN
  if (fabs(x) <= 0.78539816339744828) {
    const double z = x * x;
    double p = -7.6471637318198164759e-13;           // -1/15!
    p = fma(p, z, 1.6059043836821614599e-10);        //  1/13!
    p = fma(p, z, -2.5052108385441718775e-08);       // -1/11!
    p = fma(p, z, 2.7557319223985890653e-06);        //  1/9!
    p = fma(p, z, -1.9841269841269841270e-04);       // -1/7!
    p = fma(p, z, 8.3333333333333333333e-03);        //  1/5!
    p = fma(p, z, -1.6666666666666666667e-01);       // -1/3!
    return fma(x * z, p, x);
  }
#endif
  return sin(x);
}
DEVO double rev_asin(double x) {
#if PG_FAST_ASIN
  if (fabs(x) <= 0.5) {
    // asin x = x + x z (c_1 + z (c_2 + ... + z c_16)),  z = x^2,  c_k = (2k)! / (4^k (k!)^2 (2k+1))
    const double z = x * x;
    double p = 0.004240907093679363;         // c_16
    p = fma(p, z, 0.004660143486915096);     // c_15
    p = fma(p, z, 0.005153309682319905);     // c_14
    p = fma(p, z, 0.005740037670841924);     // c_13
    p = fma(p, z, 0.006447210311889649);     // c_12
    p = fma(p, z, 0.0073125258735988454);    // c_11
    p = fma(p, z, 0.008390335809616815);     // c_10
    p = fma(p, z, 0.009761609529194078);     // c_9
    p = fma(p, z, 0.011551800896139705);     // c_8
    p = fma(p, z, 0.01396484375);            // c_7
    p = fma(p, z, 0.017352764423076924);     // c_6
    p = fma(p, z, 0.022372159090909092);     // c_5
    p = fma(p, z, 0.030381944444444444);     // c_4
    p = fma(p, z, 0.044642857142857144);     // c_3
    p = fma(p, z, 0.075);                    // c_2
    p = fma(p, z, 0.16666666666666666);      // c_1
    return fma(x * z, p, x);
  }
#endif
  return asin(x);
}

// LDS chunk buffer: [frame][2] f64, skewed by one double per 8-frame segment so that both the linear (per sample)
// accesses and the per-segment accesses of the blocked biquad are free of bank conflicts.
#define REV_IDX(n, ch) ((n) * 2 + (ch) + ((n) >> 3))
constexpr int REV_BUF_DOUBLES = 2 * 1024 + 128 + 8;

struct Mat2 { double a, b, c, d; };
DEVO Mat2 mat2_mul(const Mat2& x, const Mat2& y) { return Mat2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; }

// Time-parallel evaluation of one TPT-SVF biquad (BiquadFilter::process_sample, src/utils/dsp/filters/biquad.rs:314-322)
// over the chunk, both channels, in place. The filter is linear and its coefficients are constant inside the chunk:
//   s' = A s + B x,  A = [[2*a1-1, -2*a2], [2*a2, 1-2*a3]],  B = [2*a2, 2*a3]
// Blocked recurrence: 128 segments of 8 frames per channel (one lane each, channel = wave & 1, two waves per channel):
//   pass 1: zero-state response z of each segment (segment 0 starts from the carried state);
//   scan  : the state in front of every segment, s_L = sum_{j < L} (A^8)^(L-1-j) z_j, in three levels that stay in registers:
//           inside a row of 16 lanes four doubling steps over DPP row shifts (zero fill at the row's start); the four row totals of a
//           wave come by v_readlane and are walked with A^128 (by every lane — the values are uniform); the state entering a row
//           reaches lane m of the row through (A^8)^m, applied as the binary product of the doubling matrices. Only the hand-over from
//           a channel's lower wave to its upper wave goes through LDS (the function's one barrier; all four waves scan their rows at
//           the same time).
//           (The first version ran a 6-step Kogge-Stone scan over ds_bpermute shuffles in the lower wave and then, seeded with its
//           result, in the upper wave: ~6 K cycles per filter and block, three filters per reverb block.)
//   pass 2: each segment re-run from its true start state, writing the outputs.
// Same arithmetic as the serial recurrence up to f64 rounding (|error| ~ 1e-16 relative). The caller puts a barrier behind the call
// (the carried state and `xchg` are rewritten by the next one).
template <int CTRL>
DEVO double dpp_zero_f64(double x) {  // the DPP-selected lane's value; 0 where the selection leaves the row
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
DEVO double readlane_f64(double x, int l) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
template <bool ROUND_F32>
DEVO void rev_biquad_scan_t(const PgBiquadCoef& c, PgState2* st, double* buf, int T, double* xchg /* LDS [2][2] */) {
#pragma clang fp contract(fast)  // mul + add pairs may fuse here (as in the mid stage): the scan reassociates the recurrence anyway, |error| ~ 1e-16 relative
  const int tid = pg_tid();
  const int wave = tid >> 6, lane = tid & 63;
  const int ch = wave & 1, half = wave >> 1;
  const int seg = half * 64 + lane;
  const int n0 = seg * 8;
  const int len = n0 >= T ? 0 : (T - n0 < 8 ? T - n0 : 8);
  const double a1 = c.a1, a2 = c.a2, a3 = c.a3, m0 = c.m0, m1 = c.m1, m2 = c.m2;
  // REV_IDX(n0 + k, ch) = REV_IDX(n0, ch) + 2k inside a segment: both passes are unrolled over constant LDS offsets. (The inputs are read
  // again in pass 2 rather than held in 16 registers: the tail stage is an out-of-line function that must fit the caller-saved registers.)
  double* seg_buf = buf + REV_IDX(n0, ch);
  // pass 1
  const double c1 = st[ch].ic1eq, c2 = st[ch].ic2eq;   // carried state
  double s1 = 0.0, s2 = 0.0;
  if (seg == 0) { s1 = c1; s2 = c2; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < len) {
      const double v3 = seg_buf[2 * k] - s2;
      const double v1 = a1 * s1 + a2 * v3;
      const double v2 = s2 + a2 * s1 + a3 * v3;
      s1 = 2.0 * v1 - s1;
      s2 = 2.0 * v2 - s2;
    }
  }
  // powers of the transition matrix: P[k] = (A^8)^(2^k). A = [[p, -q], [q, r]], and that shape is closed under multiplication: b = -c in every
  // power, a square costs six operations
  Mat2 P[5];
  {
    double pa = 2.0 * a1 - 1.0, pc = 2.0 * a2, pd = 1.0 - 2.0 * a3;
    auto sq = [&]() { const double tr = pa + pd, c2 = pc * pc; pa = pa * pa - c2; pd = pd * pd - c2; pc = pc * tr; };
    sq(); sq(); sq();  // A^8
    P[0] = Mat2{pa, -pc, pc, pd};
#pragma unroll
    for (int k = 1; k < 5; ++k) { sq(); P[k] = Mat2{pa, -pc, pc, pd}; }
  }
  // rows of 16 lanes: z_L <- sum over the row's lanes j <= L of (A^8)^(L-j) z_j
  {
    double y1, y2;
    y1 = dpp_zero_f64<0x111>(s1); y2 = dpp_zero_f64<0x111>(s2); s1 = s1 + (P[0].a * y1 + P[0].b * y2); s2 = s2 + (P[0].c * y1 + P[0].d * y2);   // row_shr:1
    y1 = dpp_zero_f64<0x112>(s1); y2 = dpp_zero_f64<0x112>(s2); s1 = s1 + (P[1].a * y1 + P[1].b * y2); s2 = s2 + (P[1].c * y1 + P[1].d * y2);   // row_shr:2
    y1 = dpp_zero_f64<0x114>(s1); y2 = dpp_zero_f64<0x114>(s2); s1 = s1 + (P[2].a * y1 + P[2].b * y2); s2 = s2 + (P[2].c * y1 + P[2].d * y2);   // row_shr:4
    y1 = dpp_zero_f64<0x118>(s1); y2 = dpp_zero_f64<0x118>(s2); s1 = s1 + (P[3].a * y1 + P[3].b * y2); s2 = s2 + (P[3].c * y1 + P[3].d * y2);   // row_shr:8
  }
  // the state entering each row of a wave: R[0] = what the wave starts from, R[r + 1] = A^128 R[r] + (total of row r)
  const int row = lane >> 4;
  const double t1[4] = {readlane_f64(s1, 15), readlane_f64(s1, 31), readlane_f64(s1, 47), readlane_f64(s1, 63)};
  const double t2[4] = {readlane_f64(s2, 15), readlane_f64(s2, 31), readlane_f64(s2, 47), readlane_f64(s2, 63)};
  double r1 = 0.0, r2 = 0.0;    // entry state of this lane's row
  auto walk = [&](double w1, double w2) {   // returns nothing; leaves the wave's final state in (w1, w2) through the last lambda argument copy
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (row == r) { r1 = w1; r2 = w2; }
      const double n1 = t1[r] + (P[4].a * w1 + P[4].b * w2), n2 = t2[r] + (P[4].c * w1 + P[4].d * w2);
      w1 = n1; w2 = n2;
    }
    if (half == 0 && lane == 0) { xchg[ch * 2] = w1; xchg[ch * 2 + 1] = w2; }
  };
  if (half == 0) walk(0.0, 0.0);
  __syncthreads();
  if (half == 1) walk(xchg[ch * 2], xchg[ch * 2 + 1]);
  // start state of the segment: the row-local part (the scan value of the lane below) + (A^8)^(lane in row) applied to the row's entry state
  double b1 = dpp_zero_f64<0x111>(s1), b2 = dpp_zero_f64<0x111>(s2);
  {
    const int m = lane & 15;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double q1 = P[k].a * r1 + P[k].b * r2, q2 = P[k].c * r1 + P[k].d * r2;
      if (m & (1 << k)) { r1 = q1; r2 = q2; }
    }
    b1 = b1 + r1; b2 = b2 + r2;
  }
  if (seg == 0) { b1 = c1; b2 = c2; }
  // pass 2
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < len) {
      const double v0 = seg_buf[2 * k];
      const double v3 = v0 - b2;
      const double v1 = a1 * b1 + a2 * v3;
      const double v2 = b2 + a2 * b1 + a3 * v3;
      b1 = 2.0 * v1 - b1;
      b2 = 2.0 * v2 - b2;
      const double y = m0 * v0 + m1 * v1 + m2 * v2;
      seg_buf[2 * k] = ROUND_F32 ? (double)(float)y : y;  // `as f32` between cascaded stages (eq5.rs:318-320)
    }
  }
  if (len > 0 && n0 + len == T) { st[ch].ic1eq = b1; st[ch].ic2eq = b2; }   // (every lane read the carried state before the barrier)
}

DEVO void rev_biquad_scan(const PgBiquadCoef& c, PgState2* st, double* buf, int T, double* xchg) { rev_biquad_scan_t<false>(c, st, buf, T, xchg); }

// steady state and a geometry whose shortest feedback lag leaves room for a chunk (always true for valid room sizes)
DEVO bool reverb_geometry_ok(const PgFx& fx) {
  const PgReverb& r = fx.u.reverb;
  double rs = (double)r.room.target;
  double size = (rs * rs * 75.0) + 25.0;
  if (!(d2u64(29.0 * size) >= 64 && d2u64(47.0 * size) >= 64 + 17)) return false;
  // every ring position inside its ring: after the room shrank a position may sit above the new ring end for one more frame (the next
  // write lands there, then the walk restarts at 0) — the serial path renders that block
  // (against the ring lengths the next block will set from the room size it runs with — update_delay_sizes, reverb.rs:196-213 — not the
  // ones the last block of a ramp left behind)
  const double k[8] = {79.0, 73.0, 71.0, 67.0, 61.0, 59.0, 53.0, 47.0};
  for (int i = 0; i < 8; ++i) { const uint32_t dl = (uint32_t)d2u64(k[i] * size), mx = r.line[i].frames - 1; if (r.line[i].count > (dl < mx ? dl : mx)) return false; }
  const double ka[4] = {43.0, 41.0, 37.0, 31.0};
  for (int i = 0; i < 4; ++i) { const uint32_t dl = (uint32_t)d2u64(ka[i] * size), mx = r.ap[i].frames - 1; if (r.ap[i].write_pos > (dl < mx ? dl : mx)) return false; }
  return true;
}
DEVO bool reverb_fast_eligible(const PgFx& fx) {
  const PgReverb& r = fx.u.reverb;
  return !sm_need_ramp(r.room) && !sm_need_ramp(r.wet) && reverb_geometry_ok(fx);
}
// The same test made by the wave (all lanes of the workgroup call, every wave gets the same answer): lane l looks at ring l mod 16 of the twelve
// and one ballot joins the answers — the serial walk is twelve dependent LDS reads and f64 conversions in a row, ~3 K cycles at the head of every
// block of a reverb on the unit kernels and on the main mixer's chain (tools/diag_bus_chain.py).
DEVO bool reverb_fast_eligible_wave(const PgFx& fx) {
  const PgReverb& r = fx.u.reverb;
  const double rs = (double)r.room.target;
  const double size = (rs * rs * 75.0) + 25.0;
  bool ok = !sm_need_ramp(r.room) && !sm_need_ramp(r.wet) && d2u64(29.0 * size) >= 64 && d2u64(47.0 * size) >= 64 + 17;
  const int i = (int)(threadIdx.x & 15u);
  if (i < 12) {
    // 79 73 71 67 61 59 53 47 | 43 41 37 31 (reverb.rs:196-213), picked without a table (a private array indexed per lane lives in scratch)
    const int kk = i == 0 ? 79 : i == 1 ? 73 : i == 2 ? 71 : i == 3 ? 67 : i == 4 ? 61 : i == 5 ? 59 : i == 6 ? 53 : i == 7 ? 47 : i == 8 ? 43 : i == 9 ? 41 : i == 10 ? 37 : 31;
    const uint32_t dl = (uint32_t)d2u64((double)kk * size);
    const uint32_t frames = i < 8 ? r.line[i < 8 ? i : 0].frames : r.ap[i < 8 ? 0 : i - 8].frames;
    const uint32_t pos = i < 8 ? r.line[i < 8 ? i : 0].count : r.ap[i < 8 ? 0 : i - 8].write_pos;
    const uint32_t mx = frames - 1;
    if (pos > (dl < mx ? dl : mx)) ok = false;
  }
  return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}
// `wet` ramping, room size at rest (reverb_wet_ramp_fast below): the ring geometry stands still, only the wet gain and the three low-pass
// cutoffs (10000 - room * wet * 3000 Hz, reverb.rs:413-424) move per frame
DEVO bool reverb_wet_ramp_eligible(const PgFx& fx) {
  const PgReverb& r = fx.u.reverb;
  return !sm_need_ramp(r.room) && sm_need_ramp(r.wet) && reverb_geometry_ok(fx);
}

// ---- the three stages of the time-parallel reverb ------------------------------------------------------------------------
// front : predelay ring (chunks <= predelay) + biquad A over the whole piece            -> bufA
// mid   : allpasses + vibrato lines in chunks (phase records, anchors, sub-chunks)      bufA -> bufA (o of every frame)
// tail  : biquad B -> clamp/asin -> biquad C -> dry mix                                 bufA -> signal
// The front does not depend on the mid stage of the same piece and the tail only on the mid stage's output, so the fused
// kernel runs them back to back on one LDS buffer and the staged pipeline (pg_stage*_kernel) runs each at its own occupancy,
// handing bufA over through HBM.
constexpr int REV_T_CAP = 1024;  // frames per piece (capacity of bufA)

struct RevLds {   // LDS carve-up of the reverb arena
  double* bufA;   // [REV_T_CAP][2] f64, skewed (REV_IDX)
  RevRec* rec;    // [16]
  double* gl;     // [16] epilogue gets / uniform scalars
  RevDesc* desc;  // [13]
  double* xchg;   // biquad scan hand-over
  double* anch;   // [9][16] {sin, cos} anchors: 8 sub-chunks + the epilogue
  double* vtab;   // vibrato rotation table (LDS copy): [8][REV_VTAB_N] {cos, sin}
  char* slack;    // 64 bytes
};
// The mid stage's tables (anchors, rotation table, slack: REV_TABLES_BYTES) follow the prefix bufA .. xchg, or start at `tables` when the
// caller keeps something of its own behind the prefix (the staged kernels: the unit's dry signal stays in LDS through the mid stage).
constexpr int REV_VTAB_HALF = 64;               // sub-chunks are 128 frames around their anchor: |j| <= 64
constexpr int REV_VTAB_N = REV_VTAB_HALF + 1;   // entries per line held in LDS (the table in HBM has 129 per line)
constexpr size_t REV_TABLES_BYTES = 9 * 16 * 2 * 8 + 8 * REV_VTAB_N * 2 * 8 + 64;
DEVO RevLds rev_lds(char* scratch, char* tables = nullptr) {
  RevLds m;
  m.bufA = (double*)scratch;
  char* lp = scratch + REV_BUF_DOUBLES * 8;
  m.rec = (RevRec*)lp;    lp += 16 * sizeof(RevRec);
  m.gl = (double*)lp;     lp += 16 * 8;
  m.desc = (RevDesc*)lp;  lp += 13 * sizeof(RevDesc);
  m.xchg = (double*)lp;   lp += 4 * 8;
  if (tables) lp = tables;
  m.anch = (double*)lp;   lp += 9 * 16 * 2 * 8;
  m.vtab = (double*)lp;   lp += 8 * REV_VTAB_N * 2 * 8;
  m.slack = lp;
  return m;
}

struct RevBlock {  // per-block uniform parameters
  double blend, regen, wet;
  uint32_t predelay, t_mid;
};

// vibrato rotation table -> LDS: entries 0 .. REV_VTAB_HALF of each line's {cos, sin} pairs (the rotation by -j is the transpose: cos is even,
// sin odd), 16-byte loads all in flight before the first LDS store (caller syncs)
DEVO void rev_load_vtab(const PgReverb& r, const RevLds& m) {
  const int tid = pg_tid();
  const gdouble* tg = (const gdouble*)r.vib_tab;
  constexpr int TRIPS = (8 * REV_VTAB_N + 255) / 256;
  double t0[TRIPS], t1[TRIPS];
#pragma unroll
  for (int k = 0; k < TRIPS; ++k) {
    const int i = tid + k * 256;
    const int line = i / REV_VTAB_N, j = (i < 8 * REV_VTAB_N) ? line * 129 + (i - line * REV_VTAB_N) : 0;
    t0[k] = tg[2 * j]; t1[k] = tg[2 * j + 1];
  }
#pragma unroll
  for (int k = 0; k < TRIPS; ++k) { const int i = tid + k * 256; if (i < 8 * REV_VTAB_N) { m.vtab[2 * i] = t0[k]; m.vtab[2 * i + 1] = t1[k]; } }
}

// (Measured and not kept, round 5: the table requested by LDS-DMA from stage 1 of the staged single launch, under the predelay — workgroup 0's
// stage-2 set-up got 1.5 K cycles shorter, the launch 1.5-3 % LONGER at every callback size: 2080 dword-granular transfers per block and a
// second spilled register. profiles/r05_ab_vtab_prefetch.txt)
// block parameters (reverb.rs:429-440): delay lengths, blend/regen, the three low-pass coefficient sets; cached in the effect
// state while room size and wet stay put. All lanes call; returns false for a degenerate geometry (serial path).
DEVO bool rev_block_params(PgFx& fx, const RevLds& m, int* ctl, RevBlock& b) {
  PgReverb& r = fx.u.reverb;
  const int tid = pg_tid();
  __syncthreads();
  if (tid == 0) {
    if (!(r.cache_valid && r.cache_room == r.room.target && r.cache_wet == r.wet.target)) {
      ReverbBlock rb;
      reverb_params(fx, (double)r.room.target, (double)r.wet.target, rb);
      r.c_blend = rb.blend; r.c_regen = rb.regen; r.c_predelay = rb.predelay;
      r.cache_room = r.room.target; r.cache_wet = r.wet.target; r.cache_valid = 1;
    }
    m.gl[0] = r.c_blend; m.gl[1] = r.c_regen; m.gl[2] = (double)r.wet.target;
    ctl[4] = (int)r.c_predelay;
  }
  __syncthreads();
  b.blend = uni_f64(m.gl[0]); b.regen = uni_f64(m.gl[1]); b.wet = uni_f64(m.gl[2]);
  b.predelay = uni_u32((uint32_t)ctl[4]);
  __syncthreads();
  // chunk length bound of the mid stage: all its reads must hit pre-chunk data
  uint32_t t_mid = 0xffffffffu;
  for (int i = 0; i < 4; ++i) t_mid = t_mid < r.ap[i].delay ? t_mid : r.ap[i].delay;
  for (int i = 0; i < 8; ++i) { uint32_t d = r.line[i].delay > 17 ? r.line[i].delay - 17 : 0; t_mid = t_mid < d ? t_mid : d; }
  if (t_mid > (uint32_t)REV_T_CAP) t_mid = REV_T_CAP;
  b.t_mid = uni_u32(t_mid);
  return b.t_mid >= 32 && b.predelay >= 32;
}

// The same parameters read back from the effect's state block (an LDS copy that rev_block_params has validated in THIS block — the staged
// single launch, stages 2 and 3): no barrier, no lane-0 work; the values are uniform and go to scalar registers.
DEVO void rev_block_params_cached(const PgFx& fx, RevBlock& b) {
  const PgReverb& r = fx.u.reverb;
  b.blend = uni_f64(r.c_blend); b.regen = uni_f64(r.c_regen); b.wet = uni_f64((double)r.wet.target);
  b.predelay = uni_u32(r.c_predelay);
  uint32_t t_mid = 0xffffffffu;
  for (int i = 0; i < 4; ++i) t_mid = t_mid < r.ap[i].delay ? t_mid : r.ap[i].delay;
  for (int i = 0; i < 8; ++i) { uint32_t d = r.line[i].delay > 17 ? r.line[i].delay - 17 : 0; t_mid = t_mid < d ? t_mid : d; }
  if (t_mid > (uint32_t)REV_T_CAP) t_mid = REV_T_CAP;
  b.t_mid = uni_u32(t_mid);
}

// ---- front: predelay (DelayLine<2>::process, delay.rs:47-66) in chunks of <= predelay frames, then biquad A ----
template <bool SCAN_A = true>
DEVO void rev_front(PgReverb& r, const float* s0, int T, const RevLds& m, const RevBlock& b, unsigned long long* diag) {
  const int tid = pg_tid(), nt = blockDim.x;
  double* bufA = m.bufA;
  const int ch0 = tid & 1;  // nt is even: a lane keeps its channel across trips
  const uint32_t fpd = ch0 ? r.fpd_r : r.fpd_l;
  RevRing pd = rev_ring_uniform(RevDesc{r.pre, r.pre_write_pos & r.pre_mask, b.predelay});
  __syncthreads();  // every lane holds the ring position before lane 0 may advance it
  for (int c0 = 0; c0 < T; c0 += (int)b.predelay) {
    const int Tc = T - c0 < (int)b.predelay ? T - c0 : (int)b.predelay;
    for (int s_base = 0; s_base < 2 * Tc; s_base += 8 * nt) {  // 8 samples per lane and trip: all loads in flight together
      double pv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int s = s_base + k * nt + tid;
        pv[k] = s < 2 * Tc ? *ring_ptr(pd, ring_at(pd, (s >> 1) + 1), ch0) : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int s = s_base + k * nt + tid;
        if (s < 2 * Tc) bufA[REV_IDX(c0 + (s >> 1), ch0)] = pv[k];
      }
    }
    __syncthreads();
    for (int s_base = 0; s_base < 2 * Tc; s_base += 8 * nt) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int s = s_base + k * nt + tid;
        if (s < 2 * Tc) *ring_ptr(pd, ring_at(pd, s >> 1), ch0) = rev_guard(s0[2 * c0 + s], fpd);
      }
    }
    ring_advance(pd, (uint32_t)Tc);
    if (c0 + Tc < T) { __threadfence_block(); __syncthreads(); }  // the next chunk reads what this one wrote
  }
  if (tid == 0) r.pre_write_pos = pd.p0;
  PG_STAMP(diag, 3);
  if (SCAN_A) rev_biquad_scan(r.ca, r.sa, bufA, T, m.xchg);
  __syncthreads();
}

// ---- mid: allpasses + vibrato lines ----
// idx_log (test hook, nullptr in the kernels that matter for speed): slot ((frame * 8 + line) * 2 + channel) receives `read_1` of that
// frame's ReverbDelayLine::get (reverb.rs:563-570) — the index stream SURVEY §8c asks to be compared separately from the samples.
// wet_seq (nullptr in the steady-state kernels): the wet gain of every frame of the piece while its smoother moves (reverb_wet_ramp_fast).
template <int ITEMS = 1>
DEVO void rev_mid(PgReverb& r, int frames, const RevLds& m, const RevBlock& b, int* ctl, unsigned long long* diag, int32_t* idx_log = nullptr,
                  const float* wet_seq = nullptr) {
  const int tid = pg_tid(), nt = blockDim.x;
  RevRec* rec = m.rec; double* gl = m.gl; RevDesc* desc = m.desc; double* anch = m.anch; double* vtab = m.vtab;
  const double blend = b.blend, regen = b.regen, wet = b.wet;
  const double omb = uni_f64(1.0 - blend);   // (f64 arithmetic is VALU work: back into scalar registers)
#ifdef PG_DIAG
  unsigned long long* lapacc = (unsigned long long*)m.slack;  // 8 lap accumulators in the scratch slack
  if (tid == 0) for (int i = 0; i < 8; ++i) lapacc[i] = 0;
#endif
  int done = 0;
  while (done < frames) {
    double* bufA = m.bufA + 0;  // chunk frames are addressed as done + n below
    // ---- chunk set-up: vibrato phase records + chunk length (16 lanes), ring descriptors (13 lanes) ----
    if (tid < 16) {
      const PgReverbLine& l = r.line[tid >> 1];
      const double d = l.depth * 0.1;
      RevRec rc;
      rc.p0 = l.vib_phase[tid & 1];
      rc.m0 = rev_phase_piece(rc.p0, d, rc.du0);
      rc.p1 = (rc.p0 + (double)rc.m0 * rc.du0) + d;  // the plain hardware add carries the phase across the binade edge
      rc.m1 = rev_phase_piece(rc.p1, d, rc.du1);
      rec[tid] = rc;
      ctl[8 + tid] = (int)(rc.m0 + 1u + rc.m1);
    } else if (tid >= 64 && tid < 64 + 8) desc[tid - 64] = RevDesc{r.line[tid - 64].buf, r.line[tid - 64].count, r.line[tid - 64].delay};
    else if (tid >= 64 + 8 && tid < 64 + 12) desc[tid - 64] = RevDesc{r.ap[tid - 72].buf, r.ap[tid - 72].write_pos, r.ap[tid - 72].delay};
    __syncthreads();
    PG_STAMP(diag, 12);
    PG_STAMP_VAL(diag, 20 + (done > 0 ? 1 : 0), done);
    PG_STAMP_VAL(diag, 22, b.t_mid);
#ifndef PG_DIAG_SCHED   // (tools/diag_stamps.py: the schedule's own stamps use slots 30-35)
    for (int i = 0; i < 16; ++i) PG_STAMP_VAL(diag, 24 + i, ctl[8 + i]);
#endif
    int T = frames - done;
    if ((uint32_t)T > b.t_mid) T = (int)b.t_mid;
    for (int i = 0; i < 16; ++i) T = T < ctl[8 + i] ? T : ctl[8 + i];
    T = (int)uni_u32((uint32_t)T);
    PG_STAMP(diag, 2);
    RevRingN D[12];  // the ring descriptors as scalars (all uses below index them with compile-time constants)
#pragma unroll
    for (int i = 0; i < 12; ++i) D[i] = rev_ringn_uniform(desc[i]);

    // ---- vibrato anchors for the whole chunk, one lane per (sub-chunk, line, channel): sin/cos (accurate libm) of the exact
    // phase of the sub-chunk's middle item (or the chunk's end, if that comes first); inside the sub-chunk sin(phase_n) follows by
    // the angle-addition rotation by j = n - anchor steps, |j| <= 64, with the per-line table. Slot 8 = the epilogue's phase (after
    // the chunk's last step).
    if (tid < 9 * 16) {
      const int sub = tid >> 4, lc = tid & 15;
      const int mid_n = sub * (nt / 2) + REV_VTAB_HALF;
      const uint32_t nb = (sub == 8 || mid_n > T) ? (uint32_t)T : (uint32_t)mid_n;
      if (sub == 8 || sub * (nt / 2) < T) {
        const RevRec rc = rec[lc];
        const double pb = rev_phase_at(rc, nb);
        anch[tid * 2] = sin(pb);
        anch[tid * 2 + 1] = cos(pb);
        if (sub == 8) gl[lc] = pb;
      }
    }
    __syncthreads();
    PG_STAMP(diag, 4);
    // ---- sub-chunks of nt/2 frames ----
    // ITEMS consecutive 128-frame sub-chunks per trip (each with its own anchor): every lane carries ITEMS (frame, channel) items through the
    // reads -> barrier -> writes sequence. 1 in the kernels that share a CU four ways (128 VGPRs); 4 in the generic kernel, whose lone
    // workgroups (the main mixer's chain, units on the exact lane) are latency chains with registers to spare: a quarter of the dependent round trips.
    for (int base = 0; base < T; base += ITEMS * (nt / 2)) {
#pragma clang fp contract(fast)  // the only place mul+add pairs may fuse: measured faster, error ~1e-16 relative (the parity gate is 1e-5 RMS)
      int lane_frame = tid >> 1;
      asm volatile("" : "+v"(lane_frame));  // keeps per-lane ring addresses from being hoisted out of the loop into 36 long-lived VGPRs
      int ch = tid & 1;
      asm volatile("" : "+v"(ch));
      double sv[ITEMS][8];   // values the eight lines are `set` to (allpass tap + feedback), reverb.rs:275-282,588-594
      double apw[ITEMS][4];  // values written into the four allpass rings
      double o_prev[ITEMS];
      PG_LAP_DECL(lap_t);
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
      const int sub_base = base + it * (nt / 2);
      const int n = sub_base + lane_frame;
      const bool active = n < T;
      o_prev[it] = 0.0;
      if (active) {
        // Order matters for latency: first everything that only needs the (known) vibrato phases — the 16 line taps of
        // the previous frame's `get` — and the 4 allpass taps go out to HBM; the f64 sin of the front end and the allpass
        // chain then run underneath those loads.
        double tv1[8], tv2[8];
        // interpolation fraction: REV_TFR_FIXED = 1 holds it as 0.32 fixed point (2^-33 error) — 8 VGPRs less, 4 VALU instructions per line
        // more (a spilled tap would put an `s_waitcnt vmcnt(0)` in the middle of the load issue); 0 = the f64 fraction itself
        REV_TFR_T tfr[8];
        if (n >= 1) {
          // sin(phase_n) ~= sin(pb + j*d): pb = the sub-chunk's exact anchor phase, j*d = tabulated rotation. The reference's
          // accumulator advances by du = d rounded to the accumulator's ulp; the neglected j*(du - d) is <= 64 * 2^-52 * |p|
          // (< 3e-14 in the tap position), far below the 1-ulp spread between libm implementations of sin itself.
          const int n_anchor = sub_base + REV_VTAB_HALF < T ? sub_base + REV_VTAB_HALF : T;
          const int js = n - n_anchor;                      // in [-64, 63]
          const int jb = js < 0 ? -js : js;
          const unsigned long long st_sign = js < 0 ? 0x8000000000000000ull : 0ull;  // sin(-x) = -sin(x)
          const double* anb = anch + (size_t)(sub_base / (nt / 2)) * 16 * 2;
          // The rotation operands of line i + 1 are fetched from LDS before line i is evaluated (two register sets): one
          // exposed LDS round trip per sub-chunk instead of one per line.
          double q_ct[2], q_st[2], q_a0[2], q_a1[2];
          q_ct[0] = vtab[jb * 2]; q_st[0] = vtab[jb * 2 + 1]; q_a0[0] = anb[ch * 2]; q_a1[0] = anb[ch * 2 + 1];
#pragma unroll
          for (int i = 0; i < 8; ++i) {  // ReverbDelayLine::get, address part (reverb.rs:563-576); count = position of frame n
            const RevRingN ld = D[i];
            if (i + 1 < 8) {
              const double* an = anb + ((i + 1) * 2 + ch) * 2;
              q_ct[(i + 1) & 1] = vtab[((i + 1) * REV_VTAB_N + jb) * 2]; q_st[(i + 1) & 1] = vtab[((i + 1) * REV_VTAB_N + jb) * 2 + 1];
              q_a0[(i + 1) & 1] = an[0]; q_a1[(i + 1) & 1] = an[1];
            }
            const double st = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(q_st[i & 1]) ^ st_sign));
            const double sn = fma(q_a0[i & 1], q_ct[i & 1], q_a1[i & 1] * st);
            const double working = (double)ring_at(ld, n) + (sn + 1.0) * 7.0;   // > 0: truncation is floor, v_fract_f64 is working - floor(working)
            REV_TFR_SET(tfr[i], __builtin_amdgcn_fract(working));
            const uint32_t w_int = (uint32_t)working;                 // < count + 15 <= delay + 15 < 2 * (delay + 1)
            if (idx_log) idx_log[((done + n - 1) * 8 + i) * 2 + ch] = (int32_t)ring_wrap(w_int, ld.m);  // (the get of frame n - 1)
            const uint32_t o1 = ring_off(ring_wrap(w_int, ld.m), ch);  // `if read > delay { read -= delay + 1 }`
            tv1[i] = *ring_ptr_off(ld, o1);
            tv2[i] = *ring_ptr_off(ld, ring_wrap(o1 + 16u, ld.m << 4));  // the next ring position, same channel: one wrap in the byte domain
          }
        }
        PG_LAP(diag, 50, lap_t);
        double dl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const RevRingN a = D[8 + i]; dl[i] = *ring_ptr(a, ring_at(a, n + 1), ch); }  // `delayed`
        // front: wet gain, sin, Schroeder allpass chain i -> j -> k -> l (reverb.rs:253-263; delay.rs:314-350)
        double apo[4];
        double v = rev_sin(bufA[REV_IDX(done + n, ch)] * (wet_seq ? (double)wet_seq[done + n] : wet));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double bb = v - (dl[i] * 0.5);
          apw[it][i] = bb;
          v = bb * 0.5 + dl[i];   // == buf*0.5 + new_delayed (delay >= 1)
          apo[i] = v;
        }
        PG_LAP(diag, 51, lap_t);
        double F[8];
        if (n >= 1) {
          double g[8];
#pragma unroll
          // interpolation + blend (reverb.rs:578-583): (1 - blend) (tv1 (1 - fr) + tv2 fr) + tv1 blend = tv1 + (1 - blend) fr (tv2 - tv1) — three
          // instructions per line instead of five; the feedback matrix (reverb.rs:303-310) g_i - (sum of the other three) = 2 g_i - S with S formed
          // once per group of four, the output (reverb.rs:321-329) from the two group sums: 24 instructions instead of 40. Same values up to
          // the last place of f64 (the kernel's VALU is busy 45-50 % of the time, DESIGN §7: instructions saved here are time)
          for (int i = 0; i < 8; ++i) g[i] = tv1[i] + (omb * REV_TFR_GET(tfr[i])) * (tv2[i] - tv1[i]);
          const double Sa = (g[0] + g[1]) + (g[2] + g[3]), Sb = (g[4] + g[5]) + (g[6] + g[7]);
          o_prev[it] = (Sa + Sb) * 0.125;
          F[0] = (g[0] * 2.0 - Sa) * regen; F[1] = (g[1] * 2.0 - Sa) * regen; F[2] = (g[2] * 2.0 - Sa) * regen; F[3] = (g[3] * 2.0 - Sa) * regen;
          F[4] = (g[4] * 2.0 - Sb) * regen; F[5] = (g[5] * 2.0 - Sb) * regen; F[6] = (g[6] * 2.0 - Sb) * regen; F[7] = (g[7] * 2.0 - Sb) * regen;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) F[i] = r.line[i].feedback[ch];  // handed over from the previous chunk
        }
        // set(): a<-l, b<-k, c<-j, d<-i, e<-i, f<-j, g<-k, h<-l  (reverb.rs:275-282)
        sv[it][0] = apo[3] + F[0]; sv[it][1] = apo[2] + F[1]; sv[it][2] = apo[1] + F[2]; sv[it][3] = apo[0] + F[3];
        sv[it][4] = apo[0] + F[4]; sv[it][5] = apo[1] + F[5]; sv[it][6] = apo[2] + F[6]; sv[it][7] = apo[3] + F[7];
      }
      }
      PG_LAP(diag, 52, lap_t);
      __syncthreads();  // every read of these sub-chunks has been issued and consumed
      PG_LAP(diag, 53, lap_t);
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int n = base + it * (nt / 2) + lane_frame;
        if (n < T) {
          if (n >= 1) bufA[REV_IDX(done + n - 1, ch)] = o_prev[it];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const RevRingN a = D[8 + i]; *ring_ptr(a, ring_at(a, n), ch) = apw[it][i]; }
#pragma unroll
          for (int i = 0; i < 8; ++i) { const RevRingN ld = D[i]; *ring_ptr(ld, ring_at(ld, n), ch) = sv[it][i]; }
        }
      }
      PG_LAP(diag, 54, lap_t);
      // no barrier here: the next trip only reads ring positions that are written by its own or later items, and LDS
      // slots >= its first frame, so its reads cannot collide with these writes
    }
    __syncthreads();
    PG_STAMP(diag, 5);
    // ---- epilogue: gets after the step of the chunk's last frame (16 lanes: one (line, channel) each) ----
    if (tid < 16) {
      const int i = tid >> 1, ch = tid & 1;
      const RevDesc ld = desc[i];
      const double ph = gl[tid];  // exact phase after the chunk's last step (anchor slot 8)
      const double sn = anch[(8 * 16 + tid) * 2];
      r.line[i].vib_phase[ch] = ph;
      gl[tid] = rev_get(ld.buf, rev_at(ld, T), ld.delay, ch, sn, blend, idx_log ? idx_log + ((done + T - 1) * 8 + i) * 2 + ch : nullptr);
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
      bufA[REV_IDX(done + T - 1, ch)] = (g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7]) / 8.0;
    }
    if (tid >= 64 && tid < 64 + 8) r.line[tid - 64].count = rev_at(desc[tid - 64], T);  // advance ring positions (uniform integer bookkeeping)
    else if (tid >= 64 + 8 && tid < 64 + 12) r.ap[tid - 72].write_pos = rev_at(desc[tid - 64], T);
    __syncthreads();
    PG_STAMP(diag, 6);
    done += T;
  }
#ifdef PG_DIAG
  if (diag && blockIdx.x == 0 && tid == 0) for (int i = 0; i < 8; ++i) diag[50 + i] += lapacc[i];
#endif
}

// ---- tail: biquad B -> clamp -> asin -> biquad C -> dry mix (reverb.rs:340-368) ----
// DRY_DMA: the dry signal is still on its way into s0 (lds_dma_dword transfers issued by the staged kernel when the stage began):
// the wait sits behind the two scans, in front of the barrier that precedes the first read.
template <bool DRY_DMA>
DEVO void rev_tail_impl(PgReverb& r, float* s0, int T, const RevLds& m, const RevBlock& b, unsigned long long* diag) {
  const int tid = pg_tid(), nt = blockDim.x;
  double* bufA = m.bufA;
  const double wet = b.wet;
  PG_STAMP(diag, 56);
  rev_biquad_scan(r.cb, r.sb, bufA, T, m.xchg);
  __syncthreads();
  PG_STAMP(diag, 57);
  for (int s = tid; s < 2 * T; s += nt) { const int bi = REV_IDX(s >> 1, s & 1); bufA[bi] = rev_asin(clampd(bufA[bi], -1.0, 1.0)); }
  __syncthreads();
  PG_STAMP(diag, 58);
  rev_biquad_scan(r.cc, r.sc, bufA, T, m.xchg);
  if (DRY_DMA) lds_dma_wait();
  __syncthreads();
  PG_STAMP(diag, 59);
  for (int s = tid; s < 2 * T; s += nt) {
    double y = bufA[REV_IDX(s >> 1, s & 1)];
    if (wet != 1.0) y += rev_guard(s0[s], (s & 1) ? r.fpd_r : r.fpd_l) * (1.0 - wet);
    s0[s] = (float)y;
  }
  __syncthreads();
  PG_STAMP(diag, 7);
}
DEVO void rev_tail(PgReverb& r, float* s0, int T, const RevLds& m, const RevBlock& b, unsigned long long* diag) { rev_tail_impl<false>(r, s0, T, m, b, diag); }

template <int ITEMS = 1>
DEVO bool reverb_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  PgReverb& r = fx.u.reverb;
  // per-frame delay sizes / coefficients while a smoother moves, or a ring position still above a ring end the room left behind when it
  // shrank (the mid stage's ring arithmetic has no case for it): exact serial path
  // (ITEMS >= 2: the generic kernel — lone workgroups, registers to spare; inlined into the wide fast kernel the wave's test cost it its three-per-CU budget)
  if (!(ITEMS >= 2 ? reverb_fast_eligible_wave(fx) : reverb_fast_eligible(fx))) return false;
  const int frames = n_samples / 2;
  if (frames == 0) return true;
  const RevLds m = rev_lds(fc.scratch);
  rev_load_vtab(r, m);
  RevBlock b;
  if (!rev_block_params(fx, m, fc.ctl, b)) return false;  // degenerate geometry: serial path
  PG_STAMP(fc.diag, 11);
  for (int done = 0; done < frames; done += REV_T_CAP) {
    const int T = frames - done < REV_T_CAP ? frames - done : REV_T_CAP;
    float* s0 = sig + 2 * done;
    rev_front(r, s0, T, m, b, fc.diag);
    rev_mid<ITEMS>(r, T, m, b, fc.ctl, fc.diag, fc.idx_log ? fc.idx_log + (size_t)done * 16 : nullptr);
    rev_tail(r, s0, T, m, b, fc.diag);
  }
  return true;
}
