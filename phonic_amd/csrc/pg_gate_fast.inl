// GateEffect (reference src/effect/gate.rs:147-195) with the transcendental work taken off the serial lane: the level in dB
// (log10) and the final dB -> linear conversion (exp) are element-wise and run on all lanes; only the envelope follower, the
// open / hold / closed decision and the gain smoothing — a data-dependent recurrence on three scalars — stay on one lane, over
// values held in LDS.
DEVO bool gate_fast(PgFx& fx, float* sig, int n_samples, FastCtx& fc) {
  PgGate& g = fx.u.gate;
  const int tid = pg_tid(), nt = blockDim.x;
  float* db = (float*)fc.scratch;  // [1024] input level, then the smoothed gate gain in dB
  const int total = n_samples / 2;
  const float threshold = g.threshold, range_db = g.range;
  const uint32_t hold_samples = f2u32(g.hold * (float)fx.sample_rate);
  for (int f0 = 0; f0 < total; f0 += 1024) {
    const int N = total - f0 < 1024 ? total - f0 : 1024;
    float* sp = sig + 2 * f0;
    __syncthreads();
    for (int n = tid; n < N; n += nt) {
      const float frame_peak = fmaxf(fabsf(sp[2 * n]), fabsf(sp[2 * n + 1]));
      db[n] = (frame_peak > 1e-6f) ? 20.0f * pg_log10f(frame_peak) : -120.0f;
    }
    __syncthreads();
    if (tid == 0) {
      float env = g.env_current, gain_db = g.gate_gain_db;
      uint32_t hold_counter = g.hold_counter;
      const float att = g.env_attack, rel = g.env_release, ac = g.attack_coeff, rc = g.release_coeff;
      auto step = [&](float level) -> float {
        const float envelope = env_run(env, att, rel, level);
        float target_gain_db;
        if (envelope >= threshold) { hold_counter = hold_samples; target_gain_db = 0.0f; }
        else if (hold_counter > 0) { hold_counter -= 1; target_gain_db = 0.0f; }
        else target_gain_db = range_db;
        if (target_gain_db > gain_db) gain_db = ac * gain_db + (1.0f - ac) * target_gain_db;
        else gain_db = rc * gain_db + (1.0f - rc) * target_gain_db;
        return gain_db;
      };
      // eight frames per trip through LDS (two 16-byte reads, two 16-byte writes): the recurrence itself is ~15 dependent f32 operations per
      // frame; read and written frame by frame it also paid an LDS round trip per frame (round 5: the Gate was the slowest of the ten effects
      // alone, 0.12 ms per 1024-unit block — profiles/r05_per_effect.jsonl)
      int n = 0;
      for (; n + 8 <= N; n += 8) {
        float4 a = *(const float4*)(db + n), b = *(const float4*)(db + n + 4);
        a.x = step(a.x); a.y = step(a.y); a.z = step(a.z); a.w = step(a.w);
        b.x = step(b.x); b.y = step(b.y); b.z = step(b.z); b.w = step(b.w);
        *(float4*)(db + n) = a; *(float4*)(db + n + 4) = b;
      }
      for (; n < N; ++n) db[n] = step(db[n]);
      g.env_current = env; g.gate_gain_db = gain_db; g.hold_counter = hold_counter;
    }
    __syncthreads();
    for (int s = tid; s < 2 * N; s += nt) {
      const float gd = db[s >> 1];
      const float gain = (gd <= -60.0f) ? 0.0f : db_to_linear(gd);
      sp[s] = sp[s] * gain;
    }
  }
  __syncthreads();
  return true;
}
