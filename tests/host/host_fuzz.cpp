// Random plans through the C ABI of libphonic_gpu against the stub HIP runtime (hip_stub.cpp), built with -fsanitize=address,undefined:
// host memory safety of graph construction and mutation, event scheduling, command lists, topology rebuilds, chunk / piece walking of
// long writes, host-fed voices, the sharded handle's routing and worker threads, the standalone effect handle. usage: host_fuzz [seeds] [steps]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/phonic_gpu.h"

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  int below(int n) { return n <= 0 ? 0 : (int)(next() % (uint64_t)n); }
  float unit() { return (float)(next() >> 40) / (float)(1 << 24); }
};

static uint32_t random_param(Rng& r, int kind, int* is_enum) {
  const int n = pg_effect_kind_param_count(kind);
  pg_param_desc d;
  memset(&d, 0, sizeof d);
  if (n <= 0 || pg_effect_kind_param(kind, r.below(n), &d) != 0) return 0;
  *is_enum = d.type != 0;
  return d.fourcc;
}

// One handle type behind function pointers: the plain graph and the sharded handle take the same plan
struct Api {
  void* h;
  bool sharded;
  int add_mixer(int parent) { return sharded ? pg_sharded_add_mixer_to((pg_sharded_graph*)h, parent) : pg_graph_add_mixer_to((pg_graph*)h, parent); }
  int add_effect(int m, int k) { return sharded ? pg_sharded_add_effect((pg_sharded_graph*)h, m, k, nullptr) : pg_graph_add_effect((pg_graph*)h, m, k, nullptr); }
  int add_voice(int m, const float* pcm, size_t n, uint32_t ch, uint32_t rate, const pg_voice_options* o) {
    return sharded ? pg_sharded_add_voice((pg_sharded_graph*)h, m, pcm, n, ch, rate, o) : pg_graph_add_voice((pg_graph*)h, m, pcm, n, ch, rate, o);
  }
  int add_stream(int m, uint32_t ch, uint32_t rate, size_t cap, const pg_voice_options* o) {
    return sharded ? pg_sharded_add_stream_voice((pg_sharded_graph*)h, m, ch, rate, cap, o) : pg_graph_add_stream_voice((pg_graph*)h, m, ch, rate, cap, o);
  }
  int feed(int v, const float* f, size_t n) { return sharded ? pg_sharded_feed_voice((pg_sharded_graph*)h, v, f, n) : pg_graph_feed_voice((pg_graph*)h, v, f, n); }
  int end_stream(int v) { return sharded ? pg_sharded_end_stream_voice((pg_sharded_graph*)h, v) : pg_graph_end_stream_voice((pg_graph*)h, v); }
  int remove_mixer(int m) { return sharded ? pg_sharded_remove_mixer((pg_sharded_graph*)h, m) : pg_graph_remove_mixer((pg_graph*)h, m); }
  int remove_effect(int e) { return sharded ? pg_sharded_remove_effect((pg_sharded_graph*)h, e) : pg_graph_remove_effect((pg_graph*)h, e); }
  int move_effect(int e, int m, int mv, int off) { return sharded ? pg_sharded_move_effect((pg_sharded_graph*)h, e, m, mv, off) : pg_graph_move_effect((pg_graph*)h, e, m, mv, off); }
  int param(int e, uint32_t id, float v, int norm, uint64_t t) { return sharded ? pg_sharded_schedule_param((pg_sharded_graph*)h, e, id, v, norm, t) : pg_graph_schedule_param((pg_graph*)h, e, id, v, norm, t); }
  int reset(int e, uint64_t t) { return sharded ? pg_sharded_schedule_reset((pg_sharded_graph*)h, e, t) : pg_graph_schedule_reset((pg_graph*)h, e, t); }
  int volume(int v, float x, uint64_t t) { return sharded ? pg_sharded_set_voice_volume((pg_sharded_graph*)h, v, x, t) : pg_graph_set_voice_volume((pg_graph*)h, v, x, t); }
  int pan(int v, float x, uint64_t t) { return sharded ? pg_sharded_set_voice_panning((pg_sharded_graph*)h, v, x, t) : pg_graph_set_voice_panning((pg_graph*)h, v, x, t); }
  int speed(int v, double x, float g, uint64_t t) { return sharded ? pg_sharded_set_voice_speed((pg_sharded_graph*)h, v, x, g, t) : pg_graph_set_voice_speed((pg_graph*)h, v, x, g, t); }
  int seek(int v, double x, uint64_t t) { return sharded ? pg_sharded_seek_voice((pg_sharded_graph*)h, v, x, t) : pg_graph_seek_voice((pg_graph*)h, v, x, t); }
  int stop(int v, uint64_t t) { return sharded ? pg_sharded_stop_voice((pg_sharded_graph*)h, v, t) : pg_graph_stop_voice((pg_graph*)h, v, t); }
  int remove_voice(int v) { return sharded ? pg_sharded_remove_voice((pg_sharded_graph*)h, v) : pg_graph_remove_voice((pg_graph*)h, v); }
  int stop_all() { return sharded ? pg_sharded_stop_all_voices((pg_sharded_graph*)h) : pg_graph_stop_all_voices((pg_graph*)h); }
  size_t write(float* out, size_t n, uint64_t pos) { return sharded ? pg_sharded_write((pg_sharded_graph*)h, out, n, pos) : pg_graph_write((pg_graph*)h, out, n, pos); }
  int playing(int v) { return sharded ? pg_sharded_is_voice_playing((pg_sharded_graph*)h, v) : pg_graph_is_voice_playing((pg_graph*)h, v); }
  void destroy() { if (sharded) pg_sharded_destroy((pg_sharded_graph*)h); else pg_graph_destroy((pg_graph*)h); }
};

static void run_plan(uint64_t seed, int steps, bool sharded) {
  Rng r(seed);
  static const size_t MF[] = {64, 256, 1000, 1024, 4096, 333};
  const size_t mf = MF[r.below(6)];
  Api a;
  a.sharded = sharded;
  if (sharded) {
    int devs[3] = {0, 1, 2};
    a.h = pg_sharded_create(48000, 2, mf, devs, 1 + r.below(3));
    if (a.h && r.below(2)) pg_sharded_set_max_blocks_per_launch((pg_sharded_graph*)a.h, 1 + r.below(8));
  } else {
    a.h = pg_graph_create(48000, 2, mf, 0);
    if (a.h && r.below(2)) pg_graph_set_max_blocks_per_launch((pg_graph*)a.h, 1 + r.below(16));
    if (a.h && r.below(8) == 0) pg_graph_set_fast_math((pg_graph*)a.h, 0);
    if (a.h && r.below(8) == 0) pg_graph_set_timing_period((pg_graph*)a.h, 1 + r.below(3));
  }
  if (!a.h) { fprintf(stderr, "create failed: %s\n", pg_last_error_message()); exit(2); }
  const size_t cap_frames = sharded ? mf * 8 : 9000;   // (a sharded write holds at most max_blocks x max_frames frames: larger calls are refused)
  std::vector<int> mixers(1, 0), fx, voices, streams;
  std::vector<float> pcm(4096), out(2 * 9000);
  for (size_t i = 0; i < pcm.size(); ++i) pcm[i] = 0.25f * (float)((int)(i % 97) - 48) / 48.0f;
  uint64_t pos = 0;
  for (int s = 0; s < steps; ++s) {
    const int act = r.below(22);
    const uint64_t t = pos + (uint64_t)r.below(12000) - (r.below(4) == 0 ? (uint64_t)r.below((int)(pos < 3000 ? pos + 1 : 3000)) : 0);
    switch (act) {
      case 0: case 1: if (mixers.size() < 12) { int m = a.add_mixer(r.below(3) ? 0 : mixers[r.below((int)mixers.size())]); if (m > 0) mixers.push_back(m); } break;
      case 2: if (mixers.size() > 1 && r.below(3) == 0) { size_t i = 1 + (size_t)r.below((int)mixers.size() - 1); a.remove_mixer(mixers[i]); mixers.erase(mixers.begin() + (long)i); } break;
      case 3: case 4: if (fx.size() < 24) { int e = a.add_effect(mixers[r.below((int)mixers.size())], r.below(10)); if (e >= 0) fx.push_back(e); } break;
      case 5: if (!fx.empty() && r.below(2)) { size_t i = (size_t)r.below((int)fx.size()); a.remove_effect(fx[i]); fx.erase(fx.begin() + (long)i); } break;
      case 6: if (!fx.empty()) a.move_effect(fx[r.below((int)fx.size())], mixers[r.below((int)mixers.size())], r.below(3), r.below(7) - 3); break;
      case 7: case 8: case 9: if (voices.size() < 40) {
        pg_voice_options o;
        pg_voice_options_default(&o);
        o.volume = r.unit(); o.panning = 2.0f * r.unit() - 1.0f; o.start_time = r.below(3) ? 0 : t;
        if (r.below(3) == 0) { o.has_repeat = 1; o.repeat = r.below(2) ? PG_REPEAT_FOREVER : (uint64_t)r.below(4); }
        if (r.below(5) == 0) { o.has_loop_range = 1; o.loop_start = (uint64_t)r.below(500); o.loop_end = o.loop_start + 16 + (uint64_t)r.below(1000); }
        if (r.below(4) == 0) o.speed = 0.25 + 3.5 * r.unit();
        if (r.below(5) == 0) o.source_rate = r.below(2) ? 32000 : 96000;
        if (r.below(6) == 0) o.non_transient = 1;
        const uint32_t ch = 1 + (uint32_t)r.below(2);
        static const uint32_t RATES[] = {8000, 22050, 44100, 48000, 96000, 192000};
        int v = a.add_voice(mixers[r.below((int)mixers.size())], pcm.data(), (size_t)(16 + r.below(1800)), ch, RATES[r.below(6)], &o);
        if (v >= 0) voices.push_back(v);
      } break;
      case 10: if (streams.size() < 6) {
        pg_voice_options o;
        pg_voice_options_default(&o);
        o.start_time = r.below(2) ? 0 : t;
        int v = a.add_stream(mixers[r.below((int)mixers.size())], 1 + (uint32_t)r.below(2), r.below(2) ? 48000 : 44100, 1024 + (size_t)r.below(3000), &o);
        if (v >= 0) { streams.push_back(v); voices.push_back(v); }
      } break;
      case 11: if (!streams.empty()) { int v = streams[r.below((int)streams.size())]; a.feed(v, pcm.data(), (size_t)r.below(1200)); if (r.below(10) == 0) a.end_stream(v); } break;
      case 12: if (!fx.empty()) { int is_enum = 0; const int e = fx[r.below((int)fx.size())]; const uint32_t id = random_param(r, r.below(10), &is_enum); a.param(e, id, r.unit(), r.below(2), t); } break;
      case 13: if (!fx.empty()) a.reset(fx[r.below((int)fx.size())], t); break;
      case 14: if (!voices.empty()) { int v = voices[r.below((int)voices.size())]; if (r.below(2)) a.volume(v, r.unit(), t); else a.pan(v, 2.0f * r.unit() - 1.0f, t); } break;
      case 15: if (!voices.empty()) { int v = voices[r.below((int)voices.size())]; if (r.below(2)) a.speed(v, 0.3 + 3.0 * r.unit(), r.below(2) ? 12.0f : 0.0f, t); else a.seek(v, 0.01 * r.unit(), t); } break;
      case 16: if (!voices.empty()) { int v = voices[r.below((int)voices.size())]; if (r.below(3)) a.stop(v, t); else a.remove_voice(v); (void)a.playing(v); } break;
      case 17: if (r.below(6) == 0) a.stop_all(); break;
      default: {  // a write: any length, host buffer (status feedback and staging spans) — the bulk of the steps
        static const size_t LEN[] = {1, 64, 333, 700, 1024, 2048, 2500, 4096, 4097, 5000, 8192, 9000};
        size_t n = LEN[r.below(12)];
        if (n > cap_frames) n = cap_frames;
        (void)a.write(out.data(), 2 * n, pos);
        pos += n;
      } break;
    }
  }
  a.destroy();
}

static void run_effects(uint64_t seed) {
  Rng r(seed);
  for (int k = 0; k < 10; ++k) {
    pg_effect* e = pg_effect_create(k, nullptr, 0);
    if (!e) continue;
    const size_t ch = (k == 0 || k == 2 || k == 3 || k == 9) ? (size_t)(1 + r.below(6)) : 2;
    if (pg_effect_initialize(e, 44100 + (uint32_t)r.below(2) * 3900, ch, 1 + (size_t)r.below(4096)) == 0) {
      std::vector<float> buf(4096 * 6, 0.1f);
      for (int i = 0; i < 12; ++i) {
        int is_enum = 0;
        const uint32_t id = random_param(r, k, &is_enum);
        (void)pg_effect_set_parameter(e, id, r.unit(), r.below(2));
        if (r.below(4) == 0) (void)pg_effect_message_reset(e);
        (void)pg_effect_process(e, buf.data(), ch * (size_t)r.below(5000), (uint64_t)i * 4096);
      }
      (void)pg_effect_tail(e);
    }
    pg_effect_destroy(e);
  }
}

int main(int argc, char** argv) {
  const int seeds = argc > 1 ? atoi(argv[1]) : 24, steps = argc > 2 ? atoi(argv[2]) : 250;
  for (int s = 0; s < seeds; ++s) {
    run_plan(1000 + (uint64_t)s, steps, false);
    run_plan(5000 + (uint64_t)s, steps, true);
    run_effects(9000 + (uint64_t)s);
  }
  printf("host_fuzz: %d seeds x %d steps on the plain graph, the sharded handle and the effect handle: ok\n", seeds, steps);
  return 0;
}
