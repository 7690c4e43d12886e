/* A plain C99 client of include/phonic_gpu.h: what a cgo / Rust `extern "C"` / any other FFI user sees. Test infrastructure.
 *   c_client describe          no GPU needed: effect descriptors, default voice options, status codes of bad calls
 *   c_client render <n_blocks> one looping stereo voice -> sub-mixer [Gain, Reverb] -> main mixer with an Eq5; prints one line per
 *                              block: "<written> <sum of samples> <sum of |samples|>" (the Python test renders the same graph
 *                              through ctypes and through the oracle) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "phonic_gpu.h"

static int describe(void) {
  int kind, i;
  pg_voice_options o;
  pg_voice_options_default(&o);
  printf("voice_defaults %.3f %.3f %.3f %.3f\n", o.volume, o.panning, o.speed, o.fade_out_seconds);
  for (kind = 0; kind < 10; ++kind) {
    const int n = pg_effect_kind_param_count(kind);
    printf("%d %s weight=%d params=%d:", kind, pg_effect_kind_name(kind), pg_effect_kind_weight(kind), n);
    for (i = 0; i < n; ++i) {
      pg_param_desc d;
      if (pg_effect_kind_param(kind, i, &d) != PG_OK) return 2;
      printf(" %c%c%c%c", (char)(d.fourcc >> 24), (char)(d.fourcc >> 16), (char)(d.fourcc >> 8), (char)d.fourcc);
    }
    printf("\n");
  }
  printf("bad_kind_params %d\n", pg_effect_kind_param_count(99));
  return 0;
}

static int render(int n_blocks) {
  enum { SR = 48000, BLOCK = 1024, SRC_RATE = 44100, SRC_FRAMES = 4410 };
  static float pcm[(SRC_FRAMES + 1) * 2]; /* + the extra zero frame */
  static float out[BLOCK * 2];
  pg_effect_init gain_init, rev_init, eq_init;
  pg_voice_options opt;
  pg_graph* g;
  int m, fx_gain, fx_rev, fx_eq, v, b, i;
  for (i = 0; i < SRC_FRAMES; ++i) {
    pcm[2 * i] = (float)(0.05 * sin(2.0 * 3.14159265358979323846 * 220.0 * i / SRC_RATE));
    pcm[2 * i + 1] = (float)(0.05 * sin(2.0 * 3.14159265358979323846 * 222.2 * i / SRC_RATE + 0.5));
  }
  g = pg_graph_create(SR, 2, BLOCK, 0);
  if (!g) { fprintf(stderr, "create: %s\n", pg_last_error_message()); return 3; }
  m = pg_graph_add_mixer(g);
  memset(&gain_init, 0, sizeof gain_init);
  gain_init.n_params = 1; gain_init.fourcc[0] = PG_FOURCC('g', 'a', 'i', 'n'); gain_init.value[0] = 0.5f;
  fx_gain = pg_graph_add_effect(g, m, 0 /* Gain */, &gain_init);
  memset(&rev_init, 0, sizeof rev_init);
  rev_init.n_params = 1; rev_init.fourcc[0] = PG_FOURCC('r', 'o', 'o', 'm'); rev_init.value[0] = 0.4f;
  rev_init.has_reverb_seeds = 1; rev_init.reverb_fpd_l = 12345u; rev_init.reverb_fpd_r = 54321u;
  for (i = 0; i < 16; ++i) rev_init.reverb_vib_phase[i] = 0.25 * i;
  fx_rev = pg_graph_add_effect(g, m, 5 /* Reverb */, &rev_init);
  memset(&eq_init, 0, sizeof eq_init);
  fx_eq = pg_graph_add_effect(g, PG_MAIN_MIXER, 3 /* Eq5 */, &eq_init);
  pg_voice_options_default(&opt);
  opt.volume = 0.8f; opt.panning = -0.25f; opt.has_repeat = 1; opt.repeat = PG_REPEAT_FOREVER;
  v = pg_graph_add_voice(g, m, pcm, SRC_FRAMES + 1, 2, SRC_RATE, &opt);
  if (m < 0 || fx_gain < 0 || fx_rev < 0 || fx_eq < 0 || v < 0) { fprintf(stderr, "build: %s\n", pg_last_error_message()); return 4; }
  if (pg_graph_schedule_param(g, fx_gain, PG_FOURCC('g', 'a', 'i', 'n'), 0.25f, 0, 2 * BLOCK + 100) != PG_OK) return 5;
  if (pg_graph_schedule_param(g, fx_gain, PG_FOURCC('n', 'o', 'p', 'e'), 0.25f, 0, 0) != PG_ERR_PARAMETER) return 6;
  if (pg_graph_remove_effect(g, 999) != PG_ERR_NOT_FOUND) return 7;
  for (b = 0; b < n_blocks; ++b) {
    double s = 0.0, a = 0.0;
    const size_t w = pg_graph_write(g, out, BLOCK * 2, (uint64_t)b * BLOCK);
    for (i = 0; i < (int)w; ++i) { s += out[i]; a += fabs(out[i]); }
    printf("%zu %.9e %.9e\n", w, s, a);
  }
  pg_graph_destroy(g);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "describe") == 0) return describe();
  if (argc >= 3 && strcmp(argv[1], "render") == 0) return render(atoi(argv[2]));
  fprintf(stderr, "usage: c_client describe | render <n_blocks>\n");
  return 1;
}
