// Host-side check of f32_phase_advance (phonic_amd/csrc/pg_dsp_dev.h): the closed-form f32 phase accumulation must equal the
// serial loop  { p += d; if (p >= 1) p -= 1; }  bit for bit. Built and run by tests/test_host_models.py (hipcc, host code only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "pg_dev.h"
#include "pg_dsp_dev.h"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  srand(3);
  long bad = 0, tests = 0;
  for (int t = 0; t < 200000; ++t) {
    float d = (t % 5 == 0) ? (float)(rand() / (double)RAND_MAX * 0.3) : (float)pow(10.0, -1.0 - 5.0 * rand() / (double)RAND_MAX);
    float p0 = (t % 7 == 0) ? 0.0f : (float)(rand() / (double)RAND_MAX);
    if (p0 >= 1.0f) p0 = 0.5f;
    const int n = 1 + rand() % 2000;
    float a = p0;
    for (int i = 0; i < n; ++i) { a += d; if (a >= 1.0f) a -= 1.0f; }
    float b = p0;
    pgd::f32_phase_advance(b, d, n);
    ++tests;
    if (bits(a) != bits(b)) { if (++bad < 5) printf("mismatch p0=%.9g d=%.9g n=%d serial=%.9g closed=%.9g\n", p0, d, n, a, b); }
  }
  printf("tests %ld bad %ld\n", tests, bad);
  return bad ? 1 : 0;
}
