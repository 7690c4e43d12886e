// Host-side check of pg_log10f (phonic_amd/csrc/pg_dsp_dev.h): the restatement of the host libm's log10f that the device's level detectors use must
// equal the host libm's log10f bit for bit (the reference calls the platform's; the oracle on this box does). Built and run by
// tests/test_host_models.py (hipcc, host code only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "pg_dev.h"
#include "pg_dsp_dev.h"

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  unsigned long bad = 0, n = 0;
  uint32_t s = 777;
  for (long i = 0; i < 40000000L; ++i) {
    s = s * 1664525u + 1013904223u;
    const uint32_t e = 60 + (s >> 24) % 120, m = (s >> 1) & 0x7fffff;
    const float x = from_bits((e << 23) | m);
    const float a = log10f(x), b = pgd::pg_log10f(x);
    if (bits(a) != bits(b) && ++bad < 5) printf("x=%a libm %a restated %a\n", x, a, b);
    ++n;
  }
  for (uint32_t u = 0x358637bdu; u < 0x358637bdu + 2000000u; ++u) {  // every float from 1e-6 upwards for two million steps (the detectors' floor)
    const float x = from_bits(u);
    if (bits(log10f(x)) != bits(pgd::pg_log10f(x))) ++bad;
    ++n;
  }
  printf("tests %lu bad %lu\n", n, bad);
  return bad ? 1 : 0;
}
