// Host-side check of pg_expf_glibc (phonic_amd/csrc/pg_dsp_dev.h): the restatement of the host libm's expf that the VolumeFader's inertia uses
// must equal the host libm's expf bit for bit. Built and run by tests/test_host_models.py (hipcc, host code only).
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
  uint32_t s = 4242;
  for (long i = 0; i < 40000000L; ++i) {   // random arguments of either sign with exponents from 2^-30 to 2^6
    s = s * 1664525u + 1013904223u;
    const uint32_t e = 97 + (s >> 24) % 37, m = (s >> 1) & 0x7fffff, sign = (s & 1u) << 31;
    const float x = from_bits(sign | (e << 23) | m);
    const float a = expf(x), b = pgd::pg_expf_glibc(x);
    if (bits(a) != bits(b) && ++bad < 5) printf("x=%a libm %a restated %a\n", x, a, b);
    ++n;
  }
  for (uint32_t rate = 8000; rate <= 192000; rate += 1) {   // the fader's own arguments: -1 / (rate * seconds / ln 100) for the usual fade lengths
    const float secs[6] = {0.01f, 0.05f, 0.2f, 0.5f, 1.0f, 0.003f};
    for (int k = 0; k < 6; ++k) {
      const float x = -1.0f / ((float)rate * secs[k] / 4.605f);
      if (bits(expf(x)) != bits(pgd::pg_expf_glibc(x))) { if (++bad < 5) printf("rate %u secs %g\n", rate, secs[k]); }
      ++n;
    }
  }
  printf("tests %lu bad %lu\n", n, bad);
  return bad ? 1 : 0;
}
