// (box / host) The smoothers' serial walk, pg_dsp_dev.h sm_sequence, against a loop of sm_next (the reference's statement order,
// smoothing.rs:21-28): bit-equal sequences and end states for the three kinds over random states, targets, lengths and ramp ends — on the host
// (always) and on one lane of the GPU (when there is one), and what one value costs there in shader clocks.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I../../phonic_amd/csrc -I../../include -o smooth.bin smooth.hip && ./smooth.bin [--host-only]
#include "pg_dsp_dev.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace pgd;

template <int MODE>
__global__ void walk(PgSmooth* g, float* out, int n, unsigned long long* cyc) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    PgSmooth s = g[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) { for (int i = 0; i < n; ++i) lds[i] = sm_next(s); } else sm_sequence(s, lds, n);
    __builtin_amdgcn_s_waitcnt(0);
    *cyc = __builtin_amdgcn_s_memtime() - t0;
    g[1] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = lds[i];
}

static PgSmooth make(std::mt19937& r, int kind) {
  std::uniform_real_distribution<float> u(0.0f, 1.0f);
  PgSmooth s{};
  s.kind = kind;
  s.comp = 44100.0f / (u(r) < 0.5f ? 48000.0f : 44100.0f);
  s.current = u(r) * 2.0f - 1.0f;
  s.target = u(r) < 0.1f ? s.current : u(r) * 2.0f - 1.0f;
  if (kind == SM_EXP) s.a = u(r) < 0.5f ? 0.002f + 0.05f * u(r) : 0.2f * u(r);
  else if (kind == SM_LIN) {
    s.a = 0.0005f + 0.01f * u(r);
    sm_set_target(s, s.target);
  } else { s.a = 0.005f + 0.1f * u(r); s.b = u(r) < 0.5f ? 0.0f : 0.01f * (u(r) - 0.5f); }
  return s;
}

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && !strcmp(argv[1], "--host-only");
  std::mt19937 r(7);
  int bad = 0, cases = 0, ended = 0;
  for (int it = 0; it < 30000; ++it) {
    const int kind = it % 3;
    PgSmooth a = make(r, kind), b = a;
    const int n = 1 + (int)(r() % 2100);
    std::vector<float> x(n), y(n);
    for (int i = 0; i < n; ++i) x[i] = sm_next(a);
    sm_sequence(b, y.data(), n);
    ++cases;
    if (!sm_need_ramp(a)) ++ended;
    if (memcmp(x.data(), y.data(), n * 4) || memcmp(&a, &b, sizeof a)) { if (++bad < 5) printf("host: kind %d n %d DIFFERS\n", kind, n); }
  }
  printf("host: %d cases (%d with the ramp ending inside the call), %d differ\n", cases, ended, bad);
  if (host_only) return bad != 0;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) { printf("no device\n"); return bad != 0; }
  const int n = 2048;
  PgSmooth* d; float* o; unsigned long long* c;
  if (hipMalloc(&d, 2 * sizeof(PgSmooth)) != hipSuccess || hipMalloc(&o, n * 4) != hipSuccess || hipMalloc(&c, 16) != hipSuccess) return 2;
  for (int kind = 0; kind < 3; ++kind) for (int v = 0; v < 3; ++v) {
    PgSmooth s{};
    s.kind = kind; s.comp = 44100.0f / 48000.0f; s.current = 0.0f; s.target = 1.0f;
    // v = 0: the ramp outlasts the call; 1: ends inside; 2: at rest from the start
    if (kind == SM_EXP) s.a = v == 0 ? 0.002f : 0.02f;
    else if (kind == SM_LIN) { s.a = v == 0 ? 0.0002f : 0.002f; sm_set_target(s, 1.0f); }
    else s.a = v == 0 ? 0.002f : 0.05f;
    if (v == 2) { s.current = s.target; s.pending = 0; s.b = kind == SM_SPRING ? 0.0f : s.b; }
    std::vector<float> seq[2]; PgSmooth end[2]; unsigned long long cy[2];
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemcpy(d, &s, sizeof s, hipMemcpyHostToDevice);
        if (mode == 0) hipLaunchKernelGGL(walk<0>, dim3(1), dim3(256), n * 4, 0, d, o, n, c);
        else hipLaunchKernelGGL(walk<1>, dim3(1), dim3(256), n * 4, 0, d, o, n, c);
        (void)hipDeviceSynchronize();
      }
      seq[mode].resize(n);
      (void)hipMemcpy(&cy[mode], c, 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(seq[mode].data(), o, n * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(&end[mode], d + 1, sizeof s, hipMemcpyDeviceToHost);
    }
    const bool same = !memcmp(seq[0].data(), seq[1].data(), n * 4) && !memcmp(&end[0], &end[1], sizeof s);
    if (!same) ++bad;
    printf("device: kind %d, %s: sm_next loop %.1f clocks per value, sm_sequence %.1f; %s\n", kind, v == 0 ? "ramp outlasts the call" : v == 1 ? "ramp ends inside" : "at rest",
           (double)cy[0] / n, (double)cy[1] / n, same ? "identical" : "DIFFER");
  }
  return bad != 0;
}
