// ringstream — the HBM access stream of the headline kernel's mid stage (pg_reverb_fast.inl: rev_mid) WITHOUT its arithmetic: what does the
// memory system of an MI355X deliver for 1024 workgroups x 12 f64 [frame][2] rings each, read in windows of G frames and rewritten a few
// microseconds later? ("or a counter-backed paragraph showing the 2 KB-granule limit is the memory system's (e.g. the same
// access stream from a parity-free micro-kernel)".) Not part of the product: no audio comes out of this.
//
//   ./ringstream <voices> <blocks> <variant> [wg_per_cu]
//     variant 0: the product's pattern — lane = (frame, channel), 8-byte accesses, sub-chunks of 128 frames, 16 line taps + 4 allpass reads,
//                barrier, 12 writes
//     variant 1: the same with sub-chunks of 256 frames (two items per lane: twice the loads in flight, 4 KB granules)
//     variant 2: lane = frame, 16-byte accesses: per line 4 loads of a whole frame (the two channels read DIFFERENT positions: vibrato phases
//                differ), 12 x 16-byte writes, sub-chunks of 256 frames
//     variant 3: windows through LDS: every ring's window of the sub-chunk (128 + 16 frames) is fetched with linear 16-byte-per-lane loads into
//                LDS, taps are read from LDS, writes as in variant 2 (16 bytes per lane, 128 lanes busy)
//     variant 4: read-only of variant 0 (no writes)      variant 5: write-only of variant 0 (no reads)
//     variant 6: rings laid out [channel][position] (two planes per ring, one guard element behind each plane mirroring position 0): lane = (frame,
//                channel) as in variant 0, but the two taps of a line are ONE 16-byte load (adjacent positions of the lane's channel plane), the allpass
//                reads and all writes stay 8 bytes per lane (a lane writing position 0 also writes the guard) — 12 load instructions per item instead of 20
// Reported: algorithmic bytes (12 rings x 32 B per voice-frame, as DESIGN.md counts them) / kernel time.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) double gdouble;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) d2v gdouble2;

struct Voice { double* ring[12]; uint32_t m[12]; uint32_t pos[12]; };
__constant__ int c_off[12][2];

__device__ __forceinline__ uint32_t wrap(uint32_t v, uint32_t m) { const uint32_t w = v - m; return v < w ? v : w; }
__device__ __forceinline__ int tid_() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }

extern __shared__ __attribute__((aligned(16))) char smem[];

template <int VARIANT>
__global__ void __launch_bounds__(256, 4) ring_kernel(Voice* voices, int n_blocks, int frames) {
  Voice& V = voices[blockIdx.x];
  double* keep = (double*)smem;  // the LDS allocation sets the occupancy; variant 3 stages windows here
  uint32_t m[12], pos[12];
  double* base[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { m[i] = __builtin_amdgcn_readfirstlane(V.m[i]); pos[i] = __builtin_amdgcn_readfirstlane(V.pos[i]);
    unsigned long long b = (unsigned long long)V.ring[i]; base[i] = (double*)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b)); }
  double acc = 0.0;
  for (int blk = 0; blk < n_blocks; ++blk) {
    constexpr int G = (VARIANT == 1 || VARIANT == 2) ? 256 : 128;
    for (int s0 = 0; s0 < frames; s0 += G) {
      if (VARIANT == 0 || VARIANT == 4 || VARIANT == 5 || VARIANT == 1) {
        constexpr int ITEMS = VARIANT == 1 ? 2 : 1;
        double val[ITEMS][20];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const int t = tid_();
          const int n = s0 + (t >> 1) + it * 128, ch = t & 1;
          if (VARIANT != 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const uint32_t p1 = wrap(wrap(pos[i] + (uint32_t)n, m[i]) + (uint32_t)c_off[i][ch], m[i]);
              const uint32_t p2 = wrap(p1 + 1, m[i]);
              val[it][2 * i] = ((const gdouble*)base[i])[p1 * 2 + ch];
              val[it][2 * i + 1] = ((const gdouble*)base[i])[p2 * 2 + ch];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) val[it][16 + i] = ((const gdouble*)base[8 + i])[wrap(pos[8 + i] + (uint32_t)n + 1, m[8 + i]) * 2 + ch];
          } else {
#pragma unroll
            for (int i = 0; i < 20; ++i) val[it][i] = (double)(n + i);
          }
        }
        double sum[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { sum[it] = 0.0;
#pragma unroll
          for (int i = 0; i < 20; ++i) sum[it] += val[it][i]; }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const int t = tid_();
          const int n = s0 + (t >> 1) + it * 128, ch = t & 1;
          if (VARIANT != 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) ((gdouble*)base[i])[wrap(pos[i] + (uint32_t)n, m[i]) * 2 + ch] = sum[it] * 0.03 + (double)i;
          } else acc += sum[it];
        }
      } else if (VARIANT == 6) {
        const int t = tid_();
        const int n = s0 + (t >> 1), ch = t & 1;
        d2v tap[8]; double ap[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t plane = (m[i] + 1u) * (uint32_t)ch;                                   // plane of the lane's channel (m + 1 entries: the guard)
          const uint32_t p1 = wrap(wrap(pos[i] + (uint32_t)n, m[i]) + (uint32_t)c_off[i][ch], m[i]);
          tap[i] = *(const gdouble2*)((const gdouble*)base[i] + plane + p1);                   // positions p1 and p1 + 1 (the guard stands in for the wrap), 8-byte aligned
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = ((const gdouble*)base[8 + i])[(m[8 + i] + 1u) * (uint32_t)ch + wrap(pos[8 + i] + (uint32_t)n + 1, m[8 + i])];
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += tap[i].x + tap[i].y;
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += ap[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const uint32_t plane = (m[i] + 1u) * (uint32_t)ch, p = wrap(pos[i] + (uint32_t)n, m[i]);
          const double v = sum * 0.03 + (double)i;
          ((gdouble*)base[i])[plane + p] = v;
          if (p == 0) ((gdouble*)base[i])[plane + m[i]] = v;   // the guard mirrors position 0
        }
      } else if (VARIANT == 2) {
        const int t = tid_();
        const int n = s0 + t;
        d2v v[8][4], a[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const uint32_t p1 = wrap(wrap(pos[i] + (uint32_t)n, m[i]) + (uint32_t)c_off[i][ch], m[i]);
            v[i][2 * ch] = ((const gdouble2*)base[i])[p1];
            v[i][2 * ch + 1] = ((const gdouble2*)base[i])[wrap(p1 + 1, m[i])];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ((const gdouble2*)base[8 + i])[wrap(pos[8 + i] + (uint32_t)n + 1, m[8 + i])];
        double s0_ = 0.0, s1_ = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0_ += v[i][0].x + v[i][1].x; s1_ += v[i][2].y + v[i][3].y; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0_ += a[i].x; s1_ += a[i].y; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 12; ++i) ((gdouble2*)base[i])[wrap(pos[i] + (uint32_t)n, m[i])] = d2v{s0_ * 0.03 + i, s1_ * 0.03 + i};
      } else if (VARIANT == 3) {
        // windows -> LDS: ring i's frames [pos + s0, pos + s0 + 128 + 16) (allpasses: + 1 .. + 129), 16 bytes per lane, linear
        constexpr int WIN = 144;
        d2v* win = (d2v*)keep;  // [12][WIN]
        const int t = tid_();
        for (int e = t; e < 12 * WIN; e += 256) {
          const int i = e / WIN, k = e - i * WIN;
          win[e] = ((const gdouble2*)base[i])[wrap(wrap(pos[i] + (uint32_t)s0, m[i]) + (uint32_t)k, m[i])];
        }
        __syncthreads();
        const int n_l = t >> 1, ch = t & 1;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const double* w = (const double*)(win + i * WIN + n_l + c_off[i][ch]); sum += w[ch] + w[2 + ch]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double* w = (const double*)(win + (8 + i) * WIN + n_l + 1); sum += w[ch]; }
        const double other = __shfl_xor(sum, 1, 64);
        __syncthreads();
        if (ch == 0) {
#pragma unroll
          for (int i = 0; i < 12; ++i) ((gdouble2*)base[i])[wrap(pos[i] + (uint32_t)(s0 + n_l), m[i])] = d2v{sum * 0.03 + i, other * 0.03 + i};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) pos[i] = wrap(pos[i] + (uint32_t)(frames % m[i]), m[i]);
  }
  if (acc == 12345.678) keep[0] = acc;
  if (threadIdx.x == 0) for (int i = 0; i < 12; ++i) V.pos[i] = pos[i];
}

int main(int argc, char** argv) {
  const int voices = argc > 1 ? atoi(argv[1]) : 1024, blocks = argc > 2 ? atoi(argv[2]) : 16, variant = argc > 3 ? atoi(argv[3]) : 0;
  const int wg_per_cu = argc > 4 ? atoi(argv[4]) : 4;
  const int in_phase = argc > 5 ? atoi(argv[5]) : 0;     // 1: every voice at the same ring positions (voices started together, as in bench.py)
  const int pad = argc > 6 ? atoi(argv[6]) : 0;           // bytes by which voice v's rings are shifted inside its allocation: v * pad (de-phases the physical addresses)
  const int frames = 1024;
  // ring lengths of the reverb at its default room size 0.6: size = 52; lines floor(k * 52) + 1, k = 79 73 71 67 61 59 53 47; allpasses 43 41 37 31
  const uint32_t len[12] = {4109, 3797, 3693, 3485, 3173, 3069, 2757, 2445, 2237, 2133, 1925, 1613};
  const size_t alloc[12] = {8112, 7512, 7312, 6912, 6312, 6112, 5512, 4912, 4511, 4311, 3911, 3311};  // the product allocates the maximum sizes
  size_t per_voice = 0;
  for (int i = 0; i < 12; ++i) per_voice += alloc[i] * 2;
  per_voice += 4096 * 2;  // the predelay ring sits in the same allocation
  std::vector<Voice> h(voices);
  for (int v = 0; v < voices; ++v) {
    double* p = nullptr;
    const size_t shift = ((size_t)v * (size_t)pad) % 65536;
    CHECK(hipMalloc((void**)&p, per_voice * 8 + 65536));   // one allocation per effect instance, as in the product
    CHECK(hipMemset(p, 0, per_voice * 8 + 65536));
    p = (double*)((char*)p + shift);
    for (int i = 0; i < 12; ++i) { h[v].ring[i] = p; p += alloc[i] * 2; h[v].m[i] = len[i]; h[v].pos[i] = in_phase ? 1u : (uint32_t)((v * 131 + i * 977) % len[i]); }
  }
  int off[12][2];
  for (int i = 0; i < 12; ++i) for (int c = 0; c < 2; ++c) off[i][c] = i < 8 ? (i * 5 + c * 9) % 15 : 0;
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_off), off, sizeof off));
  Voice* d = nullptr;
  CHECK(hipMalloc((void**)&d, voices * sizeof(Voice)));
  CHECK(hipMemcpy(d, h.data(), voices * sizeof(Voice), hipMemcpyHostToDevice));
  const size_t lds = (size_t)(160 * 1024 / wg_per_cu) - 1024;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto launch = [&]() {
    switch (variant) {
      case 0: hipLaunchKernelGGL(ring_kernel<0>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
      case 1: hipLaunchKernelGGL(ring_kernel<1>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
      case 2: hipLaunchKernelGGL(ring_kernel<2>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
      case 3: hipLaunchKernelGGL(ring_kernel<3>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
      case 4: hipLaunchKernelGGL(ring_kernel<4>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
      case 6: hipLaunchKernelGGL(ring_kernel<6>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
      default: hipLaunchKernelGGL(ring_kernel<5>, dim3(voices), dim3(256), lds, 0, d, blocks, frames); break;
    }
  };
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)ring_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int w = 0; w < 3; ++w) launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  const int reps = 7;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  CHECK(hipGetLastError());
  const double rw = variant == 4 || variant == 5 ? 0.5 : 1.0;
  const double bytes = 12.0 * 32.0 * rw * (double)voices * frames * blocks;
  printf("{\"in_phase\": %d, \"pad\": %d, \"variant\": %d, \"voices\": %d, \"blocks_per_launch\": %d, \"wg_per_cu\": %d, \"ms_per_block_avg\": %.5f, \"ms_per_block_best\": %.5f, \"algorithmic_GBps_avg\": %.1f, \"algorithmic_GBps_best\": %.1f}\n",
         in_phase, pad, variant, voices, blocks, wg_per_cu, sum / reps / blocks, best / blocks, bytes / (sum / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
  return 0;
}
