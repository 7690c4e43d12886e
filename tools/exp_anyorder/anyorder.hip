// (box) Does a kernel launched with hipExtAnyOrderLaunch start while its predecessor ON THE SAME STREAM still has workgroups running, and are
// the predecessor's workgroups all dispatched before any of its own? The question behind pipelined one-block calls (DESIGN §4 "Calls in
// flight"): 1024 workgroups of 256 lanes with 39 KB of LDS fill the chip's 1024 slots; workgroup i of every launch works for 40 + 60 * (i / 256) / 4
// microseconds (the staircase a single-block launch of the staged kernel shows); two launches back to back on one stream, with and without the flag.
// Prints per launch: first start, last start, first end, last end (microseconds from the first start of launch 0), and whether any workgroup
// of launch 1 started before a workgroup of launch 0 had started.
//   hipcc -O2 --offload-arch=gfx950 -o anyorder.bin anyorder.hip && ./anyorder.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256, 4) work(unsigned long long* stamps, int launch, int n_wg, const unsigned* prev_done, unsigned* done, int wait_prev) {
  extern __shared__ char lds[];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    stamps[(size_t)(launch * n_wg + b) * 2] = __builtin_amdgcn_s_memrealtime();
    lds[0] = 1;
    // per-unit order: wait until the SAME workgroup index of the launch before has finished (bounded)
    if (wait_prev) {
      unsigned polls = 0;
      while (__hip_atomic_load(&prev_done[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++polls < (1u << 22)) __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long dur = (unsigned long long)(4000 + 1500 * (b / 256));   // 100 MHz ticks: 40 / 55 / 70 / 85 us
    while (__builtin_amdgcn_s_memrealtime() - t0 < dur) __builtin_amdgcn_s_sleep(16);
    __hip_atomic_store(&done[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    stamps[(size_t)(launch * n_wg + b) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

int main() {
  const int n_wg = 1024, lds = 39 * 1024;
  unsigned long long* d_st; unsigned* d_done;
  hipMalloc(&d_st, sizeof(unsigned long long) * 2 * 2 * n_wg);
  hipMalloc(&d_done, sizeof(unsigned) * 3 * n_wg);
  hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  std::vector<unsigned long long> h(2 * 2 * n_wg);
  for (int mode = 0; mode < 3; ++mode) {   // 0: ordered, 1: any-order without the per-index wait, 2: any-order with it
    hipMemset(d_st, 0, sizeof(unsigned long long) * 4 * n_wg);
    hipMemset(d_done, 0, sizeof(unsigned) * 3 * n_wg);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {   // the first repetition warms up
      hipMemsetAsync(d_done, 0, sizeof(unsigned) * 3 * n_wg, s);
      hipExtLaunchKernelGGL(work, dim3(n_wg), dim3(256), lds, s, nullptr, nullptr, 0, d_st, 0, n_wg, d_done + 2 * n_wg, d_done, 0);
      hipExtLaunchKernelGGL(work, dim3(n_wg), dim3(256), lds, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, d_st, 1, n_wg, d_done, d_done + n_wg, mode == 2 ? 1 : 0);
      hipStreamSynchronize(s);
    }
    hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 4 * n_wg, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < n_wg; ++i) t0 = std::min(t0, h[2 * i]);
    printf("mode %d (%s):\n", mode, mode == 0 ? "ordered" : mode == 1 ? "any-order" : "any-order + wait for the same index of the launch before");
    unsigned long long last_start0 = 0;
    for (int l = 0; l < 2; ++l) {
      unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
      for (int i = 0; i < n_wg; ++i) {
        const unsigned long long a = h[2 * (l * n_wg + i)], b = h[2 * (l * n_wg + i) + 1];
        s0 = std::min(s0, a); s1 = std::max(s1, a); e0 = std::min(e0, b); e1 = std::max(e1, b);
      }
      if (l == 0) last_start0 = s1;
      printf("  launch %d: starts %.1f .. %.1f us, ends %.1f .. %.1f us\n", l, (s0 - t0) / 100.0, (s1 - t0) / 100.0, (e0 - t0) / 100.0, (e1 - t0) / 100.0);
      if (l == 1) printf("  launch 1's first start %s launch 0's last start (in-order dispatch %s)\n", s0 >= last_start0 ? "is behind" : "is BEFORE", s0 >= last_start0 ? "holds" : "VIOLATED");
    }
  }
  return 0;
}
