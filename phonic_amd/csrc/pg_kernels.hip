// gfx950 kernels of the phonic DSP hot path.
//
//   pg_unit_kernel   one workgroup per work unit — a sub-mixer (its sources + its effect chain), a lone
//                    main-mixer source, the main mixer's bus chain, or a standalone effect. The unit's block
//                    of interleaved stereo f32 frames lives in LDS from the source stage to the end of the
//                    effect chain (no intermediate round trips to HBM); persistent state (delay lines,
//                    filter state, smoothers, resampler history) lives in HBM.
//   pg_mix_kernel_*  the mixer-graph sum (reference add_buffers per source, src/source/mixed.rs:606-608):
//                    a deterministic two-stage tree over units, coalesced over the sample index.
//
// Written for CDNA4 only: 64-wide wavefronts, LDS-resident signal, f64 vector ALU for the filter/delay
// arithmetic the reference does in f64. No MFMA: nothing on this path is a dense contraction.
#include "pg_stage_body.inl"

// the kernels (one translation unit each: pg_k_*.hip)
__global__ void pg_unit_kernel_fast(PgLaunch L);
__global__ void pg_unit_kernel_fast_wide(PgLaunch L);
__global__ void pg_unit_kernel_fast_mid(PgLaunch L);
__global__ void pg_unit_kernel(PgLaunch L);
__global__ void pg_stage1_kernel(PgLaunch L);
__global__ void pg_stage2_kernel(PgLaunch L);
__global__ void pg_stage3_kernel(PgLaunch L);
__global__ void pg_stage_fused_kernel(PgLaunch L);
__global__ void pg_stage_fused_wide_kernel(PgLaunch L);
__global__ void pg_stage_fused_adapt_kernel(PgLaunch L);


// ---- mixer-graph sum -------------------------------------------------------------------------------------
// last float4 of a block with an odd frame count: only the samples that exist (the caller's buffer ends there)
__device__ __forceinline__ void mix_store(float* bus, int s4, int n_samples, const float4& acc) {
  if (s4 * 4 + 4 <= n_samples) { *(float4*)(bus + (size_t)s4 * 4) = acc; return; }
  const float v[4] = {acc.x, acc.y, acc.z, acc.w};
  for (int i = 0; i < 4; ++i) if (s4 * 4 + i < n_samples) bus[(size_t)s4 * 4 + i] = v[i];
}
// Stage 1: partial[g][s] = sum over the units of group g (in unit order) of unit_out[u][s].
// Stage 2: bus[s] = sum over groups (in order) of partial[g][s]; audible = OR over units.
// Lanes run over the sample index s (coalesced float4); the f32 sum order is fixed (deterministic).
__global__ void __launch_bounds__(64) pg_mix_kernel_1(const float* __restrict__ unit_out, uint32_t stride, int n_units, int group, float* __restrict__ partial,
                                                      int n_vec4) {
  int s4 = blockIdx.x * blockDim.x + pg_tid();
  int g = blockIdx.y;
  if (s4 >= n_vec4) return;
  int u0 = g * group, u1 = u0 + group;
  if (u1 > n_units) u1 = n_units;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int u = u0; u < u1; ++u) {
    float4 v = *(const float4*)(unit_out + (size_t)u * stride + (size_t)s4 * 4);
    acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
  }
  *(float4*)(partial + (size_t)g * stride + (size_t)s4 * 4) = acc;
}
__global__ void __launch_bounds__(64) pg_mix_kernel_2(const float* __restrict__ partial, uint32_t stride, int n_groups, float* __restrict__ bus, int n_samples,
                                                      const int32_t* __restrict__ audible_row, int n_units,
                                                      int* __restrict__ audible_out) {
  int s4 = blockIdx.x * blockDim.x + pg_tid();
  if (blockIdx.x == gridDim.x - 1 && audible_out) {  // the extra last block: OR of the units' audible flags of this block (wave reduction)
    int a = 0;
    for (int u = pg_tid(); u < n_units; u += 64) a |= audible_row[u];
    for (int off = 32; off > 0; off >>= 1) a |= __shfl_xor(a, off, 64);
    if (pg_tid() == 0) *audible_out = a;
    return;
  }
  const int n_vec4 = (n_samples + 3) / 4;
  if (s4 >= n_vec4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int g = 0; g < n_groups; ++g) {
    float4 v = *(const float4*)(partial + (size_t)g * stride + (size_t)s4 * 4);
    acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
  }
  mix_store(bus, s4, n_samples, acc);
}

// Both stages in one launch (the usual case, <= 4096 units): a workgroup of 256 lanes owns PG_MIX_COLS float4 columns; lane
// (sub, col) sums the 16 units of group sub, sub + 64, ... for its column — the same partial sums, in the same order, as
// pg_mix_kernel_1 — into LDS, then the lanes of sub 0 add the groups' partials in order as pg_mix_kernel_2 does. Bit-identical to
// the two-launch path; one launch gap and one trip of the partials through HBM less.
#define PG_MIX_MAX_GROUPS 256
#define PG_MIX_COLS 4
__global__ void __launch_bounds__(256) pg_mix_kernel(const float* __restrict__ unit_out, uint32_t stride, int n_units, int n_groups, float* __restrict__ bus,
                                                      int n_samples, const int32_t* __restrict__ audible_tab, size_t audible_stride, int* __restrict__ audible_out,
                                                      size_t chunk_stride) {
  // super-block launches: blockIdx.y = block of the super-block (its unit rows chunk_stride floats further, its bus n_samples further)
  unit_out += (size_t)blockIdx.y * chunk_stride;
  bus += (size_t)blockIdx.y * (size_t)n_samples;
  const int n_vec4 = (n_samples + 3) / 4;
  __shared__ float4 part[PG_MIX_MAX_GROUPS][PG_MIX_COLS];
  const int t = pg_tid();
  if (blockIdx.x == gridDim.x - 1) {  // the extra last block: OR of the units' audible flags of block blockIdx.y -> audible_out[blockIdx.y]
    if (!audible_out) return;
    const int32_t* row = audible_tab + (size_t)blockIdx.y * audible_stride;
    int a = 0;
#pragma unroll 4
    for (int u = t; u < n_units; u += 256) a |= row[u];
    a = __syncthreads_or(a);
    if (t == 0) audible_out[blockIdx.y] = a;
    return;
  }
  const int col = t & (PG_MIX_COLS - 1), sub = t / PG_MIX_COLS;  // 64 sub-groups
  const int s4 = blockIdx.x * PG_MIX_COLS + col;
  if (s4 < n_vec4) {
    for (int g = sub; g < n_groups; g += 64) {
      const int u0 = g * 16;
      const int u1 = u0 + 16 < n_units ? u0 + 16 : n_units;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
      for (int u = u0; u < u1; ++u) {
        const float4 v = *(const float4*)(unit_out + (size_t)u * stride + (size_t)s4 * 4);
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
      }
      part[g][col] = acc;
    }
  }
  __syncthreads();
  if (sub == 0 && s4 < n_vec4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < n_groups; ++g) {
      const float4 v = part[g][col];
      acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
    }
    mix_store(bus, s4, n_samples, acc);
  }
}

// ---- host-callable launchers (C++ linkage, used by pg_host.cpp) ------------------------------------------
// LDS arena the time-parallel paths of the effect kinds in `kind_mask` (bit k = pg_effect_kind k) carve up — what a fast-kernel launch
// needs besides the signal rows. Without a Reverb (and without a Compressor) a unit kernel fits three workgroups per CU instead of two.
size_t pg_fast_scratch_bytes(uint32_t kind_mask) {
  size_t need = SRC_SCRATCH_BYTES;
  auto up = [&](size_t b) { if (b > need) need = b; };
  if (kind_mask & (1u << 5)) up(FAST_SCRATCH_BYTES);
  if (kind_mask & (1u << 7)) up(FAST_SCRATCH_COMP_BYTES);
  if (kind_mask & (1u << 6)) up(FAST_SCRATCH_CHORUS_BYTES);
  if (kind_mask & ((1u << 0) | (1u << 2) | (1u << 3) | (1u << 4))) up(FAST_SCRATCH_SCAN_BYTES);
  if (kind_mask & (1u << 8)) up(FAST_SCRATCH_GATE_BYTES);
  return need;
}
size_t pg_unit_lds_bytes(uint32_t n_frames, size_t scratch_bytes) {
  if (n_frames < PG_MIN_ROW_FRAMES) n_frames = PG_MIN_ROW_FRAMES;
  static_assert(sizeof(PgUnit) <= 128, "pg_unit_body keeps a copy of the unit record in 128 bytes of LDS");
  size_t fixed = ((sizeof(PgVoice) + 15) & ~15ull) + 2 * ((sizeof(PgFx) + 15) & ~15ull) + 128 + 64 + 128;
  size_t scratch = pg_fast_scratch_bytes(0xffffffffu);  // the full arena: the largest any effect kind carves up
  if (scratch_bytes && scratch_bytes < scratch) scratch = scratch_bytes < SRC_SCRATCH_BYTES ? SRC_SCRATCH_BYTES : scratch_bytes;
  return (size_t)n_frames * 16 + fixed + ((scratch + 15) & ~15ull);
}
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames, bool wide);
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames) { return pg_stage_lds_bytes(stage, n_frames, false); }
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames, bool wide) {
  const size_t s1 = STAGE_FIXED + STAGE1_UNION + (size_t)n_frames * 16 + ((sizeof(PgVoice) + 15) & ~15ull) + (wide ? PG_STAGE_LEAD * sizeof(PgFx) : 0);  // (wide kernel: + the leading effects' state slots)
  const size_t s2 = STAGE_FIXED + ((FAST_SCRATCH_BYTES + 15) & ~15ull);
  const size_t s3 = STAGE_FIXED + ((STAGE_ARENA_PREFIX + 15) & ~15ull) + (size_t)n_frames * 8;
  if (stage == 1) return s1;
  if (stage == 2) return s2;
  if (stage == 3) return s3;
  const size_t s2r = STAGE_FIXED + STAGE1_UNION + (((size_t)n_frames * 8 + 15) & ~15ull) + REV_TABLES_BYTES;  // stage 0, the single launch: the dry signal stays
  return s1 > s2r ? s1 : s2r;
}
// The staged pipeline of one round (units flagged `staged`): single_launch = pg_stage_fused_kernel, else three launches
// (L.stage_buf must then hold n_units rows).
// The kernels' dynamic-LDS limits are a per-device function attribute: set once for every device the library launches on (graphs of
// different devices, handles used from different threads).
static hipError_t pg_ensure_func_attributes() {
  static std::mutex mtx;
  static bool done[64] = {false};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mtx);
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  const void* fns[] = {(const void*)pg_stage1_kernel, (const void*)pg_stage2_kernel, (const void*)pg_stage3_kernel, (const void*)pg_unit_kernel,
                       (const void*)pg_unit_kernel_fast, (const void*)pg_unit_kernel_fast_wide, (const void*)pg_unit_kernel_fast_mid};
  for (const void* f : fns) if ((e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)pg_stage_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)pg_stage_fused_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)pg_stage_fused_adapt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)) != hipSuccess) return e;
  if (dev >= 0 && dev < 64) done[dev] = true;
  return hipSuccess;
}
// `tail_done`: an event that rides on the LAST launch as its stop event when that launch carries none of its own (the concurrent path's "fast
// kernels done" without a marker packet behind them; single_launch only).
hipError_t pg_launch_stages(const PgLaunch& L, hipStream_t stream, int single_launch, int lean, int wide, int adapt, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t tail_done) {
  if (L.n_units <= 0) return hipSuccess;
  { hipError_t e = pg_ensure_func_attributes(); if (e != hipSuccess) return e; }
  if (single_launch) {
    // ev0/ev1 (only passed when exactly one of the two launches happens): start / stop timestamps taken from the dispatch itself —
    // no marker packets in the stream, which cost ~7 us per round with hipEventRecord
    const int last = adapt ? 2 : wide ? 1 : 0;   // which of the launches below comes last
    auto stop = [&](int which, hipEvent_t own) { return own ? own : (which == last ? tail_done : nullptr); };
    if (lean) hipExtLaunchKernelGGL(pg_stage_fused_kernel, dim3(L.n_units), dim3(256), (uint32_t)pg_stage_lds_bytes(0, L.n_frames), stream, ev0, stop(0, ev1), 0, L);
    if (wide) hipExtLaunchKernelGGL(pg_stage_fused_wide_kernel, dim3(L.n_units), dim3(256), (uint32_t)pg_stage_lds_bytes(0, L.n_frames, true), stream, lean ? nullptr : ev0, stop(1, lean ? nullptr : ev1), 0, L);
    if (adapt) hipExtLaunchKernelGGL(pg_stage_fused_adapt_kernel, dim3(L.n_units), dim3(256), (uint32_t)pg_stage_lds_bytes(0, L.n_frames, true), stream, (lean || wide) ? nullptr : ev0, stop(2, (lean || wide) ? nullptr : ev1), 0, L);
  } else {
    hipLaunchKernelGGL(pg_stage1_kernel, dim3(L.n_units), dim3(256), pg_stage_lds_bytes(1, L.n_frames), stream, L);
    hipLaunchKernelGGL(pg_stage2_kernel, dim3(L.n_units), dim3(256), pg_stage_lds_bytes(2, L.n_frames), stream, L);
    hipLaunchKernelGGL(pg_stage3_kernel, dim3(L.n_units), dim3(256), pg_stage_lds_bytes(3, L.n_frames), stream, L);
  }
  return hipGetLastError();
}
// The deferral decision of a round for every launch slot at once (see launch_level): the same test the fast kernels apply to their own unit
// — static_defer, maybe_ramping, a command addressed to the unit in this launch — left in the unit record (`deferred`, 0 or 1 for EVERY unit of
// the level: the fast kernels of the round read it instead of deciding themselves, and the generic kernel of such a round leaves it alone); the
// deferred slots are appended to the round's list, as a fast kernel would have done, so the generic kernel (mode 2) finds them the usual way.
__global__ void __launch_bounds__(256) pg_defer_scan_kernel(PgLaunch L, PgCmdPack pack) {
  // pack.n > 0: the round's commands arrive as this kernel's argument; block 0 leaves them where L.cmds points (the device ring) for the kernels
  // of the round, which start behind this one
  if (pack.n > 0 && blockIdx.x == 0 && (int)threadIdx.x < pack.n * (int)(sizeof(PgCmd) / 4))
    ((uint32_t*)const_cast<PgCmd*>(L.cmds))[threadIdx.x] = ((const uint32_t*)pack.c)[threadIdx.x];
  const int slot = (int)(blockIdx.x * 256 + threadIdx.x);
  if (slot >= L.n_units) return;
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  PgUnit& unit = L.units[u];
  int ok = !(unit.static_defer || unit.maybe_ramping);
  if (!ok && L.defer_state) atomicAdd(L.defer_state, 1);   // deferred for its state: what decides whether the graph is back in steady state (graph_steady)
  if (pack.n > 0) { for (int ci = 0; ok && ci < pack.n; ++ci) if (pack.c[ci].unit == u) ok = 0; }
  else for (int ci = 0; ok && ci < L.n_cmds; ++ci) if (L.cmds[ci].unit == u) ok = 0;
  unit.deferred = ok ? 0 : 1;
  if (!ok) L.defer_list[atomicAdd(L.defer_count, 1)] = slot;
}
// `done` rides on the launch as its stop event: what the fast kernels' stream waits for, without a marker packet between the scan and the
// generic kernel behind it on the write's stream (a kernel trace of the dynamic rounds showed 14 us between the two).
hipError_t pg_launch_defer_scan(const PgLaunch& L, hipStream_t stream, hipEvent_t done, const PgCmd* h_cmds) {
  if (L.n_units <= 0) { return done ? hipEventRecord(done, stream) : hipSuccess; }
  static_assert(PG_CMD_PACK * (sizeof(PgCmd) / 4) <= 256, "block 0 copies the pack with one dword per lane");
  PgCmdPack pack = {};
  pack.n = 0;
  if (h_cmds && L.n_cmds > 0 && L.n_cmds <= PG_CMD_PACK) { for (int i = 0; i < L.n_cmds; ++i) pack.c[i] = h_cmds[i]; pack.n = L.n_cmds; }
  hipExtLaunchKernelGGL(pg_defer_scan_kernel, dim3((L.n_units + 255) / 256), dim3(256), 0, stream, nullptr, done, 0, L, pack);
  return hipGetLastError();
}
hipError_t pg_launch_units(const PgLaunch& L, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  if (L.n_units <= 0) return hipSuccess;
  size_t lds = pg_unit_lds_bytes(L.n_frames, L.mode == 1 ? L.fast_scratch_bytes : 0);  // (the generic kernel renders any chain: full arena)
  if (L.mode == 3) lds += (size_t)(L.n_frames < PG_MIN_ROW_FRAMES ? PG_MIN_ROW_FRAMES : L.n_frames) * 8;   // pg_bus_pipeline: the next block's input buffer
  { hipError_t e = pg_ensure_func_attributes(); if (e != hipSuccess) return e; }
  if (L.mode == 1 && L.wide == 2) hipExtLaunchKernelGGL(pg_unit_kernel_fast_mid, dim3(L.n_units), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);
  else if (L.mode == 1 && L.wide) hipExtLaunchKernelGGL(pg_unit_kernel_fast_wide, dim3(L.n_units), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);
  else if (L.mode == 1) hipExtLaunchKernelGGL(pg_unit_kernel_fast, dim3(L.n_units), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);
  else hipExtLaunchKernelGGL(pg_unit_kernel, dim3(L.n_units < 256 ? L.n_units : 256), dim3(256), (uint32_t)lds, stream, ev0, ev1, 0, L);  // (mode 3: n_units = stages of the bus pipeline)
  return hipGetLastError();
}
// `done`: an event that rides on the (last) launch as its stop event — the sharded handle's "this shard's partial bus has arrived" without a
// separate hipEventRecord call per shard and write (three HIP calls per shard and call are what the handle's host time is made of).
hipError_t pg_launch_mix(const float* unit_out, uint32_t stride, int n_units, float* partial, float* bus, uint32_t n_samples, const int32_t* audible_tab,
                         size_t audible_stride, int* audible_out, hipStream_t stream, int n_chunks, size_t chunk_stride, hipEvent_t done) {
  int n_vec4 = (int)((n_samples + 3) / 4);
  int group = 16;
  int n_groups = (n_units + group - 1) / group;
  if (n_groups < 1) n_groups = 1;
  if (n_chunks < 1) n_chunks = 1;
  if (n_groups <= PG_MIX_MAX_GROUPS) {
    if (done) hipExtLaunchKernelGGL(pg_mix_kernel, dim3((n_vec4 + PG_MIX_COLS - 1) / PG_MIX_COLS + 1, n_chunks), dim3(256), 0, stream, nullptr, done, 0, unit_out, stride, n_units, n_groups, bus,
                                    (int)n_samples, audible_tab, audible_stride, audible_out, chunk_stride);
    else hipLaunchKernelGGL(pg_mix_kernel, dim3((n_vec4 + PG_MIX_COLS - 1) / PG_MIX_COLS + 1, n_chunks), dim3(256), 0, stream, unit_out, stride, n_units, n_groups, bus, (int)n_samples, audible_tab,
                            audible_stride, audible_out, chunk_stride);
    return hipGetLastError();
  }
  dim3 b(64), g1((n_vec4 + 63) / 64, n_groups), g2((n_vec4 + 63) / 64 + 1);  // +1: the flag-reduction block
  for (int c = 0; c < n_chunks; ++c) {  // (the partials buffer holds one block: the launches of consecutive blocks follow each other in stream order)
    hipLaunchKernelGGL(pg_mix_kernel_1, g1, b, 0, stream, unit_out + (size_t)c * chunk_stride, stride, n_units, group, partial, n_vec4);
    hipLaunchKernelGGL(pg_mix_kernel_2, g2, b, 0, stream, partial, stride, n_groups, bus + (size_t)c * n_samples, (int)n_samples, audible_tab + (size_t)c * audible_stride, n_units, audible_out ? audible_out + c : nullptr);
  }
  if (done) { hipError_t e = hipEventRecord(done, stream); if (e != hipSuccess) return e; }
  return hipGetLastError();
}

