// pg_unit_kernel: the generic (exact serial + time-parallel) unit kernel and the pipelined bus chain (one kernel per translation unit; the launchers are in pg_kernels.hip).
#include "pg_unit_body.inl"

// The main mixer's effect chain behind a super-block launch as a PIPELINE over the blocks: workgroup f runs effect f of the chain over the
// summed blocks 0, 1, 2 ... in order and hands block c on to workgroup f + 1 through the bus buffer itself (in place) and a progress word —
// effect f works on block c while effect f + 1 works on block c - 1. A lone workgroup running the whole chain is a pure latency chain (C2: Eq5
// 50 K + Reverb 96 K cycles per block, tools/diag_bus.py); the chain's stages are independent state machines, so its time per block becomes
// the slowest effect's instead of their sum. Per block every effect takes EffectProcessor::process's decisions with the same inputs as in the
// serial order (mixed.rs:627-655): audible_input of the block (L.bus_audible[c]) and whether an earlier effect of the chain was active on
// it (travels with the progress word); MixedSource's shortcut `effects_bypassed && input_bypassed -> skip the chain` changes nothing an
// effect would not decide for itself (a bypassed processor with silent input stays bypassed, effect.rs:88-101) and is kept as state only.
// Workgroup f waits for workgroup f - 1 only. Residency: the launch has at most PG_BUS_PIPELINE_MAX (16) workgroups of 256 lanes, one per CU at
// most, on a device with 256 CUs. Nothing is in front of them on THIS stream, but the next launch sequence's unit kernels may run beside them
// on the graph's unit stream (pg_host.hip: the bus-overlap path, up to 512 workgroups competing for CUs): a consumer stage can then be
// resident and polling while its producer has not been started yet. That is a delay, not a deadlock — the unit kernels need nothing from the
// bus chain and finish in well under a millisecond, after which every stage fits — and nothing here relies on the order the dispatcher
// starts workgroups in (the guide lists it as undefined). Should a producer never publish (a fault, a preempted queue), the consumer gives
// up after 2^24 polls of >= 0.2 us each (seconds: thousands of times the longest unit-kernel sequence it could be waiting behind), raises
// PG_DEVERR_BUS_STALLED and passes its blocks on unprocessed: a stuck stream would be invisible to the host, a raised flag disables the
// graph at the next write.
template <int KMASK>
__device__ __forceinline__ void pg_bus_pipeline(const PgLaunch& L) {
  const int f = (int)blockIdx.x, n_stages = (int)gridDim.x;
  const bool last_stage = f == n_stages - 1;
  PgUnit& unit = L.units[L.unit_base];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  const int NA = N < PG_MIN_ROW_FRAMES ? PG_MIN_ROW_FRAMES : N;
  float* sig = (float*)pg_smem;
  float* tmp = sig + 2 * NA;
  float* nxt = tmp + 2 * NA;   // the NEXT block's input, on its way global -> LDS while this block is processed (pg_launch_units adds the room in mode 3)
  char* scratch = (char*)(nxt + 2 * NA);
  scratch += (sizeof(PgVoice) + 15) & ~15ull;
  PgFx* lfx = (PgFx*)scratch;                      scratch += (sizeof(PgFx) + 15) & ~15ull;
  int* ctl = (int*)scratch;                        scratch += 128;
  float* red = (float*)scratch;                    scratch += 64;
  FastCtx fc;
  fc.tmp = tmp; fc.tmp_floats = 2 * NA; fc.scratch = scratch; fc.ctl = ctl; fc.red = red; fc.diag = L.diag; fc.err = L.error_word; fc.idx_log = nullptr;   // (diag: stamps of stage 0 in diagnostic builds, tools/diag_stamps.py bus)
  PgFx& gfx = L.fx[L.fx_index[unit.fx_off + f]];
  for (int i = tid; i < (int)(sizeof(PgFx) / 4); i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];  // the effect's state stays in LDS over all blocks
  if (tid == 0) { ctl[5] = 0; ctl[6] = 0; ctl[7] = 0; }   // (in front of the barrier: LDS holds whatever an earlier workgroup left)
  __syncthreads();
  const int n_chunks = L.n_chunks > 1 ? L.n_chunks : 1;
  int any_active = 0;
  int prefetched = -1;   // the block whose input was requested into `nxt` (uniform)
  int stalled = 0;       // the producer never published (uniform: every wave reads it from ctl[7] behind the wait block's own barrier)
  unsigned long long hist = 0, mask = 0;   // `an effect up to this one was active` per block: the last 24 blocks / all (<= 64) blocks of the launch
  for (int c = 0; c < n_chunks; ++c) {
    int active_before = 0;
    int next_ready = (f == 0 && c + 1 < n_chunks) ? 1 : 0;   // stage 0 reads the mixer sum: complete before this launch began
    if (f > 0 && !stalled) {
      if (tid == 0) {
        unsigned long long w;
        unsigned polls = 0;
        bool ok;
        do {
          w = __hip_atomic_load(&L.bus_progress[f - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          ok = (uint32_t)(w >> 32) == L.round && (int)(w & 0xffull) > c;
          if (!ok) __builtin_amdgcn_s_sleep(8);   // (~0.2 us: the producer's block takes tens of microseconds)
        } while (!ok && ++polls < (1u << 24));     // (seconds: far beyond any block time)
        if (!ok) { pg_raise_device_error(L, PG_DEVERR_BUS_STALLED); ctl[7] = 1; }
        // `an earlier effect was active on THIS block`: the producer may be several blocks ahead, so the word carries the flags of its last 24
        // blocks (bit 8 = the latest); further back, the producer's per-block mask word (stored before the count was released)
        const int cnt = ok ? (int)(w & 0xffull) : 0, back = cnt - 1 - c;
        int act = 0;
        if (ok) act = back < 24 ? (int)((w >> (8 + back)) & 1ull)
                                : (int)((__hip_atomic_load(&L.bus_progress[PG_BUS_PIPELINE_MAX + f - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> c) & 1ull);
        ctl[6] = act;
        ctl[5] = cnt;   // blocks the producer has published
      }
      __syncthreads();
      __threadfence();  // the producer's stores of the published blocks are visible (the L1 is invalidated behind the acquire)
      active_before = ctl[6];
      next_ready = (ctl[5] > c + 1 && c + 1 < n_chunks) ? 1 : 0;
      stalled = ctl[7];
    }
    PG_STAMP(L.diag, 60);
    float* blk = L.bus + (size_t)c * 2 * (size_t)N;
    if (prefetched == c) {   // requested while the block before was processed: in LDS by now, or nearly
      lds_dma_wait();
      __syncthreads();
      float* t = sig; sig = nxt; nxt = t;
    } else {
      for (int i = tid; i < 2 * N; i += nt) sig[i] = __builtin_nontemporal_load(blk + i);
      __syncthreads();
    }
    // The next block's input: one dword per lane and trip, global -> LDS directly (no registers; nothing waits for it until the next trip
    // of this loop) — the load at the top of a block was a round trip on a workgroup whose block is a latency chain.
    if (next_ready) {
      const float* nb = blk + 2 * (size_t)N;
      for (int k = 0; k * 256 < 2 * N; ++k) { const int i = tid + k * 256; if (i < 2 * N) lds_dma_dword(nb + i, nxt + k * 256 + (tid & ~63)); }
      prefetched = c + 1;
    }
    PG_STAMP(L.diag, 61);
    // (per chunk of the main mixer: the flag of its summed input sits in the word of its last piece, the processor decides at its first)
    const PgPiece pc = pg_piece(L, c);
    const bool audible_input = L.bus_audible ? (L.bus_audible[pc.c_last] != 0) : true;
    const bool input_bypassed = !audible_input && !active_before;
    const bool is_active = fx_processor_process<false, KMASK>(*lfx, sig, 2 * N, input_bypassed, pc.first, pc.last, L.sample_rate, fc, L.fast, ctl, red);
    __syncthreads();
    PG_STAMP(L.diag, 62);
    if (is_active) for (int i = tid; i < 2 * N; i += nt) blk[i] = sig[i];
    any_active = (active_before || is_active) ? 1 : 0;
    PG_STAMP(L.diag, 63);
    hist = ((hist << 1) | (unsigned long long)any_active) & 0xffffffull;
    mask |= (unsigned long long)any_active << c;
    if (!last_stage) {   // (nobody reads the last stage's words: its stores are complete when the kernel ends)
      __threadfence();
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(&L.bus_progress[PG_BUS_PIPELINE_MAX + f], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ordered before the release below)
        __hip_atomic_store(&L.bus_progress[f], ((unsigned long long)L.round << 32) | (hist << 8) | (unsigned long long)(c + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (prefetched >= n_chunks) lds_dma_wait();   // (never: the last block requests nothing)
  __syncthreads();
  for (int i = tid; i < (int)(sizeof(PgFx) / 4); i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
  if (last_stage && tid == 0) unit.effects_bypassed = any_active ? 0 : 1;  // of the last block, as the serial order leaves it
}

// The generic kernel holds one workgroup per CU (its register footprint): the grid is capped at the CU count and every workgroup
// walks its share of the units, so the launch that finds nothing deferred costs 256 workgroup starts instead of n_units.
__global__ void __launch_bounds__(256) pg_unit_kernel(PgLaunch L) {
  if (L.mode == 3) { pg_bus_pipeline<PG_KMASK_GENERIC>(L); return; }
  if (L.mode == 2 && L.defer_list) {  // deferred units only: the compact list the fast kernels of this round appended to
    PG_STAMP(L.diag, 56);
    const int n = *L.defer_count;
    // the round waits for this kernel's longest unit while the time-parallel kernel beside it has slack: its waves go first on their SIMDs
    // (- 1 % per dynamic round, profiles/r05_ab_dyn_round_overhead.txt)
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0 && pg_tid() == 0) {
      *L.defer_reset = 0;
      // tell the host how many units this round deferred: after a round with none (and no change since) it skips this launch. Word 3: how many
      // of them were deferred for their STATE (a pre-scanned round counts those apart; else all of them): a round whose units were deferred
      // for commands alone — source volume / panning / stop / seek, which leave no smoother of an effect moving — leaves the graph in steady state
      const int n_state = (L.pad_chunks && L.defer_state) ? *L.defer_state : n;
      if (L.defer_state_reset) *L.defer_state_reset = 0;
      if (L.host_feedback) {
        *(volatile unsigned long long*)(L.host_feedback + 3) = ((unsigned long long)L.round << 32) | (unsigned long long)(uint32_t)n_state;
        *(volatile unsigned long long*)L.host_feedback = ((unsigned long long)L.round << 32) | (unsigned long long)(uint32_t)n;
        __threadfence_system();
      }
      // statistics (pg_graph_dynamic_stats): unit-blocks that left the fast kernels, and the generic launches that found any
      if (L.error_word && n > 0) { atomicAdd((unsigned long long*)(L.error_word + 2), (unsigned long long)n); atomicAdd((unsigned long long*)(L.error_word + 4), 1ull); }
    }
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
      PgUnitCarry carry;
      carry.resident = 0; carry.fx_valid = 0;
      pg_unit_body<false, PG_KMASK_GENERIC>(L, L.defer_list[i], 0, carry);
      __syncthreads();
      PG_STAMP(L.diag, 59);
    }
    return;
  }
  // n_chunks > 1 reaches this kernel only as the bus launch behind a super-block (the host renders steady-state graphs that way): the
  // chain runs over the summed blocks one after the other, every per-block decision (audible_input, bypass, tails) taken per block
  const int n_chunks = L.n_chunks > 1 ? L.n_chunks : 1;
  for (int slot = (int)blockIdx.x; slot < L.n_units; slot += (int)gridDim.x) {
    for (int c = 0; c < n_chunks; ++c) {
      PgUnitCarry carry;
      carry.resident = 0; carry.fx_valid = 0;
      pg_unit_body<false, PG_KMASK_GENERIC>(L, slot, c, carry);
      __syncthreads();
    }
  }
}
