// Shared device code of the staged kernels for [leading effects]* -> Reverb units (included by pg_k_staged*.hip, pg_k_stages.hip and, for the
// LDS plan's constants, by pg_kernels.hip).
#pragma once
#include "pg_unit_body.inl"

// ---- staged pipeline for [Gain|Panning]* -> Reverb units -------------------------------------------------------------------
// The fused fast kernel above is one function at 256 VGPRs / 55 KB LDS: two workgroups per CU, latency-bound. Here the same
// stage functions of the reverb are separate units of register allocation with an LDS plan of 38 KB, so four workgroups fit a CU:
//   stage 1: deferral decision, source stage, leading Gain/Panning effects, reverb bypass logic + front (predelay, biquad A)
//   stage 2: reverb mid (allpasses + vibrato lines)
//   stage 3: reverb tail (biquad B, asin, biquad C, dry mix), tail counters, sub-mixer silence gate, unit output
// Two drivers: pg_stage_fused_kernel runs the three stages back to back in one launch (chunk buffer and effect state stay in
// LDS, the dry signal is parked in the unit's output row during stage 2); pg_stage1/2/3_kernel are one launch per stage with
// the chunk buffer handed over through HBM/L2 (kept for profiling the stages in isolation).
// LDS plan (all stages): [PgFx][ctl 128][red 64][arena ...]; the arena starts with bufA in every stage.
//   stage 1: arena = union(source scratch, bufA .. xchg), then [sig][tmp][PgVoice]                        36.6 KB at 1024 frames
//   stage 2: single launch: bufA .. xchg (inside the union), [sig] where stage 1 left it, then anchors + rotation table   38.9 KB
//            (round 2: the table holds |j| <= 64 only, which makes room for the dry signal: it no longer travels to the unit's output
//            row and back — 16 B per voice-frame less HBM traffic and no reload at the top of stage 3);
//            per-stage launches: the plain reverb arena (bufA, records, anchors, rotation table)           30.4 KB
//   stage 3: single launch: bufA .. xchg, [sig] in place; per-stage launches: bufA .. xchg, then [sig]     27.9 KB
__device__ __forceinline__ int stage_image_doubles(int T) { return 2 * T + (T >> 3) + 2; }
__device__ __forceinline__ void stage_store_image(double* g, const double* lds, int T) {
  const int n2 = (stage_image_doubles(T) + 1) >> 1;
  for (int i = pg_tid(); i < n2; i += blockDim.x) ((double2*)g)[i] = ((const double2*)lds)[i];
}
__device__ __forceinline__ void stage_load_image(double* lds, const double* g, int T) {
  const int n2 = (stage_image_doubles(T) + 1) >> 1;
  for (int i = pg_tid(); i < n2; i += blockDim.x) ((double2*)lds)[i] = ((const double2*)g)[i];
}
constexpr int PG_STAGE_LEAD = 3;  // state blocks of effects in front of the reverb that stage 1 requests ahead (slot_lead)
constexpr size_t STAGE_ARENA_PREFIX = (size_t)REV_BUF_DOUBLES * 8 + 16 * sizeof(RevRec) + 16 * 8 + 13 * sizeof(RevDesc) + 4 * 8;  // bufA .. xchg
constexpr size_t STAGE_FIXED = ((sizeof(PgFx) + 15) & ~15ull) + 128 + 64;
constexpr size_t STAGE1_UNION = ((SRC_SCRATCH_BYTES > STAGE_ARENA_PREFIX ? SRC_SCRATCH_BYTES : STAGE_ARENA_PREFIX) + 15) & ~15ull;

struct StageLds { PgFx* lfx; int* ctl; float* red; char* arena; };
// `base`: the dynamic LDS of the kernel. An out-of-line stage function gets it as an argument: naming pg_smem there makes the
// compiler look the kernel's LDS offset up in a table in global memory — a dependent load at the top of the stage.
__device__ __forceinline__ StageLds stage_lds(char* base = pg_smem) {
  StageLds m;
  char* p = base;
  m.lfx = (PgFx*)p;   p += (sizeof(PgFx) + 15) & ~15ull;
  m.ctl = (int*)p;    p += 128;
  m.red = (float*)p;  p += 64;
  m.arena = p;
  return m;
}
__device__ __forceinline__ PgFx& stage_reverb(const PgLaunch& L, const PgUnit& unit) { return L.fx[L.fx_index[unit.fx_off + unit.n_fx - 1]]; }

// Stage 1. RESIDENT: the later stages run in the same launch (bufA and the effect state stay in LDS). Returns false when the
// unit was deferred to the generic kernel.
// Returns the unit's stage flags (PG_STAGE_*, also left in the unit record for the per-stage launches), or -1 when the unit was
// deferred to the generic kernel.
// Single launch: the later stages of the block take the reverb's block parameters from the effect's LDS copy as this stage left it
// (rev_block_params_cached) instead of asking rev_block_params again — four barriers and a lane-0 trip through LDS each, on a workgroup
// whose block is a latency chain. The sub-mixer's call record (peak and frames of the chunk's earlier pieces) is read here with the unit
// record and waits in ctl[26] / ctl[27] for stage 3: it used to cost that stage a dependent trip through the loaded memory system at its head.
template <int TAG, bool RESIDENT>
__device__ __forceinline__ int stage1_run(const PgLaunch& L, int slot, const int4 si, const int chunk = 0, char* smem = pg_smem) {
  // `si` = L.slot_info[slot], loaded by the kernel: one load names the unit, its first voice and its reverb (and the unit's staged
  // level): their state blocks are then fetched side by side
  const int u = si.x;
  PgUnit& unit = L.units[u];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  float* out = L.unit_out + (size_t)chunk * L.chunk_stride + (size_t)slot * L.out_stride;
  const uint64_t pos0 = L.pos + (uint64_t)chunk * (uint64_t)N;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  PgFx& gfx = L.fx[si.z];
  uint32_t voice_word = 0;
  if ((si.w & 0xffffff) > 0 && tid < (int)(sizeof(PgVoice) / 4)) voice_word = ((const uint32_t*)&L.voices[si.y])[tid];
  unsigned long long fxr_word = 0;  // the reverb's state block (one qword per lane), used after the source stage
  if (tid < n_fx_words / 2) fxr_word = ((const unsigned long long*)&gfx)[tid];
  const StageLds m0 = stage_lds(smem);
  PgFx* lfx = m0.lfx; int* ctl = m0.ctl; float* red = m0.red;
  float* sig = (float*)(m0.arena + STAGE1_UNION);
  float* tmp = sig + 2 * N;
  PgVoice* lv = (PgVoice*)(tmp + 2 * N);
  // The effects in front of the reverb (wide kernel; C5: Filter -> Eq5 -> Delay): their state blocks used to be fetched one after the other,
  // each behind the index table and behind the write-back of the one before — two dependent trips through the loaded memory system per
  // effect on a workgroup whose stage 1 is a latency chain. The host names the first three in the slot table; they travel global -> LDS
  // directly (no registers, nothing waits here) while the source stage runs, into slots behind the voice record (pg_stage_lds_bytes).
  PgFx* const lead = (PgFx*)((char*)lv + ((sizeof(PgVoice) + 15) & ~15ull));
  int lead0 = -1, lead1 = -1, lead2 = -1;
  if (TAG >= 3) {
    const int4 sl = L.slot_lead[slot];
    lead0 = __builtin_amdgcn_readfirstlane(sl.x); lead1 = __builtin_amdgcn_readfirstlane(sl.y); lead2 = __builtin_amdgcn_readfirstlane(sl.z);
#pragma unroll
    for (int k = 0; k < PG_STAGE_LEAD; ++k) {
      const int li = k == 0 ? lead0 : k == 1 ? lead1 : lead2;
      if (li < 0) continue;
      const float* src = (const float*)&L.fx[li];
      float* dst = (float*)(lead + k);
#pragma unroll
      for (int w = 0; w < n_fx_words; w += 256) if (w + tid < n_fx_words) lds_dma_dword(src + w + tid, dst + w + (tid & ~63));
    }
  }
  SrcScratch S;
  src_carve(m0.arena, S);
  S.diag = L.diag;
  S.sched_rd = nullptr;
  FastCtx fc;
  fc.tmp = tmp; fc.tmp_floats = 2 * N; fc.scratch = m0.arena; fc.ctl = ctl; fc.red = red; fc.diag = L.diag; fc.err = L.error_word; fc.idx_log = nullptr;
  PG_STAMP(L.diag, 0);
  // deferral decision: identical to the fused fast kernel
  if (tid == 0) {
    // (L.pad_chunks: a pre-scanned round — pg_defer_scan_kernel decided for every unit and left the decision in `deferred`; the generic kernel
    // runs beside this one and may already be rewriting maybe_ramping of the units it renders)
    int ok = !(unit.static_defer || (L.pad_chunks ? unit.deferred : unit.maybe_ramping));
    for (int ci0 = 0; ok && ci0 < L.n_cmds; ++ci0) if (L.cmds[ci0].unit == u) ok = 0;
    unit.deferred = ok ? 0 : 1;
    if (!ok && L.n_chunks > 1) pg_raise_super_deferred(L);
    else if (!ok && L.defer_list) L.defer_list[atomicAdd(L.defer_count, 1)] = slot;
    ctl[5] = ok;
  }
  __syncthreads();
  if (!ctl[5]) { if (TAG >= 3) lds_dma_wait(); return -1; }   // (nothing may still be on its way into LDS when the workgroup moves on)
  // the unit record is read once: every later `unit.x` would be another dependent trip to L2 on this workgroup's critical path
  const int n_voices = unit.n_voices, voice_off = unit.voice_off, n_fx = unit.n_fx, fx_off = unit.fx_off, effects_bypassed = unit.effects_bypassed;
  const int chunk_audible_input = unit.chunk_audible_input;
  if (RESIDENT && tid == 0) { ctl[26] = __float_as_int(unit.call_max); ctl[27] = (int)unit.call_frames; }   // (written by stage 3 of the piece before; ctl[24..31]: no stage writes them)
  for (int i = tid; i < 2 * N; i += nt) sig[i] = 0.0f;  // clear_buffer (mixed.rs:673)
  __syncthreads();
  // (staged units are sub-mixers of the main mixer without commands: their chunks are the main mixer's)
  const PgPiece pc = pg_piece(L, chunk);
  bool audible_input = false;
  int later = 0;
  for (int vi = 0; vi < n_voices; ++vi) {
    PgVoice* gv = &L.voices[vi == 0 ? si.y : L.voice_index[voice_off + vi]];
    const int r = voice_process<false, TAG == 4 ? 2 : 0>(gv, lv, sig, tmp, N, pos0, S, L.sched, L.sched_bank, vi == 0, voice_word, pc.chunk_end, pc.first, pc.chunk_end);   // (TAG 4, pg_stage_fused_adapt_kernel: also voices behind a ResampledSource and host-fed ones — the host sends their units there, level 3)
    audible_input |= (r & 1) != 0;
    later |= r & 2;
  }
  // audible_input of the chunk: decided at its first piece (sources that wrote here or start in a later piece), kept for the others
  if (pc.first) { audible_input = audible_input || later != 0; if (tid == 0) unit.chunk_audible_input = audible_input ? 1 : 0; }
  else audible_input = chunk_audible_input != 0;
  PG_STAMP(L.diag, 1);
  if (TAG >= 3) lds_dma_wait();   // the leading effects' state blocks (requested in front of the source stage: long there); the loop's first barrier publishes them
  int flags = audible_input ? PG_STAGE_AUDIBLE : 0;
  if (pc.first) flags |= PG_STAGE_FIRST;
  if (pc.last) flags |= PG_STAGE_LAST;
  bool input_bypassed = !audible_input;
  if (effects_bypassed && input_bypassed) flags |= PG_STAGE_SKIPPED;  // process_effects (mixed.rs:627-655)
  else {
    bool all_bypassed = true;
    for (int fi = 0; fi + 1 < n_fx; ++fi) {  // leading effects
      const int li = fi == 0 ? lead0 : fi == 1 ? lead1 : fi == 2 ? lead2 : -1;
      const bool pre = TAG >= 3 && li >= 0;
      PgFx& g1 = L.fx[pre ? li : L.fx_index[fx_off + fi]];
      PgFx* const cfx = pre ? lead + fi : lfx;
      __syncthreads();
      if (!pre) for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)cfx)[i] = ((const uint32_t*)&g1)[i];
      __syncthreads();
      if (fi == 0) PG_STAMP(L.diag, 46);
      bool is_active;
      constexpr int KM = TAG >= 3 ? PG_KMASK_LEADING : PG_KMASK_GAINPAN;
      if (cfx->standalone) {
        __syncthreads();
        if (tid == 0) fx_call_begin(*cfx);
        __syncthreads();
        fx_process_wg<true, KM>(*cfx, sig, N * 2, fc, L.fast, true);
        if (tid == 0) cfx->call_ramp = 0;
        is_active = true;
      }
      else is_active = fx_processor_process<true, KM>(*cfx, sig, N * 2, input_bypassed, pc.first, pc.last, L.sample_rate, fc, L.fast, ctl, red);
      if (is_active) { input_bypassed = false; all_bypassed = false; }
      __syncthreads();
      if (fi == 0) PG_STAMP(L.diag, 47);
      for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&g1)[i] = ((const uint32_t*)cfx)[i];
      PG_STAMP(L.diag, 40 + fi);
    }
    __syncthreads();
    if (tid < n_fx_words / 2) ((unsigned long long*)lfx)[tid] = fxr_word;
    __syncthreads();
    PG_STAMP(L.diag, 8);
    const bool active = lfx->standalone ? true : !fx_processor_pre(*lfx, input_bypassed, pc.first, ctl);
    if (active) {
      flags |= PG_STAGE_ACTIVE;
      const RevLds m = rev_lds(m0.arena);
      RevBlock b;
      (void)rev_block_params(*lfx, m, ctl, b);  // geometry was validated by the eligibility check (reverb_fast_eligible)
      PG_STAMP(L.diag, 11);
      rev_front(lfx->u.reverb, sig, N, m, b, L.diag);
      if (!RESIDENT) stage_store_image(L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, m.bufA, N);
    }
    if (input_bypassed) flags |= PG_STAGE_INPUT_BYPASSED;
    if (all_bypassed) flags |= PG_STAGE_ALL_BYPASSED;
    __syncthreads();
    if (!RESIDENT || !(flags & PG_STAGE_ACTIVE)) for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
  }
  if (!RESIDENT) for (int i = tid; i < 2 * N; i += nt) out[i] = sig[i];  // per-stage launches: the dry signal waits in the unit's output row
  if (!RESIDENT && tid == 0) unit.stage_flags = flags;  // (the single-launch kernels hand the flags over in registers)
  PG_STAMP(L.diag, 14);
  // schedule cache (ratio < 0.5 only): representatives replay the next block's resampler schedule. A single voice that took the
  // time-parallel schedule needs nothing published — decided from its LDS copy, without a trip to the voice table.
  if (L.sched && tid == 0 && !(n_voices == 1 && lv->sched_hit == 2)) {
    const int piece = N < SRC_OUT_CAP ? N : SRC_OUT_CAP;
    for (int vi = 0; vi < n_voices; ++vi) sched_publish(&L.voices[L.voice_index[voice_off + vi]], L.sched, L.sched_bank, piece);
  }
  PG_STAMP(L.diag, 13);
  return flags;
}

template <int TAG, bool RESIDENT>
__device__ __forceinline__ void stage2_run(const PgLaunch& L, int slot, int flags, char* smem = pg_smem) {
  if (!(flags & PG_STAGE_ACTIVE)) return;
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  PgUnit& unit = L.units[u];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  const StageLds m0 = stage_lds(smem);
  PgFx* lfx = m0.lfx;
  const RevLds m = rev_lds(m0.arena, RESIDENT ? m0.arena + STAGE1_UNION + (((size_t)N * 8 + 15) & ~15ull) : nullptr);  // (single launch: behind the dry signal)
  if (!RESIDENT) {
    PgFx& gfx = stage_reverb(L, unit);
    for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];
    stage_load_image(m.bufA, L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, N);
  }
  __syncthreads();
  rev_load_vtab(lfx->u.reverb, m);   // (visible behind the first barrier of rev_mid's chunk set-up)
  RevBlock b;
  if (RESIDENT) rev_block_params_cached(*lfx, b);
  else (void)rev_block_params(*lfx, m, m0.ctl, b);
  rev_mid(lfx->u.reverb, N, m, b, m0.ctl, L.diag);
  __syncthreads();
  if (!RESIDENT) {
    PgFx& gfx = stage_reverb(L, unit);
    stage_store_image(L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, m.bufA, N);
    for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
  }
}

template <int TAG, bool RESIDENT>
__device__ __forceinline__ void stage3_run(const PgLaunch& L, int slot, int flags, char* smem, const int chunk, const int4 si_in) {
  // single launch: the slot's table entry, the block parameters and the sub-mixer's call record come from stage 1 (registers / ctl[26..27]).
  // Per-stage launches: one load names the unit and its reverb (as in stage 1); it is issued ahead of the dry-signal transfer below so that
  // waiting for it does not wait for the transfer (loads return in order)
  const int4 si = RESIDENT ? si_in : L.slot_info[slot];
  PgUnit& unit = L.units[si.x];
  PgFx& gfx = L.fx[si.z];
  const int tid = pg_tid(), nt = blockDim.x;
  const int N = (int)L.n_frames;
  float* out = L.unit_out + (size_t)chunk * L.chunk_stride + (size_t)slot * L.out_stride;
  const int n_fx_words = (int)(sizeof(PgFx) / 4);
  const StageLds m0 = stage_lds(smem);
  PgFx* lfx = m0.lfx; int* ctl = m0.ctl; float* red = m0.red;
  float* sig = (float*)(m0.arena + (RESIDENT ? STAGE1_UNION : ((STAGE_ARENA_PREFIX + 15) & ~15ull)));  // single launch: where stage 1 left it
  // the sub-mixer's call record (peak and frames of the chunk's earlier pieces) is needed behind the tail: requested here, it arrives under the
  // two scans instead of costing the hand-over a trip through the loaded memory system (nothing in this stage writes it before it is used)
  float call_max0 = 0.0f; uint32_t call_frames0 = 0;
  if (!RESIDENT && tid == 0) { call_max0 = unit.call_max; call_frames0 = unit.call_frames; }
  if (!RESIDENT) __syncthreads();
  // Per-stage launches: the dry signal (stage 1 left it in the unit's output row) is only needed at the end of the tail. It travels
  // global -> LDS directly (lds_dma_dword: no registers, no wait here) while the two scans run; a dependent load at this point would
  // cost a full trip through the loaded memory system. Lane l of wave w, trip k: sample k * 256 + w * 64 + l.
  if (!RESIDENT) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      if (i < 2 * N) lds_dma_dword(out + i, sig + k * 256 + (tid & ~63));
    }
  }
  if (!(flags & PG_STAGE_SKIPPED)) {
    bool all_bypassed = (flags & PG_STAGE_ALL_BYPASSED) != 0;
    if (flags & PG_STAGE_ACTIVE) {
      const RevLds m = rev_lds(m0.arena);
      if (!RESIDENT) {
        for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)lfx)[i] = ((const uint32_t*)&gfx)[i];
        stage_load_image(m.bufA, L.stage_buf + (size_t)slot * PG_STAGE_BUF_DOUBLES, N);
      }
      __syncthreads();
      RevBlock b;
      if (RESIDENT) rev_block_params_cached(*lfx, b);
      else (void)rev_block_params(*lfx, m, ctl, b);
      rev_tail_impl<!RESIDENT>(lfx->u.reverb, sig, N, m, b, L.diag);
      PG_STAMP(L.diag, 60);
      if (!lfx->standalone) fx_processor_post(*lfx, sig, N * 2, (flags & PG_STAGE_INPUT_BYPASSED) != 0, (flags & PG_STAGE_LAST) != 0, L.sample_rate, ctl, red);
      PG_STAMP(L.diag, 61);
      all_bypassed = false;
      __syncthreads();
      for (int i = tid; i < n_fx_words; i += nt) ((uint32_t*)&gfx)[i] = ((const uint32_t*)lfx)[i];
    }
    __syncthreads();
    if (tid == 0 && (flags & PG_STAGE_LAST)) unit.effects_bypassed = all_bypassed ? 1 : 0;  // (counts from the next chunk on)
  }
  if (!RESIDENT) lds_dma_wait();  // (bypassed / skipped reverb: the dry signal is the output)
  __syncthreads();
  PG_STAMP(L.diag, 62);
  // ---- hand the block to the parent mixer: staged units are sub-mixers (SubMixerProcessor::process, submixer.rs:47-77), one call per chunk ----
  const bool closes = (flags & PG_STAGE_LAST) != 0;
  if (tid == 0) { ctl[8] = RESIDENT ? ctl[26] : __float_as_int(call_max0); ctl[9] = RESIDENT ? ctl[27] : (int)call_frames0; }
  __syncthreads();
  const bool aud = submixer_call_piece(unit, ctl + 8, sig, out, 0, N, closes, L.sample_rate, (size_t)L.chunk_stride, (int)(L.out_stride / 2), ctl, red);
  if (tid == 0) {
    if (closes) { unit.audible = aud ? 1 : 0; unit.call_audible = aud ? 1ull : 0ull; }
    if (L.audible_tab) L.audible_tab[(size_t)chunk * L.audible_stride + slot] = (closes && aud) ? 1 : 0;
  }
  PG_STAMP(L.diag, 15);
}

#ifndef PG_STAGE_WAVES
#define PG_STAGE_WAVES 4
#endif
// unit.staged: 0 no, 1 = leading effects are Gain / Panning only, 2 = any kind of PG_KMASK_LEADING, 3 = as 2 with a voice behind a ResampledSource or a
// host-fed one (the source stage with the adapters: a kernel of its own so that the other two keep their register allocation). A launch renders the levels
// up to L.staged_on; LEVEL selects which of them this kernel takes.
template <int LEVEL>
__device__ __forceinline__ bool stage_unit_staged(const PgLaunch& L, int slot) {
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  return L.units[u].staged == LEVEL;
}
// the same from the slot-info word (the single-launch kernels start from that one load)
__device__ __forceinline__ int4 stage_slot_info(const PgLaunch& L, int slot) {
  int4 si = L.slot_info[slot];
  si.x = __builtin_amdgcn_readfirstlane(si.x); si.y = __builtin_amdgcn_readfirstlane(si.y);
  si.z = __builtin_amdgcn_readfirstlane(si.z); si.w = __builtin_amdgcn_readfirstlane(si.w);
  return si;
}
__device__ __forceinline__ int stage_unit_flags(const PgLaunch& L, int slot, bool& deferred) {
  const int u = L.unit_order ? L.unit_order[slot] : L.unit_base + slot;
  deferred = L.units[u].deferred != 0;
  return L.units[u].stage_flags;
}
// One launch per round; the workgroup runs the three stages of its unit's block back to back (chunk buffer and effect state stay in
// LDS). PG_STAGE_OUTLINE: bit 2 set = the tail stage is an out-of-line call (own register allocation); clear = inlined into the kernel
// function. A callee that needs more than the 80 caller-saved VGPRs saves the callee-saved ones it uses to scratch, and scratch is real
// HBM traffic at 1024 workgroups: fifteen registers = 15 MB written and 15 MB read back per 1024-voice block (-enable-ipra does not remove
// those saves). The tail once fitted the caller-saved set and its call was free; it no longer does (FETCH_SIZE / WRITE_SIZE showed
// 496 MB per block against 444 algorithmic), and inlined the kernel still allocates 125 VGPRs without a spill:
// all inline 0.0916 ms per headline block against 0.0939 with the call, same box, interleaved (C5: -0.6 %).
// The launch structure lives in LDS (written by one lane from the scalar registers the arguments arrive in) so that an
// out-of-line stage can take it by pointer. (The kernarg segment is not addressable from a callee: llvm.amdgcn.kernarg.segment.ptr
// lowers to NULL outside kernels.)
#ifndef PG_STAGE_OUTLINE
#define PG_STAGE_OUTLINE 0
#endif
typedef __attribute__((address_space(3))) char* PgLdsPtr;
#if PG_STAGE_OUTLINE & 4
static __device__ __noinline__ void stage3_call(const PgLaunch* L, int slot, int flags, PgLdsPtr smem, int chunk, int4 si) { stage3_run<2, true>(*L, slot, flags, (char*)smem, chunk, si); }
#else
__device__ __forceinline__ void stage3_call(const PgLaunch* L, int slot, int flags, PgLdsPtr smem, int chunk, int4 si) { stage3_run<2, true>(*L, slot, flags, (char*)smem, chunk, si); }
#endif
// The kernel's dynamic LDS as an opaque value: handed to the out-of-line stage as is, constant propagation would put the name
// pg_smem (and with it the offset-table lookup) back into the callee.
__device__ __forceinline__ PgLdsPtr stage_smem_arg() {
  uint32_t a = (uint32_t)(uintptr_t)(PgLdsPtr)pg_smem;
  asm volatile("" : "+s"(a));
  return (PgLdsPtr)(uintptr_t)a;
}
#ifdef PG_DIAG
#define PG_SLOT_STAMP(i) do { if (L.diag && threadIdx.x == 0 && slot < 4096) L.diag[64 + 4 * slot + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PG_SLOT_STAMP(i) do { } while (0)
#endif
// Super-block launches (L.n_chunks > 1): the workgroup renders the consecutive blocks of ITS unit one after the other. Units are
// independent until the mixer sum, so nothing synchronises the workgroups of a launch: they drift apart across the blocks, and the
// latency-bound source / front / tail stages of some run under the bandwidth-bound mid stage of others (with one block per launch
// all resident workgroups pass the stages in lock-step and HBM idles during the first and the last fifth of the kernel).
// A block leaves all of its state in global memory; the barrier between two blocks orders it before the next block's loads.
// (Per-lane values derive from pg_tid(), which keeps the stages' address arithmetic from being hoisted out of this loop.)
template <int LEVEL, int TAG>
__device__ __forceinline__ void stage_fused_body(const PgLaunch& L) {
  if ((int)blockIdx.x >= L.n_units) return;
  const int slot = blockIdx.x;
  const int4 si = stage_slot_info(L, slot);
  if ((si.w >> 24) != LEVEL) return;
  __shared__ PgLaunch sL;  // for the out-of-line stage; the barriers in front of that stage make it visible
  if (threadIdx.x == 0) sL = L;
  const int n_chunks = L.n_chunks > 1 ? L.n_chunks : 1;
#ifdef PG_STAGGER_US   // experiment: in a launch of ONE block every other workgroup starts PG_STAGGER_US microseconds late
  if (n_chunks == 1 && (blockIdx.x & 1)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(PG_STAGGER_US) * 100ull) __builtin_amdgcn_s_sleep(32);
  }
#endif
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    PG_SLOT_STAMP(0);
    const int flags = stage1_run<TAG, true>(L, slot, si, chunk);
    if (flags < 0) return;  // deferred to the generic kernel (never inside a super-block: the host launches those in steady state only)
    __syncthreads();
    PG_SLOT_STAMP(1);
    stage2_run<TAG, true>(L, slot, flags);
    __syncthreads();
    PG_SLOT_STAMP(2);
    stage3_call(&sL, slot, flags, stage_smem_arg(), chunk, si);
    PG_SLOT_STAMP(3);
    __syncthreads();
  }
}
