// Host side of libphonic_gpu.so: the C ABI of include/phonic_gpu.h above the gfx950 kernels.
//
// Mirrors the reference's control plane only as far as the hot path needs it: graph construction as
// `Player` does it (src/player.rs:519-602,773-822,893-939), the mixer's message/event bookkeeping
// (src/source/mixed.rs:294-499,679-712, src/utils/event.rs) — all integer sample-time arithmetic, done
// here on the host and handed to the kernels as per-launch command lists — and the parameter descriptor
// logic (pg_params.h). All per-sample work happens in pg_kernels.hip.
#include "pg_host_internal.h"

// ---- errors ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// ---- device memory helpers (counted: pg_debug_hip_calls) ---------------------------------------------------
static std::atomic<uint64_t> g_n_alloc{0}, g_n_free{0}, g_n_sync{0}, g_n_blocking_copy{0};
hipError_t pg_malloc(void** p, size_t bytes) { g_n_alloc++; return hipMalloc(p, bytes); }
hipError_t pg_host_malloc(void** p, size_t bytes, unsigned flags) { g_n_alloc++; return hipHostMalloc(p, bytes, flags); }
hipError_t pg_free(void* p) { g_n_free++; return hipFree(p); }
hipError_t pg_host_free(void* p) { g_n_free++; return hipHostFree(p); }
hipError_t pg_stream_sync(hipStream_t s) { g_n_sync++; return hipStreamSynchronize(s); }
hipError_t pg_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { g_n_blocking_copy++; return hipMemcpy(d, s, n, k); }
hipError_t pg_memset(void* d, int v, size_t n) { g_n_blocking_copy++; return hipMemset(d, v, n); }

// Test hook (pg_debug_fail_launch_round): the n-th launch round from now fails as if a HIP launch had — exercises the sticky failed state
static std::atomic<int> g_fail_round_countdown{0};
static int graph_fail(pg_graph* g, int code) { g->failed = true; return code; }

// Calls that change the graph (add_* / remove_* / move_* / mode switches) are not real-time calls: they first wait for everything the
// last write enqueued — on the graph's own stream and on the caller's stream the last write used — so that no launch in flight reads a
// table that is about to be re-uploaded, re-allocated or patched.
int graph_quiesce(pg_graph* g) {
  (void)hipSetDevice(g->device);
  HIP_TRY(pg_stream_sync(g->stream));
  if (g->last_stream && g->last_stream != g->stream) HIP_TRY(pg_stream_sync(g->last_stream));
  if (g->unit_stream) HIP_TRY(pg_stream_sync(g->unit_stream));   // (every launch there is awaited by a sum on the write's stream: drained already)
  g->cmds_since_sync = 0;
  g->rows_free_fresh = false;
  // Sources that were removed (RemoveSource, RemoveAllPendingEvents, their mixer's removal) and whose removal a topology upload has carried to the
  // device since: no unit names them any more and nothing is in flight — their PCM / ring / staging memory goes back now (a host that cycles
  // add_stream_voice / remove_voice would otherwise grow without bound until pg_graph_destroy; round-4 advisor finding). The reference drops a
  // removed source on its collector thread (src/player.rs:1178-1196), never on the audio thread: this is a graph-changing call, not a write.
  for (int id : g->retired_ready) {
    HostVoice& hv = g->voices[id];
    if (hv.d_pcm) (void)pg_free(hv.d_pcm);
    if (hv.d_stage) (void)pg_free(hv.d_stage);
    if (hv.h_ring) (void)pg_host_free(hv.h_ring);
    hv.d_pcm = nullptr; hv.d_stage = nullptr; hv.h_ring = nullptr;
    if (hv.stream) g->stream_voices.erase(std::remove(g->stream_voices.begin(), g->stream_voices.end(), id), g->stream_voices.end());
  }
  g->retired_ready.clear();
  return PG_OK;
}

// topology tables: unit -> voices / effects. Unit slots are stable; PgUnit state fields live on the device and are
// preserved: only (kind, n_voices, voice_off, n_fx, fx_off) are patched.
__global__ void pg_patch_units_kernel(PgUnit* units, const PgUnit* topo, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  units[i].kind = topo[i].kind;
  units[i].n_voices = topo[i].n_voices; units[i].voice_off = topo[i].voice_off;
  units[i].n_fx = topo[i].n_fx; units[i].fx_off = topo[i].fx_off;
  units[i].static_defer = topo[i].static_defer;
  units[i].voice0 = topo[i].voice0;
  units[i].fx0 = topo[i].fx0;
  units[i].staged = topo[i].staged;
  units[i].child_off = topo[i].child_off; units[i].n_children = topo[i].n_children;
  units[i].maybe_ramping = 1;  // topology changed: the generic kernel re-evaluates the steady-state condition on the next block
}
// Device feedback for the "nothing left to play" early return of MixedSource::write (mixed.rs:664-670, :715): how many main-mixer sources
// are still alive, written to a mapped host word (visible to the host once the stream has drained).
__global__ void pg_status_kernel(const PgVoice* voices, const int32_t* idx, int n, volatile int* status) {  // one wave
  int active = 0;
  for (int i = threadIdx.x; i < n; i += 64) active += voices[idx[i]].active ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) active += __shfl_xor(active, off, 64);
  if (threadIdx.x == 0) { *status = active; __threadfence_system(); }
}

static int new_unit(pg_graph* g, int kind) {
  PgUnit u;
  memset(&u, 0, sizeof u);
  u.kind = kind;
  u.effects_bypassed = 1;  // MixedSource::new: effects_bypassed = true (mixed.rs:230)
  int idx = -1;
  if (g->d_units.push(u, &idx)) return -1;
  g->h_units.push_back(u);
  g->topo_dirty = true;
  return idx;
}

static int rebuild_topology(pg_graph* g, hipStream_t stream) {
  std::vector<int32_t> vidx, fidx;
  uint32_t kind_mask = 0;
  std::vector<PgUnit> topo = g->h_units;
  // sub-mixers and the bus
  for (size_t m = 0; m < g->mixers.size(); ++m) {
    HostMixer& mx = g->mixers[m];
    PgUnit& u = topo[mx.unit_slot];
    u.fx_off = (int)fidx.size(); u.n_fx = (int)mx.fx.size();
    u.static_defer = 0;
    u.fx0 = mx.fx.empty() ? 0 : mx.fx[0];
    for (int f : mx.fx) {
      fidx.push_back(f);
      const int k = g->fx[f]->kind;  // kinds with a time-parallel path (pg_fx_fast.h: fx_fast_eligible)
      if (m != 0 && !mx.removed) kind_mask |= 1u << k;
      if (m != 0 && (k == PG_FX_FILTER || k == PG_FX_EQ5 || k == PG_FX_DISTORTION || k == PG_FX_DELAY || k == PG_FX_CHORUS || k == PG_FX_COMPRESSOR || k == PG_FX_GATE)) g->wide = true;
      if (!(k == PG_FX_GAIN || k == PG_FX_PANNING || k == PG_FX_FILTER || k == PG_FX_EQ5 || k == PG_FX_DELAY || k == PG_FX_REVERB || k == PG_FX_CHORUS || k == PG_FX_COMPRESSOR || k == PG_FX_GATE || k == PG_FX_DISTORTION)) u.static_defer = 1;
      if (m != 0 && k == PG_FX_GAIN && (int)g->fx[f]->init_raw[1] != 0) g->wide = true;  // DC filter: blocked scan, compiled into the wide variants only
    }
    // staged pipeline: a sub-mixer whose chain is [Gain (no DC filter) | Panning]* -> Reverb
    u.staged = 0;
    if (m != 0 && !u.static_defer && !mx.fx.empty() && g->fx[mx.fx.back()]->kind == PG_FX_REVERB) {
      u.staged = 1;  // 1: leading Gain / Panning only (lean staged kernel); 2: also Filter, Eq5, Delay, Distortion (wide staged kernel)
      for (size_t i = 0; i + 1 < mx.fx.size(); ++i) {
        const int k = g->fx[mx.fx[i]]->kind;
        if ((k == PG_FX_GAIN && (int)g->fx[mx.fx[i]]->init_raw[1] == 0) || k == PG_FX_PANNING) continue;
        if (k == PG_FX_GAIN || k == PG_FX_FILTER || k == PG_FX_EQ5 || k == PG_FX_DELAY || k == PG_FX_DISTORTION) { if (u.staged) u.staged = 2; }
        else u.staged = 0;
      }
    }
    if (m == 0) { u.n_voices = 0; u.voice_off = 0; continue; }
    if (!mx.children.empty()) u.staged = 0;  // sums its sub-mixers' rows first (previous level): not a staged unit; the fast kernels take it in steady state
    for (int v : mx.voices) {
      if ((g->voices[v].outer || g->voices[v].stream) && u.staged) u.staged = 3;  // ResampledSource staging / host-fed ring: the staged kernel whose source stage carries the adapters (round 5: pg_stage_fused_adapt_kernel)
      if (g->voices[v].outer) g->any_outer = true;
    }
    u.voice_off = (int)vidx.size(); u.n_voices = (int)mx.voices.size();
    u.voice0 = mx.voices.empty() ? 0 : g->voices[mx.voices[0]].dev_index;
    for (int v : mx.voices) vidx.push_back(g->voices[v].dev_index);
  }
  // main-mixer sources: one unit each
  g->order.clear();
  g->levels.clear();
  int max_depth = 1;
  for (size_t m = 1; m < g->mixers.size(); ++m) if (!g->mixers[m].removed) max_depth = std::max(max_depth, g->mixers[m].depth);
  std::vector<int> row_of_mixer(g->mixers.size(), -1);
  for (int d = max_depth; d >= 1; --d) {
    Level lv;
    lv.off = (int)g->order.size();
    for (size_t m = 1; m < g->mixers.size(); ++m) {
      if (g->mixers[m].depth != d || g->mixers[m].removed) continue;
      row_of_mixer[m] = (int)g->order.size();
      g->order.push_back(g->mixers[m].unit_slot);
    }
    lv.cnt = (int)g->order.size() - lv.off;
    g->levels.push_back(lv);
  }
  std::vector<int2> child_rows;
  for (size_t m = 1; m < g->mixers.size(); ++m) {
    PgUnit& u = topo[g->mixers[m].unit_slot];
    u.child_off = (int)child_rows.size(); u.n_children = (int)g->mixers[m].children.size();
    for (int c : g->mixers[m].children) child_rows.push_back(make_int2(row_of_mixer[c], g->mixers[c].unit_slot));
  }
  for (int v : g->mixers[0].voices) {
    int slot = g->source_unit_of_voice[v];
    PgUnit& u = topo[slot];
    u.voice_off = (int)vidx.size(); u.n_voices = 1; u.n_fx = 0; u.fx_off = 0;
    u.static_defer = 0;
    if (g->voices[v].outer) g->any_outer = true;
    u.voice0 = g->voices[v].dev_index;
    vidx.push_back(g->voices[v].dev_index);
    g->order.push_back(slot);
  }
  g->n_graph_units = (int)g->order.size();
  g->levels.back().cnt = g->n_graph_units - g->levels.back().off;  // the main mixer's sources belong to the last level
  int rc;
  if ((rc = g->d_child_rows.upload_async(child_rows, stream))) return rc;
  {
    std::vector<int4> info;
    for (int slot : g->order) {
      const PgUnit& u = topo[slot];
      info.push_back(make_int4(slot, u.voice0, u.n_fx > 0 ? fidx[u.fx_off + u.n_fx - 1] : 0, (u.n_voices & 0xffffff) | (u.staged << 24)));
    }
    if ((rc = g->d_slot_info.upload_async(info, stream))) return rc;
    std::vector<int2> sfx;
    for (int slot : g->order) {
      const PgUnit& u = topo[slot];
      sfx.push_back(make_int2(u.n_fx > 0 ? fidx[u.fx_off] : -1, u.n_fx > 1 ? fidx[u.fx_off + 1] : -1));
    }
    if ((rc = g->d_slot_fx.upload_async(sfx, stream))) return rc;
    std::vector<int4> lead;
    for (int slot : g->order) {
      const PgUnit& u = topo[slot];
      auto at = [&](int k) { return (u.staged && k + 1 < u.n_fx) ? fidx[u.fx_off + k] : -1; };
      lead.push_back(make_int4(at(0), at(1), at(2), 0));
    }
    if ((rc = g->d_slot_lead.upload_async(lead, stream))) return rc;
  }
  g->n_staged = 0; g->n_staged_wide = 0; g->n_staged_adapt = 0; g->n_static_defer = 0;
  for (Level& lv : g->levels) {
    lv.n_staged = lv.n_staged_wide = lv.n_staged_adapt = lv.n_static_defer = 0;
    for (int i = lv.off; i < lv.off + lv.cnt; ++i) {
      const PgUnit& u = topo[g->order[i]];
      lv.n_staged += u.staged ? 1 : 0; lv.n_staged_wide += u.staged == 2 ? 1 : 0; lv.n_staged_adapt += u.staged == 3 ? 1 : 0; lv.n_static_defer += u.static_defer ? 1 : 0;
    }
    g->n_staged += lv.n_staged; g->n_staged_wide += lv.n_staged_wide; g->n_staged_adapt += lv.n_staged_adapt; g->n_static_defer += lv.n_static_defer;
  }
  g->h_units = topo;
  g->fast_kind_mask = kind_mask;
  if ((rc = g->d_voice_index.upload_async(vidx, stream))) return rc;
  if ((rc = g->d_fx_index.upload_async(fidx, stream))) return rc;
  if ((rc = g->d_order.upload_async(g->order, stream))) return rc;
  // elements appended since the last build (units, effects, voices, schedule-cache entries), then the topology fields of every unit
  if ((rc = g->d_units.flush_async(stream)) || (rc = g->d_fx.flush_async(stream)) || (rc = g->d_voices.flush_async(stream)) || (rc = g->d_sched.flush_async(stream))) return rc;
  if ((rc = g->d_topo.upload_async(topo, stream))) return rc;
  int n = (int)topo.size();
  hipLaunchKernelGGL(pg_patch_units_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, g->d_units.d, g->d_topo.d, n);
  HIP_TRY(hipGetLastError());
  if ((size_t)std::max(g->n_graph_units, 1) > g->unit_out_rows || g->unit_out_blocks < std::max<size_t>(g->max_blocks, (PG_MAX_FRAMES + g->max_frames - 1) / g->max_frames))
    return set_error(PG_ERR_STATE, "per-unit buffers were not reserved by the mutating call");
  g->topo_dirty = false;
  // (the removals collected so far travel with this upload: once the stream has drained no launch can reach those sources' memory)
  g->retired_ready.insert(g->retired_ready.end(), g->retired_voices.begin(), g->retired_voices.end());
  g->retired_voices.clear();
  g->last_change_round = g->launch_counter;  // the patch kernel marked every unit: the generic kernel must look at them again
  return PG_OK;
}

// Per-unit output tables the write path needs: one per block of a super-block launch, and one per piece of a chunk (the mixer sum runs
// behind a chunk's last piece).
static size_t graph_table_blocks(const pg_graph* g) { return std::max<size_t>(g->max_blocks, (PG_MAX_FRAMES + g->max_frames - 1) / g->max_frames); }
// Device capacity for the graph as the host mirror describes it now. Called at the end of every mutating call (after graph_quiesce):
// this is where the library allocates — grow-by-doubling — so that rebuild_topology and everything else inside write never does.
static int graph_reserve(pg_graph* g) {
  int rc;
  const size_t n_units = g->h_units.size(), n_voices = g->voices.size(), n_fx = g->fx.size(), n_mixers = g->mixers.size();
  if ((rc = g->d_topo.reserve(n_units)) || (rc = g->d_order.reserve(n_units)) || (rc = g->d_slot_info.reserve(n_units)) || (rc = g->d_slot_fx.reserve(n_units)) || (rc = g->d_slot_lead.reserve(n_units)) ||
      (rc = g->d_voice_index.reserve(n_voices)) || (rc = g->d_fx_index.reserve(n_fx)) || (rc = g->d_child_rows.reserve(n_mixers)))
    return rc;
  if (!g->d_cmd_ring) {
    HIP_TRY(pg_malloc((void**)&g->d_cmd_ring, PG_CMD_RING * sizeof(PgCmd)));
    HIP_TRY(pg_host_malloc((void**)&g->h_cmd_ring, PG_CMD_RING * sizeof(PgCmd), hipHostMallocDefault));
  }
  // per-unit output rows (one table of rows per block of a super-block launch), deferral list, stage hand-over, mixer partials
  const size_t rows = std::max<size_t>(n_units, 1);
  const size_t blocks = graph_table_blocks(g);
  if (rows > g->unit_out_rows || blocks != g->unit_out_blocks) {
    if (g->d_unit_out) (void)pg_free(g->d_unit_out);
    g->d_unit_out = nullptr;
    size_t nr = rows > g->unit_out_rows ? std::max(rows, g->unit_out_rows * 2) : g->unit_out_rows;
    HIP_TRY(pg_malloc((void**)&g->d_unit_out, (nr * g->stride * blocks + 4) * sizeof(float)));  // +4: the mixer sum reads whole float4s (odd max_frames)
    if (g->d_audible_tab) (void)pg_free(g->d_audible_tab);
    g->d_audible_tab = nullptr;
    HIP_TRY(pg_malloc((void**)&g->d_audible_tab, nr * blocks * sizeof(int32_t)));  // one `audible` word per unit row and block
    HIP_TRY(pg_memset(g->d_audible_tab, 0, nr * blocks * sizeof(int32_t)));
    g->unit_out_rows = nr;
    g->unit_out_blocks = blocks;
  }
  if (rows > g->defer_rows) {
    if (g->d_defer) (void)pg_free(g->d_defer);
    g->d_defer = nullptr;
    const size_t nr = std::max(rows, g->defer_rows * 2);
    HIP_TRY(pg_malloc((void**)&g->d_defer, (4 + nr) * sizeof(int32_t)));
    HIP_TRY(pg_memset(g->d_defer, 0, (4 + nr) * sizeof(int32_t)));
    g->defer_rows = nr;
  }
  bool any_reverb = false;
  for (const auto& f : g->fx) any_reverb |= f->kind == PG_FX_REVERB;
  if (any_reverb && rows > g->stage_rows) {  // hand-over buffer of the one-launch-per-stage mode (pg_graph_set_staged(g, 2))
    if (g->d_stage) (void)pg_free(g->d_stage);
    g->d_stage = nullptr;
    const size_t nr = std::max(rows, g->stage_rows * 2);
    HIP_TRY(pg_malloc((void**)&g->d_stage, nr * (size_t)PG_STAGE_BUF_DOUBLES * sizeof(double)));
    g->stage_rows = nr;
  }
  const size_t prow = (rows + 15) / 16;
  if (prow > g->partial_rows) {
    if (g->d_partial) (void)pg_free(g->d_partial);
    g->d_partial = nullptr;
    const size_t nr = std::max(prow, g->partial_rows * 2);
    HIP_TRY(pg_malloc((void**)&g->d_partial, (nr * g->stride + 4) * sizeof(float)));
    g->partial_rows = nr;
  }
  return PG_OK;
}
// blocking upload of whatever still waits in the staging blocks (introspection calls that read device state)
static int graph_flush_blocking(pg_graph* g) {
  int rc;
  if ((rc = g->d_units.flush()) || (rc = g->d_fx.flush()) || (rc = g->d_voices.flush()) || (rc = g->d_sched.flush())) return rc;
  return PG_OK;
}

extern "C" {

const char* pg_last_error_message(void) { return g_last_error.c_str(); }
int pg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -PG_ERR_DEVICE;
  return n;
}

// ---- graph --------------------------------------------------------------------------------------------------
pg_graph* pg_graph_create(uint32_t sample_rate, uint32_t channel_count, size_t max_frames, int device) {
  if (channel_count != 2) { set_error(PG_ERR_PARAMETER, "only stereo graphs are supported (reference default: enforce_stereo_playback)"); return nullptr; }
  if (sample_rate == 0 || max_frames == 0 || max_frames > PG_MAX_FRAMES) { set_error(PG_ERR_PARAMETER, "max_frames must be in 1..=%d", PG_MAX_FRAMES); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { set_error(PG_ERR_DEVICE, "hipSetDevice(%d) failed — phonic_gpu needs an AMD GPU", device); return nullptr; }
  std::unique_ptr<pg_graph> g(new pg_graph());
  g->device = device; g->sample_rate = sample_rate; g->channels = 2; g->max_frames = max_frames;
  g->stride = (uint32_t)(2 * max_frames);
  if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { set_error(PG_ERR_DEVICE, "hipStreamCreate failed"); return nullptr; }
  // staging of pg_graph_write (host buffers): whole chunks, so at least one of PG_MAX_FRAMES frames; one `audible` word per piece of a chunk
  // (and per block of a super-block launch)
  g->bus_frames = std::max<size_t>(max_frames, PG_MAX_FRAMES);
  g->audible_slots = std::max<size_t>(PG_AUDIBLE_SLOTS, (PG_MAX_FRAMES + max_frames - 1) / max_frames);
  if (pg_malloc((void**)&g->d_bus, (2 * g->bus_frames + 4) * sizeof(float)) != hipSuccess || pg_malloc((void**)&g->d_audible, g->audible_slots * sizeof(int)) != hipSuccess ||
      pg_host_malloc((void**)&g->h_pinned, (2 * g->bus_frames + 4) * sizeof(float), hipHostMallocDefault) != hipSuccess) {
    set_error(PG_ERR_DEVICE, "device allocation failed");
    return nullptr;
  }
  (void)pg_memset(g->d_audible, 0, g->audible_slots * sizeof(int));
  if (pg_malloc((void**)&g->d_bus_progress, 2 * PG_BUS_PIPELINE_MAX * 8) == hipSuccess) (void)pg_memset(g->d_bus_progress, 0xff, 2 * PG_BUS_PIPELINE_MAX * 8); else g->d_bus_progress = nullptr;
  if (pg_malloc((void**)&g->d_error, 32) == hipSuccess) (void)pg_memset(g->d_error, 0, 32); else g->d_error = nullptr;   // [0] flags, [2..3] u64 deferred unit-blocks, [4..5] u64 generic launches with work
  if (pg_host_malloc((void**)&g->h_feedback, 64, hipHostMallocMapped) == hipSuccess) {
    g->h_feedback[0] = ~0ull;  // nothing reported yet
    g->h_feedback[3] = ~0ull;
    g->h_feedback[1] = 0; g->h_feedback[2] = 0;  // [1] status word (graph_enqueue_status), [2] consistency flags the kernels mirror here
    if (hipHostGetDevicePointer((void**)&g->d_feedback, g->h_feedback, 0) != hipSuccess) g->d_feedback = nullptr;
  }
  { const char* e = getenv("PHONIC_BUS_PIPELINE"); if (e && e[0] == '0') g->bus_pipeline = false; }
  { const char* e = getenv("PHONIC_BUS_OVERLAP"); if (e && e[0] == '0') g->overlap_bus = false; }
  { const char* e = getenv("PHONIC_BUS_GROUP"); if (e && atoi(e) > 0) g->bus_group = (uint64_t)atoi(e); }
  if (hipStreamCreateWithFlags(&g->unit_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g->ev_units_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g->ev_rows_free, hipEventDisableTiming) != hipSuccess) {
    g->overlap_bus = false;   // (not fatal: the launches stay on one stream)
  }
  { const char* e = getenv("PHONIC_CONCURRENT_GENERIC"); if (e && e[0] == '0') g->concurrent_generic = false; }
  if (!g->unit_stream || hipEventCreateWithFlags(&g->ev_scan_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->ev_generic_done, hipEventDisableTiming) != hipSuccess)
    g->concurrent_generic = false;
  g->mixers.emplace_back();
  g->mixers[0].depth = 0;
  g->mixers[0].unit_slot = new_unit(g.get(), UNIT_BUS);
  if (g->mixers[0].unit_slot < 0) return nullptr;
  if (graph_reserve(g.get())) return nullptr;
  return g.release();
}

void pg_graph_destroy(pg_graph* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)pg_stream_sync(g->stream);
  if (g->last_stream && g->last_stream != g->stream) (void)pg_stream_sync(g->last_stream);
  for (auto& v : g->voices) { if (v.d_pcm) (void)pg_free(v.d_pcm); if (v.d_stage) (void)pg_free(v.d_stage); if (v.h_ring) (void)pg_host_free(v.h_ring); }
  for (auto& f : g->fx) if (f->d_mem) (void)pg_free(f->d_mem);
  g->d_units.release(); g->d_voices.release(); g->d_fx.release(); g->d_voice_index.release(); g->d_fx_index.release(); g->d_order.release();
  g->d_sched.release(); g->d_slot_info.release(); g->d_slot_fx.release(); g->d_slot_lead.release(); g->d_child_rows.release(); g->d_topo.release();
  if (g->d_cmd_ring) (void)pg_free(g->d_cmd_ring);
  if (g->d_cmd_overflow) (void)pg_free(g->d_cmd_overflow);
  if (g->h_cmd_ring) (void)pg_host_free(g->h_cmd_ring);
  if (g->d_unit_out) (void)pg_free(g->d_unit_out);
  if (g->d_audible_tab) (void)pg_free(g->d_audible_tab);
  if (g->d_partial) (void)pg_free(g->d_partial);
  if (g->d_stage) (void)pg_free(g->d_stage);
  if (g->d_defer) (void)pg_free(g->d_defer);
  if (g->d_bus) (void)pg_free(g->d_bus);
  if (g->d_audible) (void)pg_free(g->d_audible);
  if (g->d_error) (void)pg_free(g->d_error);
  if (g->d_bus_progress) (void)pg_free(g->d_bus_progress);
  if (g->h_pinned) (void)pg_host_free(g->h_pinned);
  if (g->h_feedback) (void)pg_host_free(g->h_feedback);
  for (auto& e : g->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : g->ev_bus_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : g->ev_gen_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (g->ev_units_done) (void)hipEventDestroy(g->ev_units_done);
  if (g->ev_rows_free) (void)hipEventDestroy(g->ev_rows_free);
  if (g->unit_stream) { (void)pg_stream_sync(g->unit_stream); (void)hipStreamDestroy(g->unit_stream); }
  if (g->ev_scan_done) (void)hipEventDestroy(g->ev_scan_done);
  if (g->ev_generic_done) (void)hipEventDestroy(g->ev_generic_done);
  (void)hipStreamDestroy(g->stream);
  delete g;
}

int pg_graph_add_mixer_to(pg_graph* g, int parent_mixer_id) {
  if (parent_mixer_id < 0 || parent_mixer_id >= (int)g->mixers.size() || g->mixers[parent_mixer_id].removed) return -set_error(PG_ERR_NOT_FOUND, "Mixer with id %d not found", parent_mixer_id);
  if (graph_quiesce(g)) return -graph_fail(g, PG_ERR_DEVICE);
  int slot = new_unit(g, UNIT_SUBMIXER);
  if (slot < 0) return -graph_fail(g, PG_ERR_DEVICE);
  const int id = (int)g->mixers.size();
  g->mixers.emplace_back();
  g->mixers.back().unit_slot = slot;
  g->mixers.back().parent = parent_mixer_id;
  g->mixers.back().depth = g->mixers[parent_mixer_id].depth + 1;
  if (parent_mixer_id != 0) g->mixers[parent_mixer_id].children.push_back(id);
  g->mixers.back().events.reserve(64);
  if (graph_reserve(g)) return -graph_fail(g, PG_ERR_DEVICE);
  return id;
}
int pg_graph_add_mixer(pg_graph* g) { return pg_graph_add_mixer_to(g, 0); }

int pg_graph_add_effect(pg_graph* g, int mixer_id, int kind, const pg_effect_init* init) {
  if (mixer_id < 0 || mixer_id >= (int)g->mixers.size() || g->mixers[mixer_id].removed) return -set_error(PG_ERR_NOT_FOUND, "Mixer with id %d not found", mixer_id);
  if (graph_quiesce(g)) return -graph_fail(g, PG_ERR_DEVICE);
  std::unique_ptr<HostFx> h(new HostFx());
  int rc = host_fx_from_init(kind, init, *h);
  if (rc) return -rc;
  PgFx fx;
  rc = build_fx_device_state(*h, g->sample_rate, g->device, false, fx);
  if (rc) return -rc;
  int idx = -1;
  rc = g->d_fx.push(fx, &idx);
  if (rc) return -graph_fail(g, rc);
  g->fx.push_back(std::move(h));
  g->fx_mixer.push_back(mixer_id);
  g->mixers[mixer_id].fx.push_back(idx);
  if (!g->fx_kind_tab.append((int8_t)kind)) return -set_error(PG_ERR_STATE, "too many effects");
  g->topo_dirty = true;
  if (graph_reserve(g)) return -graph_fail(g, PG_ERR_DEVICE);
  return idx;
}

// Player::stop_all_sources (src/player.rs:1012-1045): every playing file source is told to stop (fades out from the next write on,
// like pg_graph_stop_voice at "now"), and every mixer gets MixerMessage::RemoveAllPendingEvents (src/source/mixed.rs:298-305), which
// at the start of the next write drops the sources that have not started yet and the events scheduled after that write's position.
static void drain_control_messages(pg_graph* g);
static void stop_all_voices_now(pg_graph* g) {
  for (size_t v = 0; v < g->voices.size(); ++v) {
    if (g->voices[v].mixer < 0 || !g->voices[v].transient) continue;   // (stop_all_sources stops the transient sources, player.rs:1013-1031)
    PgCmd c;
    memset(&c, 0, sizeof c);
    c.type = CMD_VOICE_STOP; c.target = g->voices[v].dev_index; c.value64 = 0; c.param = (int)v;
    g->mixers[g->voices[v].mixer].messages.push_back(c);
  }
  for (HostMixer& mx : g->mixers) if (!mx.removed) { mx.remove_pending = true; mx.remove_event_seq = g->event_seq; mx.remove_voice_limit = g->voices.size(); }
}
static int ctrl_push(pg_graph* g, const pgc::CtrlMsg& m) {
  if (!g->ctrl.push(m)) return set_error(PG_ERR_QUEUE_FULL, "mixer's message queue is full");  // Error::SendError
  return PG_OK;
}
int pg_graph_stop_all_voices(pg_graph* g) {
  pgc::CtrlMsg m;
  memset(&m, 0, sizeof m);
  m.type = pgc::CT_STOP_ALL;
  return ctrl_push(g, m);
}
static void apply_remove_pending(pg_graph* g, uint64_t pos) {
  for (HostMixer& mx : g->mixers) {
    if (!mx.remove_pending) continue;
    mx.remove_pending = false;
    for (size_t i = 0; i < mx.voices.size();) {
      const int v = mx.voices[i];
      if ((size_t)v < mx.remove_voice_limit && g->voices[v].transient && g->voices[v].start_time > pos) {   // (source.is_transient && source.start_time > time.pos_in_frames, mixed.rs:300-302)
        mx.messages.erase(std::remove_if(mx.messages.begin(), mx.messages.end(), [v](const PgCmd& c) { return c.param == v; }), mx.messages.end());
        g->voices[v].mixer = -1;
        g->voice_alive_tab.set((size_t)v, 0);
        g->retired_voices.push_back(v);
        mx.voices.erase(mx.voices.begin() + i);
        if (&mx == &g->mixers[0] && g->main_active_voices > 0) g->main_active_voices -= 1;
        g->topo_dirty = true;
      } else ++i;
    }
    const uint64_t lim = mx.remove_event_seq;
    mx.events.erase(std::remove_if(mx.events.begin(), mx.events.end(), [pos, lim](const Event& e) { return e.seq < lim && e.sample_time > pos; }), mx.events.end());
    mx.bus_events.erase(std::remove_if(mx.bus_events.begin(), mx.bus_events.end(), [pos, lim](const Event& e) { return e.seq < lim && e.sample_time > pos; }), mx.bus_events.end());
  }
}

// Player::remove_mixer -> MixerMessage::RemoveMixer to the parent (src/player.rs:825-867, src/source/mixed.rs:422-424): from the next write
// on the parent no longer sums this sub-mixer; its effects, sources, nested sub-mixers and their pending events go with it (dropped
// with the SubMixerProcessor in the reference). Ids are never reused.
int pg_graph_remove_mixer(pg_graph* g, int mixer_id) {
  if (mixer_id == 0) return set_error(PG_ERR_PARAMETER, "Cannot remove the main mixer");
  if (mixer_id < 0 || mixer_id >= (int)g->mixers.size() || g->mixers[mixer_id].removed) return set_error(PG_ERR_NOT_FOUND, "Mixer with id %d not found", mixer_id);
  if (graph_quiesce(g)) return graph_fail(g, PG_ERR_DEVICE);
  drain_control_messages(g);  // messages sent before the removal still find their target
  std::vector<int>& siblings = g->mixers[g->mixers[mixer_id].parent].children;
  siblings.erase(std::remove(siblings.begin(), siblings.end(), mixer_id), siblings.end());
  std::vector<int> gone(1, mixer_id);
  for (size_t i = 0; i < gone.size(); ++i) {
    HostMixer& mx = g->mixers[gone[i]];
    for (int c : mx.children) gone.push_back(c);
    for (int f : mx.fx) { g->fx_mixer[f] = -1; g->fx_kind_tab.set((size_t)f, -1); }
    for (int v : mx.voices) { g->voices[v].mixer = -1; g->voice_alive_tab.set((size_t)v, 0); g->retired_voices.push_back(v); }
    mx.children.clear(); mx.fx.clear(); mx.voices.clear(); mx.events.clear(); mx.messages.clear(); mx.bus_events.clear();
    mx.removed = true;
  }
  g->topo_dirty = true;
  return PG_OK;
}

// Player::remove_effect -> MixerMessage::RemoveEffect (src/player.rs:977-990, src/source/mixed.rs:433-440): the effect leaves its mixer's
// chain at the start of the next write; events still queued for it would find no effect when they fire (mixed.rs:880-924) and are
// dropped here. The id is never reused; its device state stays allocated until the graph is destroyed (the reference drops the
// effect on the collector thread, never on the audio thread).
int pg_graph_remove_effect(pg_graph* g, int effect_id) {
  if (effect_id < 0 || effect_id >= (int)g->fx.size() || g->fx_mixer[effect_id] < 0) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  if (graph_quiesce(g)) return graph_fail(g, PG_ERR_DEVICE);
  drain_control_messages(g);
  HostMixer& mx = g->mixers[g->fx_mixer[effect_id]];
  mx.fx.erase(std::remove(mx.fx.begin(), mx.fx.end(), effect_id), mx.fx.end());
  // Events already scheduled for the effect stay in the mixer's queue (RemoveEffect only takes the effect out of the chain, mixed.rs:433-440):
  // when they come due they find no effect and do nothing — but they still split the block there, and the per-call logic of the effect
  // processors and sub-mixers (tail counters start on one call and count down from the next, effect.rs:113-127) sees the extra call.
  auto disarm = [effect_id](Event& e) { if ((e.cmd.type == CMD_FX_PARAM || e.cmd.type == CMD_FX_RESET) && e.cmd.target == effect_id) { e.cmd.type = CMD_NOP; e.cmd.target = 0; } };
  for (Event& e : mx.events) disarm(e);
  for (Event& e : mx.bus_events) disarm(e);
  g->fx[effect_id]->last_mixer = g->fx_mixer[effect_id];
  g->fx_mixer[effect_id] = -1;
  g->fx_kind_tab.set((size_t)effect_id, -1);
  g->topo_dirty = true;
  return PG_OK;
}
// Player::move_effect -> MixerMessage::MoveEffect (src/player.rs:942-972, src/source/mixed.rs:441-462). movement: PG_MOVE_DIRECTION
// (offset < 0 towards the start, clamped to the chain), PG_MOVE_START, PG_MOVE_END; mixer_id must be the effect's mixer.
int pg_graph_move_effect(pg_graph* g, int effect_id, int mixer_id, int movement, int offset) {
  if (effect_id < 0 || effect_id >= (int)g->fx.size() || g->fx_mixer[effect_id] < 0) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  if (g->fx_mixer[effect_id] != mixer_id) return set_error(PG_ERR_PARAMETER, "Effect %d does not belong to mixer %d", effect_id, mixer_id);
  if (movement < PG_MOVE_DIRECTION || movement > PG_MOVE_END) return set_error(PG_ERR_PARAMETER, "unknown effect movement %d", movement);
  if (graph_quiesce(g)) return graph_fail(g, PG_ERR_DEVICE);
  std::vector<int>& fx = g->mixers[mixer_id].fx;
  const auto it = std::find(fx.begin(), fx.end(), effect_id);
  if (it == fx.end()) return PG_OK;  // (logged and ignored in the reference)
  const int current_pos = (int)(it - fx.begin());
  fx.erase(it);
  int new_pos;
  if (movement == PG_MOVE_DIRECTION) new_pos = std::max(0, std::min((int)fx.size(), current_pos + offset));
  else new_pos = movement == PG_MOVE_START ? 0 : (int)fx.size();
  fx.insert(fx.begin() + new_pos, effect_id);
  g->topo_dirty = true;
  return PG_OK;
}

int pg_graph_add_voice(pg_graph* g, int mixer_id, const float* pcm, size_t n_frames, uint32_t src_channels, uint32_t src_rate,
                       const pg_voice_options* opt) {
  if (mixer_id < 0 || mixer_id >= (int)g->mixers.size() || g->mixers[mixer_id].removed) return -set_error(PG_ERR_NOT_FOUND, "Mixer with id %d not found", mixer_id);
  // AudioFileBuffer::new validation (file/buffer.rs:22-60)
  if (src_rate == 0) return -set_error(PG_ERR_PARAMETER, "file buffer sample rate must be > 0");
  if (src_channels != 1 && src_channels != 2) return -set_error(PG_ERR_PARAMETER, "only mono and stereo file buffers are supported");
  if (n_frames == 0 || !pcm) return -set_error(PG_ERR_PARAMETER, "file buffer must not be empty");
  drain_control_messages(g);  // control calls made before this one come first (a stop_all_voices must not take this source with it)
  pg_voice_options def;
  if (!opt) { pg_voice_options_default(&def); opt = &def; }
  if (!(opt->speed > 0.0)) return -set_error(PG_ERR_PARAMETER, "speed must be > 0");
  if (opt->volume < 0.0f || opt->panning < -1.0f || opt->panning > 1.0f) return -set_error(PG_ERR_PARAMETER, "invalid volume or panning");
  if (graph_quiesce(g)) return -graph_fail(g, PG_ERR_DEVICE);
  PgVoice v;
  memset(&v, 0, sizeof v);
  size_t n_samples = n_frames * src_channels;
  void* d_pcm = nullptr;
  if (pg_malloc(&d_pcm, n_samples * sizeof(float)) != hipSuccess) return -graph_fail(g, set_error(PG_ERR_DEVICE, "pg_malloc(pcm) failed"));
  if (pg_memcpy(d_pcm, pcm, n_samples * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -graph_fail(g, set_error(PG_ERR_DEVICE, "pcm upload failed"));
  v.pcm = (const float*)d_pcm;
  // the rate the file source is created with: the mixer's, unless the caller asks for a ResampledSource behind it (source_rate)
  const uint32_t inner_rate = opt->source_rate ? opt->source_rate : g->sample_rate;
  v.n_samples = n_samples; v.channels = src_channels; v.src_rate = src_rate; v.out_rate = inner_rate;
  // FileSourceImpl::new: resampler file_rate -> (out_rate / speed) as u32  (file/common.rs:78-86); ratio = (in/out as f64) as f32 (cubic.rs:164)
  uint32_t res_out = (uint32_t)d2u64((double)inner_rate / opt->speed);
  if (res_out == 0) return -set_error(PG_ERR_PARAMETER, "Invalid resampling ratio");
  v.ratio = (float)((double)src_rate / (double)res_out);
  if (!(v.ratio > 0.0f) || v.ratio > 64.0f) return -set_error(PG_ERR_PARAMETER, "Invalid resampling ratio");
  // repeat / loop range (preloaded.rs:89-104): no embedded loop points (decoding is out of scope)
  v.repeat = opt->has_repeat ? opt->repeat : 0;
  v.repeat_count = v.repeat;
  if (opt->has_loop_range) {
    uint64_t fc = n_frames;
    v.has_loop = 1;
    v.loop_start = std::min<uint64_t>(opt->loop_start, fc > 0 ? fc - 1 : 0);
    v.loop_end = std::min<uint64_t>(opt->loop_end, fc);
    if (v.loop_start >= v.loop_end) return -set_error(PG_ERR_PARAMETER, "file buffer loop range is out of bounds");
  }
  // VolumeFader::new + optional fade-in (file/common.rs:69-75, fader.rs:36-91)
  v.fader_state = 0; v.fader_current = 1.0f; v.fader_target = 1.0f; v.fader_inertia = 1.0f;
  if (opt->fade_in_seconds > 0.0f) {
    v.fader_state = 1; v.fader_current = 0.0f; v.fader_target = 1.0f;
    float samples_duration = (float)inner_rate * opt->fade_in_seconds / 4.605f;
    v.fader_inertia = 1.0f - std::exp(-1.0f / samples_duration);
  }
  v.fade_out_seconds = opt->fade_out_seconds;
  // AmplifiedSource / PannedSource: ExponentialSmoothedValue::new(value, source.sample_rate())
  ParamSpec exp_spec = {0, PG_PARAM_FLOAT, 0, 0, 0, 0, 0, 0, 0, "", S_EXP, 0};
  v.volume = make_smooth(exp_spec, opt->volume, g->sample_rate);
  v.panning = make_smooth(exp_spec, opt->panning, g->sample_rate);
  v.start_time = opt->start_time;
  v.active = 1;
  v.persistent = opt->non_transient != 0;
  v.current_speed = opt->speed; v.target_speed = opt->speed; v.speed_glide_rate = 0.0f; v.samples_to_next_speed_update = 0;
  {  // resampler schedule cache class: voices sharing the f32 ratio; the first one publishes
    uint32_t rb;
    memcpy(&rb, &v.ratio, 4);
    auto it = g->sched_class_of_ratio.find(rb);
    if (it == g->sched_class_of_ratio.end()) {
      PgSchedEntry blank;
      memset(&blank, 0, sizeof blank);
      int i0 = -1, i1 = -1;
      if (g->d_sched.push(blank, &i0) || g->d_sched.push(blank, &i1)) return -graph_fail(g, PG_ERR_DEVICE);
      it = g->sched_class_of_ratio.emplace(rb, i0 / 2).first;
      v.sched_rep = 1;
    }
    v.sched_class = it->second;
  }
  // ConvertedSource::new (converted.rs:15-45): a source whose rate is not the mixer's gets ResampledSource::new(source, mixer rate, Default)
  // = a cubic resampler inner rate -> mixer rate (speed 1.0, resampled.rs:44-98) with two TempBuffers of 512 frames
  void* d_stage = nullptr;
  if (inner_rate != g->sample_rate) {
    v.outer_on = 1;
    v.outer_ratio = (float)((double)inner_rate / (double)g->sample_rate);
    const size_t stage_floats = 2 * 512 * (size_t)src_channels;
    if (pg_malloc(&d_stage, stage_floats * sizeof(float)) != hipSuccess || pg_memset(d_stage, 0, stage_floats * sizeof(float)) != hipSuccess)
      return -graph_fail(g, set_error(PG_ERR_DEVICE, "hipMalloc(staging) failed"));
    v.stage_in = (float*)d_stage; v.stage_out = v.stage_in + 512 * src_channels;
    v.sched_class = -1;  // (rendered serially on the generic kernel: no schedule cache)
  }
  int dev_index = -1;
  int rc = g->d_voices.push(v, &dev_index);
  if (rc) return -graph_fail(g, rc);
  int id = (int)g->voices.size();
  { HostVoice hv; hv.mixer = mixer_id; hv.dev_index = dev_index; hv.start_time = opt->start_time; hv.d_pcm = d_pcm; hv.d_stage = d_stage; hv.outer = inner_rate != g->sample_rate;
    hv.transient = opt->non_transient == 0; g->voices.push_back(hv); }
  g->source_unit_of_voice.push_back(-1);
  // AddSource: sort by start time, insert BEFORE equal start times (mixed.rs:324-329)
  HostMixer& mx = g->mixers[mixer_id];
  size_t pos = 0;
  while (pos < mx.voices.size() && g->voices[mx.voices[pos]].start_time < opt->start_time) ++pos;
  mx.voices.insert(mx.voices.begin() + pos, id);
  if (mixer_id == 0) {
    int slot = new_unit(g, UNIT_SOURCE);
    if (slot < 0) return -graph_fail(g, PG_ERR_DEVICE);
    g->source_unit_of_voice[id] = slot;
    g->main_active_voices += 1;
    g->ever_had_main_voice = true;
  }
  if (!g->voice_alive_tab.append(1)) return -set_error(PG_ERR_STATE, "too many voices");
  g->topo_dirty = true;
  if (graph_reserve(g)) return -graph_fail(g, PG_ERR_DEVICE);
  return id;
}

// ---- host-fed sources ------------------------------------------------------------------------------------------------------------
// The mixer takes any `Box<dyn Source>` (MixerMessage::AddSource, src/source/mixed.rs:117-123): synth sources (SynthSourceImpl,
// src/source/synth/common.rs:194-263), streamed files, generators. Such a source lives on the host; to sit inside a GPU (sub-)mixer
// its output has to reach the device. A stream voice is the device half: a ring of `capacity_frames` frames in device memory that the
// host fills with what it pulled from its source (pg_graph_feed_voice, at the source's own rate and channel count), read by
// stream_source_write where a file voice reads its preloaded buffer; behind it the same adapter chain as for any source —
// ResampledSource when the rate is not the mixer's (src/source/converted.rs:15-45), mono -> stereo, AmplifiedSource, PannedSource
// (src/player.rs:540-558). All memory (device ring, pinned staging ring) is reserved here; feed and write allocate nothing.
int pg_graph_add_stream_voice(pg_graph* g, int mixer_id, uint32_t channels, uint32_t rate, size_t capacity_frames, const pg_voice_options* opt) {
  if (mixer_id < 0 || mixer_id >= (int)g->mixers.size() || g->mixers[mixer_id].removed) return -set_error(PG_ERR_NOT_FOUND, "Mixer with id %d not found", mixer_id);
  if (rate == 0) return -set_error(PG_ERR_PARAMETER, "source sample rate must be > 0");
  if (channels != 1 && channels != 2) return -set_error(PG_ERR_PARAMETER, "only mono and stereo sources are supported");
  if (capacity_frames < 1024 || capacity_frames > (1u << 30)) return -set_error(PG_ERR_PARAMETER, "ring capacity must be in 1024..=2^30 frames");
  drain_control_messages(g);
  pg_voice_options def;
  if (!opt) { pg_voice_options_default(&def); opt = &def; }
  if (opt->volume < 0.0f || opt->panning < -1.0f || opt->panning > 1.0f) return -set_error(PG_ERR_PARAMETER, "invalid volume or panning");
  if (graph_quiesce(g)) return -graph_fail(g, PG_ERR_DEVICE);
  PgVoice v;
  memset(&v, 0, sizeof v);
  const size_t ring_floats = capacity_frames * channels;
  void* d_ring = nullptr;
  float* h_ring = nullptr;
  void* d_stage = nullptr;
  auto release = [&]() { if (d_ring) (void)pg_free(d_ring); if (h_ring) (void)pg_host_free(h_ring); if (d_stage) (void)pg_free(d_stage); };   // (error returns below)
  if (pg_malloc(&d_ring, ring_floats * sizeof(float)) != hipSuccess || pg_memset(d_ring, 0, ring_floats * sizeof(float)) != hipSuccess ||
      pg_host_malloc((void**)&h_ring, ring_floats * sizeof(float), hipHostMallocDefault) != hipSuccess) {
    release();
    return -graph_fail(g, set_error(PG_ERR_DEVICE, "ring allocation failed"));
  }
  v.pcm = (const float*)d_ring;
  v.channels = channels; v.src_rate = rate; v.out_rate = rate; v.ratio = 1.0f;
  v.stream_on = 1; v.stream_cap = (uint32_t)capacity_frames; v.stream_fed = 0;
  v.fader_state = 0; v.fader_current = 1.0f; v.fader_target = 1.0f; v.fader_inertia = 1.0f;
  ParamSpec exp_spec = {0, PG_PARAM_FLOAT, 0, 0, 0, 0, 0, 0, 0, "", S_EXP, 0};
  v.volume = make_smooth(exp_spec, opt->volume, g->sample_rate);
  v.panning = make_smooth(exp_spec, opt->panning, g->sample_rate);
  v.start_time = opt->start_time;
  v.active = 1;
  v.persistent = opt->non_transient != 0;
  v.current_speed = 1.0; v.target_speed = 1.0;
  v.sched_class = -1;
  if (rate != g->sample_rate) {  // ConvertedSource::new -> ResampledSource::new(source, mixer rate, Default)  (converted.rs:15-45, resampled.rs:44-98)
    v.outer_on = 1;
    v.outer_ratio = (float)((double)rate / (double)g->sample_rate);
    const size_t stage_floats = 2 * 512 * (size_t)channels;
    if (pg_malloc(&d_stage, stage_floats * sizeof(float)) != hipSuccess || pg_memset(d_stage, 0, stage_floats * sizeof(float)) != hipSuccess) {
      release();
      return -graph_fail(g, set_error(PG_ERR_DEVICE, "hipMalloc(staging) failed"));
    }
    v.stage_in = (float*)d_stage; v.stage_out = v.stage_in + 512 * channels;
  }
  int dev_index = -1;
  int rc = g->d_voices.push(v, &dev_index);
  if (rc) { release(); return -graph_fail(g, rc); }
  const int id = (int)g->voices.size();
  HostVoice hv;
  hv.mixer = mixer_id; hv.dev_index = dev_index; hv.start_time = opt->start_time; hv.d_pcm = d_ring; hv.d_stage = d_stage; hv.outer = rate != g->sample_rate;
  hv.stream = true; hv.h_ring = h_ring; hv.channels = channels; hv.cap_frames = capacity_frames;
  hv.transient = opt->non_transient == 0;
  g->voices.push_back(hv);
  g->stream_voices.push_back(id);
  g->source_unit_of_voice.push_back(-1);
  HostMixer& mx = g->mixers[mixer_id];
  size_t pos = 0;
  while (pos < mx.voices.size() && g->voices[mx.voices[pos]].start_time < opt->start_time) ++pos;
  mx.voices.insert(mx.voices.begin() + pos, id);
  if (mixer_id == 0) {
    int slot = new_unit(g, UNIT_SOURCE);
    if (slot < 0) return -graph_fail(g, PG_ERR_DEVICE);
    g->source_unit_of_voice[id] = slot;
    g->main_active_voices += 1;
    g->ever_had_main_voice = true;
  }
  if (!g->voice_alive_tab.append(2)) return -set_error(PG_ERR_STATE, "too many voices");   // (2: alive and host-fed — no seek, no speed)
  g->topo_dirty = true;
  if (graph_reserve(g)) return -graph_fail(g, PG_ERR_DEVICE);
  return id;
}
static HostVoice* stream_voice(pg_graph* g, int voice_id) {
  if (voice_id < 0 || voice_id >= (int)g->voices.size() || !g->voices[voice_id].stream) { set_error(PG_ERR_NOT_FOUND, "Source with id %d is not a host-fed source", voice_id); return nullptr; }
  if (g->voices[voice_id].mixer < 0) { set_error(PG_ERR_NOT_FOUND, "Source with id %d is gone (its mixer was removed)", voice_id); return nullptr; }   // (nobody would ever drain its ring)
  return &g->voices[voice_id];
}
// The next `n_frames` frames of the host's source (interleaved, the voice's channel count) -> the pinned staging ring; they travel to the
// device ring at the top of the next write, on that write's stream. Owner thread only. PG_ERR_QUEUE_FULL (nothing taken) when the ring
// has no room for all of them: room = capacity - (fed - consumed), `consumed` as last reported by pg_graph_stream_voice_consumed.
int pg_graph_feed_voice(pg_graph* g, int voice_id, const float* frames, size_t n_frames) {
  HostVoice* hv = stream_voice(g, voice_id);
  if (!hv) return PG_ERR_NOT_FOUND;
  if (hv->ended) return set_error(PG_ERR_STATE, "the stream has been ended");
  if (n_frames == 0) return PG_OK;
  if (!frames) return set_error(PG_ERR_PARAMETER, "no frames");
  if (hv->fed - hv->consumed_known + n_frames > hv->cap_frames) return set_error(PG_ERR_QUEUE_FULL, "the source's ring is full");
  const size_t C = hv->channels, w0 = (size_t)(hv->fed % hv->cap_frames), first = std::min(n_frames, hv->cap_frames - w0);
  memcpy(hv->h_ring + w0 * C, frames, first * C * sizeof(float));
  if (first < n_frames) memcpy(hv->h_ring, frames + first * C, (n_frames - first) * C * sizeof(float));
  hv->fed += n_frames;
  return PG_OK;
}
// The host's source is exhausted (Source::is_exhausted, src/source.rs:88-93): the voice ends when everything fed has been played.
int pg_graph_end_stream_voice(pg_graph* g, int voice_id) {
  HostVoice* hv = stream_voice(g, voice_id);
  if (!hv) return PG_ERR_NOT_FOUND;
  hv->ended = true;
  return PG_OK;
}
// Frames of the fed stream the device has read so far (waits for the graph's stream; -1 on failure). Frees that much room for feeds.
int64_t pg_graph_stream_voice_consumed(pg_graph* g, int voice_id) {
  HostVoice* hv = stream_voice(g, voice_id);
  if (!hv) return -1;
  (void)hipSetDevice(g->device);
  if (pg_stream_sync(g->stream) != hipSuccess || (g->last_stream && g->last_stream != g->stream && pg_stream_sync(g->last_stream) != hipSuccess)) return -1;
  if (graph_flush_blocking(g)) return -1;
  uint64_t rd = 0;
  if (pg_memcpy(&rd, (const char*)(g->d_voices.d + hv->dev_index) + offsetof(PgVoice, playback_pos), 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  hv->consumed_known = rd;
  return (int64_t)rd;
}

// ---- control calls: any thread, concurrently with write() ------------------------------------------------------------------
// Each call validates what it can from immutable descriptor tables and the append-only id tables (kind of the effect, whether the
// voice still exists), resolves Raw / Normalized to a raw value, and pushes ONE record into the graph's lock-free ring — nothing
// else is touched. The thread inside write() drains the ring at the top of the call (drain_control_messages == process_messages,
// src/source/mixed.rs:294-499): sample-time-tagged messages become sorted events of their mixer, StopSource stays a message.
static int fx_kind_of(pg_graph* g, int effect_id) {  // -1: unknown or removed
  if (effect_id < 0 || (size_t)effect_id >= g->fx_kind_tab.size()) return -1;
  return (int)g->fx_kind_tab.get((size_t)effect_id);
}
static bool voice_alive(pg_graph* g, int voice_id) { return voice_id >= 0 && (size_t)voice_id < g->voice_alive_tab.size() && g->voice_alive_tab.get((size_t)voice_id) != 0; }
// Seek and speed exist on FilePlaybackHandle only (src/player/handles/file.rs): a host-fed source (the host's own `dyn Source` behind a ring) has
// neither a position to seek to nor a resampler to re-target — the device would rewind the ring's read position over stale frames.
static bool voice_is_host_fed(pg_graph* g, int voice_id) { return voice_id >= 0 && (size_t)voice_id < g->voice_alive_tab.size() && g->voice_alive_tab.get((size_t)voice_id) == 2; }

int pg_graph_schedule_param(pg_graph* g, int effect_id, uint32_t fourcc, float value, int is_normalized, uint64_t sample_time) {
  const int kind = fx_kind_of(g, effect_id);
  if (kind < 0) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  int pi = find_param(kind, fourcc);
  if (pi < 0) return set_error(PG_ERR_PARAMETER, "Unknown parameter: 0x%08x for effect '%s'", fourcc, KINDS[kind].name);
  float raw;
  if (!resolve_update(KINDS[kind].params[pi], value, is_normalized != 0, raw)) return PG_OK;  // logged + ignored in the reference
  pgc::CtrlMsg m;
  memset(&m, 0, sizeof m);
  m.type = pgc::CT_FX_PARAM; m.id = effect_id; m.param = pi; m.value = raw; m.sample_time = sample_time;
  return ctrl_push(g, m);
}
int pg_graph_schedule_reset(pg_graph* g, int effect_id, uint64_t sample_time) {
  const int kind = fx_kind_of(g, effect_id);
  if (kind < 0) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  if (kind != PG_FX_DELAY && kind != PG_FX_REVERB && kind != PG_FX_CHORUS)
    return set_error(PG_ERR_PARAMETER, "%sEffect: Invalid/unknown message payload", KINDS[kind].name);
  pgc::CtrlMsg m;
  memset(&m, 0, sizeof m);
  m.type = pgc::CT_FX_RESET; m.id = effect_id; m.sample_time = sample_time;
  return ctrl_push(g, m);
}
static int voice_message(pg_graph* g, int voice_id, int type, float value, double dvalue, uint64_t sample_time) {
  if (!voice_alive(g, voice_id)) return set_error(PG_ERR_NOT_FOUND, "Source with id %d not found", voice_id);
  pgc::CtrlMsg m;
  memset(&m, 0, sizeof m);
  m.type = type; m.id = voice_id; m.value = value; m.dvalue = dvalue; m.sample_time = sample_time;
  return ctrl_push(g, m);
}
int pg_graph_set_voice_volume(pg_graph* g, int voice_id, float volume, uint64_t sample_time) { return voice_message(g, voice_id, pgc::CT_VOICE_VOLUME, volume, 0.0, sample_time); }
int pg_graph_set_voice_panning(pg_graph* g, int voice_id, float panning, uint64_t sample_time) { return voice_message(g, voice_id, pgc::CT_VOICE_PAN, panning, 0.0, sample_time); }
int pg_graph_set_voice_speed(pg_graph* g, int voice_id, double speed, float glide, uint64_t sample_time) {
  if (!(speed > 0.0)) return set_error(PG_ERR_PARAMETER, "speed must be > 0");
  if (voice_is_host_fed(g, voice_id)) return set_error(PG_ERR_PARAMETER, "Source with id %d is host-fed: it takes volume, panning and stop only", voice_id);
  return voice_message(g, voice_id, pgc::CT_VOICE_SPEED, glide, speed, sample_time);
}
int pg_graph_seek_voice(pg_graph* g, int voice_id, double position_seconds, uint64_t sample_time) {
  if (!(position_seconds >= 0.0)) return set_error(PG_ERR_PARAMETER, "seek position must be >= 0");
  if (voice_is_host_fed(g, voice_id)) return set_error(PG_ERR_PARAMETER, "Source with id %d is host-fed: it takes volume, panning and stop only", voice_id);
  return voice_message(g, voice_id, pgc::CT_VOICE_SEEK, 0.0f, position_seconds, sample_time);
}
int pg_graph_stop_voice(pg_graph* g, int voice_id, uint64_t sample_time) {  // MixerMessage::StopSource (mixed.rs:389-400): not an event
  return voice_message(g, voice_id, pgc::CT_VOICE_STOP, 0.0f, 0.0, sample_time);
}
int pg_graph_remove_voice(pg_graph* g, int voice_id) {  // MixerMessage::RemoveSource (mixed.rs:149-151,400-402)
  const int rc = voice_message(g, voice_id, pgc::CT_VOICE_REMOVE, 0.0f, 0.0, 0);
  // the id is dead for every later call from here on (a second remove, a volume change: PG_ERR_NOT_FOUND as the header says), not only once the
  // next write has drained the message; messages pushed before this one are still delivered (the ring keeps their order)
  if (rc == PG_OK) g->voice_alive_tab.set((size_t)voice_id, 0);
  return rc;
}

static void push_event(pg_graph* g, int mixer, uint64_t sample_time, const PgCmd& cmd) {
  HostMixer& mx = g->mixers[mixer];
  Event e{sample_time, g->event_seq++, cmd, mixer};
  size_t pos = mx.events.size();  // partition_point(|e| e.sample_time <= sample_time); automation usually arrives in time order: search from the back
  while (pos > 0 && mx.events[pos - 1].sample_time > sample_time) --pos;
  mx.events.insert(mx.events.begin() + pos, e);
}
// MixedSource::process_messages (src/source/mixed.rs:294-499) for the whole graph: the writing thread only.
static void drain_control_messages(pg_graph* g) {
  pgc::CtrlMsg m;
  while (g->ctrl.pop(m)) {
    PgCmd c;
    memset(&c, 0, sizeof c);
    switch (m.type) {
      case pgc::CT_FX_PARAM: {
        if (m.id < 0 || m.id >= (int)g->fx.size()) break;
        if (g->fx_mixer[m.id] < 0) {  // removed in the meantime: the event will find no effect (mixed.rs:880-924), but it is an event of that mixer all the same
          const int lm = g->fx[m.id]->last_mixer;
          if (lm >= 0 && lm < (int)g->mixers.size() && !g->mixers[lm].removed) { c.type = CMD_NOP; push_event(g, lm, m.sample_time, c); }
          break;
        }
        HostFx& h = *g->fx[m.id];
        h.target[m.param] = m.value;
        // a Gain whose DC filter gets switched on later needs the kernel variants that carry the DC scan: classify the chain again
        if (h.kind == PG_FX_GAIN && m.param != P_GAIN_GAIN && (int)m.value != 0 && (int)h.init_raw[1] == 0) { h.init_raw[1] = m.value; g->topo_dirty = true; }
        c.type = CMD_FX_PARAM; c.target = m.id; c.param = m.param; c.value = m.value; c.value64 = fx_param_aux(h.kind, m.param, m.value, g->sample_rate);
        push_event(g, g->fx_mixer[m.id], m.sample_time, c);
      } break;
      case pgc::CT_FX_RESET: {
        if (m.id < 0 || m.id >= (int)g->fx.size()) break;
        if (g->fx_mixer[m.id] < 0) {
          const int lm = g->fx[m.id]->last_mixer;
          if (lm >= 0 && lm < (int)g->mixers.size() && !g->mixers[lm].removed) { c.type = CMD_NOP; push_event(g, lm, m.sample_time, c); }
          break;
        }
        c.type = CMD_FX_RESET; c.target = m.id;
        push_event(g, g->fx_mixer[m.id], m.sample_time, c);
      } break;
      case pgc::CT_STOP_ALL: stop_all_voices_now(g); break;
      default: {
        if (m.id < 0 || m.id >= (int)g->voices.size() || g->voices[m.id].mixer < 0) break;
        const HostVoice& hv = g->voices[m.id];
        c.target = hv.dev_index; c.param = m.id;
        if (m.type == pgc::CT_VOICE_REMOVE) {  // remove_matching_sources(|s| s.playback_id == playback_id): gone before this write renders anything
          HostMixer& mx = g->mixers[hv.mixer];
          mx.voices.erase(std::remove(mx.voices.begin(), mx.voices.end(), m.id), mx.voices.end());
          mx.messages.erase(std::remove_if(mx.messages.begin(), mx.messages.end(), [&](const PgCmd& x) { return x.param == m.id; }), mx.messages.end());
          // events already queued for the source stay the mixer's events: when they come due they find no source (mixed.rs:810-845) but still
          // split the block there — like the events of a removed effect
          for (Event& e : mx.events) if ((e.cmd.type == CMD_VOICE_VOLUME || e.cmd.type == CMD_VOICE_PAN || e.cmd.type == CMD_VOICE_SPEED || e.cmd.type == CMD_VOICE_SEEK) && e.cmd.param == m.id) {
            e.cmd.type = CMD_NOP; e.cmd.target = 0;
            if (hv.mixer == 0) e.cmd.param = -1;
          }
          // (the count of live main-mixer sources comes from the device after a synchronous write; a source the mixer keeps is live by
          // definition — a transient one may have ended already: its removal shows with the next count)
          if (hv.mixer == 0 && !hv.transient && g->main_active_voices > 0) g->main_active_voices -= 1;
          g->voices[m.id].mixer = -1;
          g->voice_alive_tab.set((size_t)m.id, 0);
          g->retired_voices.push_back(m.id);
          g->topo_dirty = true;
          break;
        }
        if (m.type == pgc::CT_VOICE_STOP) { c.type = CMD_VOICE_STOP; c.value64 = m.sample_time; g->mixers[hv.mixer].messages.push_back(c); break; }
        if (m.type == pgc::CT_VOICE_VOLUME) { c.type = CMD_VOICE_VOLUME; c.value = m.value; }
        else if (m.type == pgc::CT_VOICE_PAN) { c.type = CMD_VOICE_PAN; c.value = m.value; }
        else if (m.type == pgc::CT_VOICE_SPEED) { c.type = CMD_VOICE_SPEED; c.value = m.value; memcpy(&c.value64, &m.dvalue, 8); }
        else { c.type = CMD_VOICE_SEEK; memcpy(&c.value64, &m.dvalue, 8); }
        push_event(g, hv.mixer, m.sample_time, c);
      } break;
    }
  }
}

int pg_graph_diag(pg_graph* g, unsigned long long* out, int n) {  // diagnostic builds: shader-clock stamps of workgroup 0
  (void)hipSetDevice(g->device);
  const int cap = 64 + 4 * 4096;  // 64 stamps of workgroup 0, then {start, stage-1 end, stage-2 end, end} (s_memrealtime, 100 MHz) per launch slot
  if (!g->d_diag) { HIP_TRY(pg_malloc((void**)&g->d_diag, (size_t)cap * 8)); HIP_TRY(pg_memset(g->d_diag, 0, (size_t)cap * 8)); return PG_OK; }
  HIP_TRY(pg_stream_sync(g->stream));
  HIP_TRY(pg_memcpy(out, g->d_diag, (size_t)(n > cap ? cap : n) * 8, hipMemcpyDeviceToHost));
  return PG_OK;
}
int pg_graph_set_max_blocks_per_launch(pg_graph* g, int n_blocks) {
  if (n_blocks < 1 || n_blocks > 64) return set_error(PG_ERR_PARAMETER, "blocks per launch must be in 1..=64");
  { int rc = graph_quiesce(g); if (rc) return rc; }
  // staging of pg_graph_write (host buffers): one super-block + the status words
  float* nb = nullptr; float* np = nullptr;
  const size_t frames = std::max<size_t>(g->max_frames * (size_t)n_blocks, PG_MAX_FRAMES);
  const size_t words = 2 * frames + 4;
  HIP_TRY(pg_malloc((void**)&nb, words * sizeof(float)));
  if (pg_host_malloc((void**)&np, words * sizeof(float), hipHostMallocDefault) != hipSuccess) { (void)pg_free(nb); return set_error(PG_ERR_DEVICE, "pinned allocation failed"); }
  (void)pg_free(g->d_bus); (void)pg_host_free(g->h_pinned);
  g->d_bus = nb; g->h_pinned = np;
  g->bus_frames = frames;
  g->max_blocks = (size_t)n_blocks;
  g->topo_dirty = true;
  return graph_reserve(g);  // the per-unit output table grows here, never inside write
}
// dynamic LDS of the launches whose occupancy is budgeted (tools/check_kernel_resources.py): which 0 = the staged single launch, 1 = a fast unit
// kernel for the effect kinds of `kind_mask`, 2 = the wide staged single launch
size_t pg_debug_lds_bytes(int which, uint32_t n_frames, uint32_t kind_mask) {
  if (which == 2) return pg_stage_lds_bytes(0, n_frames, true);   // the wide staged single launch
  return which == 0 ? pg_stage_lds_bytes(0, n_frames) : pg_unit_lds_bytes(n_frames, pg_fast_scratch_bytes(kind_mask));
}
// (a process-wide fault injector has no business in a production process: it only arms when PHONIC_DEBUG_HOOKS=1 is in the environment)
void pg_debug_fail_launch_round(int nth) {
  static const bool armed = [] { const char* e = getenv("PHONIC_DEBUG_HOOKS"); return e && e[0] == '1'; }();
  g_fail_round_countdown.store(armed && nth > 0 ? nth : 0);
}
void pg_debug_hip_calls(uint64_t out[4]) {
  out[0] = g_n_alloc.load(); out[1] = g_n_free.load(); out[2] = g_n_sync.load(); out[3] = g_n_blocking_copy.load();
}
int pg_graph_device_errors(pg_graph* g) {
  (void)hipSetDevice(g->device);
  if (!g->d_error) return 0;
  int32_t e = 0;
  if (pg_stream_sync(g->stream) != hipSuccess || pg_memcpy(&e, g->d_error, 4, hipMemcpyDeviceToHost) != hipSuccess) return -PG_ERR_DEVICE;
  return (int)e;
}
int pg_graph_set_defer_bus(pg_graph* g, int defer) { g->defer_bus = defer != 0; return PG_OK; }
int pg_graph_set_fast_math(pg_graph* g, int level) { g->fast = level != 0; g->last_change_round = g->launch_counter; return PG_OK; }
const char* pg_graph_dominant_kernel(pg_graph* g) {
  if (g->topo_dirty) { (void)graph_quiesce(g); (void)rebuild_topology(g, g->stream); (void)pg_stream_sync(g->stream); }
  if (!g->fast || g->n_static_defer * 2 > g->n_graph_units) return "pg_unit_kernel";
  const int n_lean = g->n_staged - g->n_staged_wide - g->n_staged_adapt;
  const int n_handled = g->staged_mode == 1 ? g->n_staged : n_lean;
  if (g->staged_mode && n_handled > 0) {
    if (g->staged_mode == 2) return n_handled < g->n_graph_units ? "pg_stage1_kernel + pg_stage2_kernel + pg_stage3_kernel + pg_unit_kernel_fast" : "pg_stage1_kernel + pg_stage2_kernel + pg_stage3_kernel";
    if (n_handled < g->n_graph_units) return "pg_stage_fused_kernel + pg_unit_kernel_fast";
    const int kinds = (n_lean > 0) + (g->n_staged_wide > 0) + (g->n_staged_adapt > 0);
    if (kinds > 1) return n_lean > 0 ? (g->n_staged_wide > 0 ? (g->n_staged_adapt > 0 ? "pg_stage_fused_kernel + pg_stage_fused_wide_kernel + pg_stage_fused_adapt_kernel" : "pg_stage_fused_kernel + pg_stage_fused_wide_kernel") : "pg_stage_fused_kernel + pg_stage_fused_adapt_kernel") : "pg_stage_fused_wide_kernel + pg_stage_fused_adapt_kernel";
    return g->n_staged_adapt > 0 ? "pg_stage_fused_adapt_kernel" : (g->n_staged_wide > 0 ? "pg_stage_fused_wide_kernel" : "pg_stage_fused_kernel");
  }
  return g->wide ? (((g->fast_kind_mask & ((1u << PG_FX_REVERB) | (1u << PG_FX_COMPRESSOR))) || g->levels.size() > 1 || g->any_outer) ? "pg_unit_kernel_fast_wide" : "pg_unit_kernel_fast_mid") : "pg_unit_kernel_fast";
}
int pg_graph_set_timing_period(pg_graph* g, int every_n_rounds) {
  g->timing_period = every_n_rounds < 0 ? 0 : every_n_rounds;
  (void)hipSetDevice(g->device);
  while (g->timing_period > 0 && g->ev_pool.size() < 512) {  // created here, never inside write: when all are in use the later rounds go untimed
    hipEvent_t a, b;                                           // until pg_graph_kernel_ms / pg_graph_kernel_stats collects them
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    g->ev_pool.emplace_back(a, b);
    g->ev_blocks.push_back(1);
  }
  while (g->timing_period > 0 && g->ev_gen_pool.size() < 512) {   // the generic kernel's launches (pg_graph_dynamic_stats)
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    g->ev_gen_pool.emplace_back(a, b);
  }
  while (g->timing_period > 0 && g->ev_bus_pool.size() < 512) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    g->ev_bus_pool.emplace_back(a, b);
    g->ev_bus_blocks.push_back(1);
  }
  return PG_OK;
}
int pg_graph_set_staged(pg_graph* g, int mode) { g->staged_mode = (mode < 0 || mode > 2) ? 1 : mode; g->last_change_round = g->launch_counter; return PG_OK; }
int pg_graph_voice_count(pg_graph* g) { return (int)g->voices.size(); }
int pg_graph_synchronize(pg_graph* g) {
  (void)hipSetDevice(g->device);
  HIP_TRY(pg_stream_sync(g->stream));
  return PG_OK;
}
int pg_graph_is_voice_playing(pg_graph* g, int voice_id) {
  if (voice_id < 0 || voice_id >= (int)g->voices.size() || g->voices[voice_id].mixer < 0) return 0;
  (void)hipSetDevice(g->device);
  (void)pg_stream_sync(g->stream);
  if (graph_flush_blocking(g)) return 0;
  PgVoice v;
  if (pg_memcpy(&v, g->d_voices.d + g->voices[voice_id].dev_index, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return v.active && !v.finished;
}
int pg_graph_deferred_units(pg_graph* g) {
  (void)hipSetDevice(g->device);
  (void)pg_stream_sync(g->stream);
  if (!g->h_feedback) return 0;
  // (round << 32 | deferred units) as the generic kernel of the last round that launched it reported; rounds that skipped the launch
  // did so because this word said 0 and nothing had changed since
  const unsigned long long fb = *(volatile unsigned long long*)g->h_feedback;
  return fb == ~0ull ? 0 : (int)(uint32_t)fb;
}
// Timing of the dominant kernel: every timed launch holds a hipEvent pair (riding the dispatch or bracketing the launches). The
// pairs are waited for one by one (hipEventSynchronize on the stop event: the launches may sit on a caller's stream), pairs that
// cannot be read are left out of the sum AND of the count.
int pg_graph_kernel_stats(pg_graph* g, int reset, double* total_ms, uint64_t* launches, uint64_t* blocks) {
  (void)hipSetDevice(g->device);
  double total = 0.0;
  uint64_t n_ok = 0, n_blocks = 0;
  for (size_t i = 0; i < g->ev_used; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(g->ev_pool[i].second) != hipSuccess) continue;
    if (hipEventElapsedTime(&ms, g->ev_pool[i].first, g->ev_pool[i].second) != hipSuccess) continue;
    total += ms; n_ok += 1; n_blocks += g->ev_blocks[i];
  }
  if (total_ms) *total_ms = total;
  if (launches) *launches = n_ok;
  if (blocks) *blocks = n_blocks;
  if (reset) g->ev_used = 0;
  return PG_OK;
}
// What a workload off the steady state costs (bench.py --workload dyn): out[0] = unit-blocks rendered since the last reset (units x blocks of
// max_frames, every level), out[1] = unit-blocks that left the time-parallel kernels for the generic kernel (a command inside the block, a
// smoother still moving, a topology change: the deferral protocol), out[2] = generic launches issued, out[3] = of those, the launches
// that found work; *generic_ms = GPU time of the generic launches that were hipEvent-timed (pg_graph_set_timing_period), *generic_timed their
// number. Waits for the graph's stream.
int pg_graph_dynamic_stats(pg_graph* g, int reset, uint64_t out[4], double* generic_ms, uint64_t* generic_timed) {
  (void)hipSetDevice(g->device);
  HIP_TRY(pg_stream_sync(g->stream));
  if (g->last_stream && g->last_stream != g->stream) HIP_TRY(pg_stream_sync(g->last_stream));
  unsigned long long dev[3] = {0, 0, 0};
  if (g->d_error) HIP_TRY(pg_memcpy(dev, (const char*)g->d_error + 8, 16, hipMemcpyDeviceToHost));
  if (out) { out[0] = g->stat_unit_blocks; out[1] = dev[0]; out[2] = g->stat_generic_launches; out[3] = dev[1]; }
  double total = 0.0;
  uint64_t n_ok = 0;
  for (size_t i = 0; i < g->ev_gen_used; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(g->ev_gen_pool[i].second) != hipSuccess) continue;
    if (hipEventElapsedTime(&ms, g->ev_gen_pool[i].first, g->ev_gen_pool[i].second) != hipSuccess) continue;
    total += ms; n_ok += 1;
  }
  if (generic_ms) *generic_ms = total;
  if (generic_timed) *generic_timed = n_ok;
  if (reset) {
    g->ev_gen_used = 0; g->stat_unit_blocks = 0; g->stat_generic_launches = 0;
    if (g->d_error) HIP_TRY(pg_memset((char*)g->d_error + 8, 0, 16));
  }
  return PG_OK;
}
// The same for the launches of the main mixer's bus chain (timed in the rounds whose unit launches are timed).
int pg_graph_bus_kernel_stats(pg_graph* g, int reset, double* total_ms, uint64_t* launches, uint64_t* blocks) {
  (void)hipSetDevice(g->device);
  double total = 0.0;
  uint64_t n_ok = 0, n_blocks = 0;
  for (size_t i = 0; i < g->ev_bus_used; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(g->ev_bus_pool[i].second) != hipSuccess) continue;
    if (hipEventElapsedTime(&ms, g->ev_bus_pool[i].first, g->ev_bus_pool[i].second) != hipSuccess) continue;
    total += ms; n_ok += 1; n_blocks += g->ev_bus_blocks[i];
  }
  if (total_ms) *total_ms = total;
  if (launches) *launches = n_ok;
  if (blocks) *blocks = n_blocks;
  if (reset) g->ev_bus_used = 0;
  return PG_OK;
}
// Name of the bus launch as the last write issued it ("" when the main mixer has no effects)
const char* pg_graph_bus_kernel(pg_graph* g) {
  const size_t n_fx = g->mixers[0].fx.size();
  if (n_fx == 0) return "";
  const bool pipelined = n_fx <= PG_BUS_PIPELINE_MAX && g->bus_pipeline && g->max_blocks > 1;
  if (pipelined && n_fx >= 2) return "pg_unit_kernel (main mixer's chain: one workgroup per effect, pipelined over the blocks)";
  return pipelined ? "pg_unit_kernel (main mixer's chain: one workgroup, the effect's state resident over the blocks)" : "pg_unit_kernel (main mixer's chain: one workgroup)";
}
double pg_graph_kernel_ms(pg_graph* g, int reset, uint64_t* launches) {
  double total = 0.0;
  uint64_t n = 0;
  (void)pg_graph_kernel_stats(g, reset, &total, &n, nullptr);
  if (launches) *launches = n;
  return n ? total / (double)n : 0.0;
}

// Steady state: the generic kernel of an earlier round (not older than the last topology change / command / mode switch) found
// nothing deferred, and units leave the steady state only through those host-visible events.
static bool graph_steady(const pg_graph* g) {
  if (!(g->fast && g->levels.size() == 1 && g->n_static_defer == 0 && g->d_feedback)) return false;
  // word 3: (round << 32 | units that round deferred for their STATE); units deferred for a command alone do not count — see cmd_may_ramp
  const unsigned long long fb = *(volatile unsigned long long*)(g->h_feedback + 3);
  return fb != ~0ull && (uint32_t)fb == 0u && (int32_t)((uint32_t)(fb >> 32) - (uint32_t)g->last_change_round) >= 0;
}
// A super-block launch sequence renders several blocks of max_frames per workgroup: only in steady state (nobody would render the
// later blocks of a unit that defers itself) and in the single-launch kernels. A bus chain behind the sum is no obstacle: the mixer
// sum leaves one `audible` word per block (PgLaunch::audible_tab) and ONE bus launch walks the summed blocks in order, taking the
// chain's per-block decisions block by block (the reference's chunk loop inside one write call, mixed.rs:679-712); the bus unit is
// rendered by the generic kernel, which needs no steady state of its own.
static bool graph_super_ok(const pg_graph* g) {
  return g->max_blocks > 1 && g->staged_mode != 2 && graph_steady(g);
}

// The command list of one round -> a fresh region of the device ring (asynchronous copy from the pinned ring on the round's stream).
// `h_out` (optional): the caller takes the list to the device itself — as an argument of the round's decision-scan kernel, which leaves it in the
// region reserved here (launch_level) — and gets the pinned copy; no copy is enqueued then.
static int stage_commands(pg_graph* g, const std::vector<PgCmd>& cmds, hipStream_t stream, const PgCmd** d_out, const PgCmd** h_out = nullptr) {
  const size_t n = cmds.size();
  if (n > PG_CMD_RING) {
    // More commands in ONE launch round than the ring holds (tens of thousands of events between two samples): the degenerate case
    // leaves the allocation-free path — a table of its own, released when the next oversized round or the graph's end comes.
    HIP_TRY(pg_stream_sync(stream));
    if (g->d_cmd_overflow) (void)pg_free(g->d_cmd_overflow);
    g->d_cmd_overflow = nullptr;
    HIP_TRY(pg_malloc((void**)&g->d_cmd_overflow, n * sizeof(PgCmd)));
    HIP_TRY(pg_memcpy(g->d_cmd_overflow, cmds.data(), n * sizeof(PgCmd), hipMemcpyHostToDevice));
    *d_out = g->d_cmd_overflow;
    return PG_OK;
  }
  size_t skipped = 0;
  if (g->cmd_head + n > PG_CMD_RING) { skipped = PG_CMD_RING - g->cmd_head; g->cmd_head = 0; }
  if (g->cmds_since_sync + skipped + n > PG_CMD_RING) {  // the region may still be read by a round in flight: wait once per ring revolution
    HIP_TRY(pg_stream_sync(stream));
    g->cmds_since_sync = 0; skipped = 0;
  }
  memcpy(g->h_cmd_ring + g->cmd_head, cmds.data(), n * sizeof(PgCmd));
  if (h_out) *h_out = g->h_cmd_ring + g->cmd_head;
  else HIP_TRY(hipMemcpyAsync(g->d_cmd_ring + g->cmd_head, g->h_cmd_ring + g->cmd_head, n * sizeof(PgCmd), hipMemcpyHostToDevice, stream));
  *d_out = g->d_cmd_ring + g->cmd_head;
  g->cmd_head += n;
  g->cmds_since_sync += skipped + n;
  return PG_OK;
}


// A bus launch that walks several blocks with nothing to apply: one workgroup per effect of the chain, pipelined over the blocks
// (pg_bus_pipeline; a chain of ONE effect too: its state stays in LDS over the blocks and each block's input is requested a block ahead). The progress words carry the launch's round number, so they never need clearing.
static void bus_pipeline_setup(pg_graph* g, PgLaunch& B) {
  const size_t n_fx = g->mixers[0].fx.size();
  if (B.n_chunks > 1 && B.n_chunks <= 64 && B.n_cmds == 0 && n_fx >= 1 && n_fx <= PG_BUS_PIPELINE_MAX && g->d_bus_progress && g->bus_pipeline) {
    B.mode = 3; B.n_units = (int)n_fx; B.bus_progress = g->d_bus_progress;
    B.round = ++g->bus_epoch;  // (the words of an earlier launch never match; 0xffffffff is what the words are created with)
    if (B.round == 0xffffffffu) B.round = g->bus_epoch = 1;
  }
}

#define HIP_TRY_FAIL(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error(PG_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); g->failed = true; return 0; } } while (0)
// The unit kernels of ONE level of the mixer tree for n_chunks blocks of n frames starting at t0: block c goes to the per-unit output table
// `row_block + c` (a chunk's pieces sit in consecutive tables; the mixer sum and the bus chain run behind a chunk's last piece). `round`: the
// launch round the blocks belong to (deferral feedback, schedule-cache bank); grid_off / grid_span: PgLaunch (pg_dev.h).
struct LaunchSpan {
  uint32_t n = 0;            // frames per block
  uint64_t t0 = 0;           // position of block 0
  int n_chunks = 1;
  int row_block = 0;
  uint32_t grid_off = 0, grid_span = 0;
  const PgCmd* d_cmds = nullptr;
  int n_cmds = 0;
  uint64_t round = 0;
  bool timed = false;        // a hipEvent pair is reserved for this span (g->ev_used names it)
  bool generic_idle = false;
  const PgCmd* h_cmds = nullptr;   // the round's commands travel as an argument of its decision-scan kernel (no copy was enqueued): the pinned list
};
static void fill_launch(pg_graph* g, const LaunchSpan& sp, PgLaunch& L) {
  memset(&L, 0, sizeof L);
  L.units = g->d_units.d; L.voices = g->d_voices.d; L.fx = g->d_fx.d;
  L.voice_index = g->d_voice_index.d; L.fx_index = g->d_fx_index.d;
  L.cmds = sp.d_cmds; L.n_cmds = sp.n_cmds;
  L.n_frames = sp.n; L.pos = sp.t0; L.sample_rate = g->sample_rate; L.fast = g->fast;
  L.out_stride = g->stride;
  L.chunk_stride = (uint64_t)g->unit_out_rows * g->stride;
  L.rows_base = g->d_unit_out + (size_t)sp.row_block * (size_t)L.chunk_stride; L.child_rows = g->d_child_rows.d;
  L.call_end = g->call_end;
  L.diag = g->d_diag;
  L.n_chunks = sp.n_chunks; L.error_word = g->d_error;
  L.fast_scratch_bytes = (uint32_t)pg_fast_scratch_bytes(g->fast_kind_mask);
  L.round = (uint32_t)sp.round; L.host_feedback = g->d_feedback;
  L.grid_off = sp.grid_off; L.grid_span = sp.grid_span;
  // (a super-block runs without the resampler schedule cache: its banks alternate per launch, not per block; voices of a cached
  // class replay their schedule serially, and the cache re-validates itself by key when single-block rounds resume)
  L.sched = sp.n_chunks > 1 ? nullptr : g->d_sched.d; L.sched_bank = (int)(sp.round & 1);
}
// Does this round of level li run the generic kernel BESIDE the fast kernels behind a decision-scan kernel (launch_level)?
static bool level_concurrent(const pg_graph* g, size_t li, const LaunchSpan& sp, hipStream_t stream) {
  const Level& lv = g->levels[li];
  const bool time_generic = g->fast && lv.n_static_defer * 2 > lv.cnt;
  return g->fast && lv.cnt > 0 && g->concurrent_generic && !sp.generic_idle && !time_generic && g->levels.size() == 1 && sp.n_chunks == 1 && g->staged_mode != 2 &&
         g->d_defer && g->unit_stream && stream != g->unit_stream;
}
static int launch_level(pg_graph* g, size_t li, const LaunchSpan& sp, hipStream_t stream) {
  if (li == 0 && g_fail_round_countdown.load(std::memory_order_relaxed) > 0 && g_fail_round_countdown.fetch_sub(1) == 1)   // (once per launch round)
    return set_error(PG_ERR_DEVICE, "injected device failure (pg_debug_fail_launch_round)");
  const Level& lv = g->levels[li];
  if (lv.cnt == 0) return PG_OK;
  PgLaunch L;
  fill_launch(g, sp, L);
  const uint32_t n = sp.n;
  const bool nested = g->levels.size() > 1;
  // this level's slice of the per-slot tables: launch slot b of the level = row lv.off + b
  L.n_units = lv.cnt; L.unit_order = g->d_order.d + lv.off;
  L.unit_out = g->d_unit_out + (size_t)sp.row_block * (size_t)L.chunk_stride + (size_t)lv.off * g->stride;
  L.audible_tab = g->d_audible_tab + (size_t)sp.row_block * g->unit_out_rows + lv.off; L.audible_stride = g->unit_out_rows;
  L.slot_info = g->d_slot_info.d + lv.off; L.slot_fx = g->d_slot_fx.d + lv.off; L.slot_lead = g->d_slot_lead.d + lv.off;
  if (g->d_defer) {
    L.defer_count = g->d_defer + (g->defer_phase & 1); L.defer_reset = g->d_defer + ((g->defer_phase & 1) ^ 1); L.defer_list = g->d_defer + 4;
    L.defer_state = g->d_defer + 2 + (g->defer_phase & 1); L.defer_state_reset = g->d_defer + 2 + ((g->defer_phase & 1) ^ 1);
    // (a round that skips the generic launch right behind one that had it: that round's counts would still be in its words when their phase
    // comes round again — a command round may be followed by a steady one at once since voice commands no longer end the steady state)
    if (sp.generic_idle && g->defer_dirty) { HIP_TRY(hipMemsetAsync(g->d_defer, 0, 4 * sizeof(int32_t), stream)); g->defer_dirty = false; }
    if (!sp.generic_idle) g->defer_dirty = true;
  }
  g->defer_phase++;
  // The event pair times the launch(es) that do the bulk of this graph's work: the fast / staged kernels, or — when most units
  // hold an effect without a time-parallel path — the generic kernel. When that is a single launch the events ride on the
  // dispatch itself (hipExtLaunchKernel: no marker packets in the stream); several launches are bracketed by event records.
  const bool timed_here = sp.timed;
  const bool time_generic = g->fast && lv.n_static_defer * 2 > lv.cnt;
  hipEvent_t e0 = timed_here ? g->ev_pool[g->ev_used].first : nullptr, e1 = timed_here ? g->ev_pool[g->ev_used].second : nullptr;
  if (g->fast) {
    // fast kernel; units it cannot run (ramping parameters, effects without a fast path) are deferred ...
    L.mode = 1; L.wide = g->wide ? (((g->fast_kind_mask & ((1u << PG_FX_REVERB) | (1u << PG_FX_COMPRESSOR))) || nested || g->any_outer) ? 1 : 2) : 0;  // (2: the four-per-CU kernel; it neither sums nested mixers nor stages a ResampledSource)
    // reverb-terminated sub-mixers go through the staged kernels; level 2 (wide leading effects) only in the single-launch mode
    const int n_lean = lv.n_staged - lv.n_staged_wide - lv.n_staged_adapt;
    const int n_handled = g->staged_mode == 1 ? lv.n_staged : n_lean;
    const bool staged = g->staged_mode && n_handled > 0 && g->d_stage && n <= 1024;
    const bool lean = staged && n_lean > 0, wide = staged && g->staged_mode == 1 && lv.n_staged_wide > 0, adapt = staged && g->staged_mode == 1 && lv.n_staged_adapt > 0;
    const bool fused = !staged || n_handled < lv.cnt;
    const int n_launches = (staged ? (g->staged_mode == 1 ? (int)lean + (int)wide + (int)adapt : 3) : 0) + (int)fused;
    const bool ride = timed_here && !time_generic && n_launches == 1;       // one dominant launch: timestamps from its dispatch
    const bool bracket = timed_here && !time_generic && n_launches > 1;
    // Off the steady state (commands in the block, smoothers still moving): the units the time-parallel kernels cannot take and the units
    // they can are disjoint, and which is which follows from the unit records as the round begins — a small scan kernel decides it for every
    // unit up front (the decision the fast kernels used to take themselves, one by one), and the generic kernel then runs BESIDE the fast
    // kernels instead of behind them: a commanded unit is a lone workgroup's latency chain of 0.15-0.3 ms, and every round used to wait for it
    // with 250 CUs idle. The generic kernel stays on the write's stream, right behind the scan — it is dispatched first and takes its few CUs
    // (one workgroup each: its register footprint) — the fast kernels go to the graph's second stream behind an event and fill the rest (they
    // share a CU with a generic workgroup one or two at a time instead of four); the sum waits for both. (A first version had it the other way
    // round: the fast kernels, 1024 workgroups that fill every CU, won the race for the machine and the generic kernel ran in their tail:
    // no gain.) One level, single blocks, single-launch staged mode.
    const bool concurrent = level_concurrent(g, li, sp, stream);
    hipStream_t fs = stream;   // the stream of the fast kernels
    if (concurrent) {
      HIP_TRY(pg_launch_defer_scan(L, stream, g->ev_scan_done, sp.h_cmds));   // (the event rides on the scan's dispatch: no marker packet in front of the generic kernel; h_cmds: the round's commands as its argument)
      HIP_TRY(hipStreamWaitEvent(g->unit_stream, g->ev_scan_done, 0));
      PgLaunch G = L;
      G.mode = 2; G.pad_chunks = 1;   // (a pre-scanned round: the fast kernels beside it read PgUnit::deferred — it stays as the scan left it)
      hipEvent_t g0 = nullptr, g1 = nullptr;
      if (timed_here && g->ev_gen_used < g->ev_gen_pool.size()) { g0 = g->ev_gen_pool[g->ev_gen_used].first; g1 = g->ev_gen_pool[g->ev_gen_used].second; g->ev_gen_used++; }
      HIP_TRY(pg_launch_units(G, stream, g0, g1));
      g->stat_generic_launches++;
      L.defer_list = nullptr; L.defer_count = nullptr; L.pad_chunks = 1;   // the fast kernels read their unit's decision word instead of deciding (and append nothing)
      static const bool diag_generic_only = getenv("PHONIC_DIAG_GENERIC") != nullptr;   // (diagnostic builds: both kernels' first workgroups stamp the same words)
      if (diag_generic_only) L.diag = nullptr;
      fs = g->unit_stream;
    }
    if (bracket) HIP_TRY(hipEventRecord(e0, fs));
    L.stage_buf = nullptr; L.staged_on = 0;
    if (staged) {
      L.stage_buf = g->d_stage + (size_t)lv.off * PG_STAGE_BUF_DOUBLES; L.staged_on = g->staged_mode == 1 ? 3 : 1;
      // (concurrent: "the fast kernels are done" rides on their last launch as its stop event when no timing event does)
      HIP_TRY(pg_launch_stages(L, fs, g->staged_mode == 1 ? 1 : 0, lean, wide, adapt, ride ? e0 : nullptr, ride ? e1 : nullptr,
                               (concurrent && !fused && !ride && !bracket && g->staged_mode == 1) ? g->ev_generic_done : nullptr));
    }
    const bool tail_on_fused = concurrent && fused && !bracket && !(ride && !staged);
    if (fused) HIP_TRY(pg_launch_units(L, fs, ride && !staged ? e0 : nullptr, ride && !staged ? e1 : (tail_on_fused ? g->ev_generic_done : nullptr)));
    if (bracket) HIP_TRY(hipEventRecord(e1, fs));
    L.mode = 2;  // ... to the generic kernel, which walks the list of deferred units (skipped while the host knows the list is empty)
    if (concurrent) {   // (the sum behind this level needs both)
      // what the write's stream waits for: the timing stop event when one was taken (it sits behind the fast kernels), else the event that rode
      // on their last launch; a marker packet only where neither exists (the per-stage launch mode never gets here)
      hipEvent_t fast_done = (ride || bracket) ? e1 : g->ev_generic_done;
      const bool rode = ride || bracket || tail_on_fused || (staged && !fused && g->staged_mode == 1);
      if (!rode) HIP_TRY(hipEventRecord(g->ev_generic_done, fs));
      HIP_TRY(hipStreamWaitEvent(stream, fast_done, 0));
    } else if (!sp.generic_idle) {
      hipEvent_t g0 = timed_here && time_generic ? e0 : nullptr, g1 = timed_here && time_generic ? e1 : nullptr;
      if (timed_here && !time_generic && g->ev_gen_used < g->ev_gen_pool.size()) { g0 = g->ev_gen_pool[g->ev_gen_used].first; g1 = g->ev_gen_pool[g->ev_gen_used].second; g->ev_gen_used++; }   // (rides on the dispatch)
      HIP_TRY(pg_launch_units(L, stream, g0, g1));
      g->stat_generic_launches++;
    }
  } else {
    L.mode = 0;
    if (g->d_defer) HIP_TRY(hipMemsetAsync(g->d_defer, 0, 4 * sizeof(int32_t), stream));  // no deferral protocol this round: keep the counters clean
    HIP_TRY(pg_launch_units(L, stream, e0, e1));
  }
  if (timed_here) { g->ev_blocks[g->ev_used] = (uint32_t)sp.n_chunks; g->ev_used++; }
  g->stat_unit_blocks += (uint64_t)lv.cnt * (uint64_t)sp.n_chunks;
  return PG_OK;
}
// The level that holds most units carries the timing events of a round
static size_t timed_level_of(const pg_graph* g) {
  size_t t = 0;
  for (size_t li = 1; li < g->levels.size(); ++li) if (g->levels[li].cnt > g->levels[t].cnt) t = li;
  return t;
}
}  // extern "C"

// MixedSource::write of the main mixer (src/source/mixed.rs:659-719)
// What the host fed since the last write -> the device rings (one or two copies per voice from its pinned ring), then the voice's
// `stream_fed` word; asynchronous on the write's stream, in front of the kernels that read them.
__global__ void pg_store_u64_kernel(unsigned long long* p, unsigned long long v) { *p = v; }
static int flush_stream_feeds(pg_graph* g, hipStream_t stream) {
  for (int id : g->stream_voices) {
    HostVoice& hv = g->voices[id];
    if (hv.mixer < 0 || (hv.sent == hv.fed && hv.ended == hv.ended_sent)) continue;
    const size_t C = hv.channels;
    uint64_t a = hv.sent;
    while (a < hv.fed) {
      const size_t off = (size_t)(a % hv.cap_frames), n = (size_t)std::min<uint64_t>(hv.fed - a, hv.cap_frames - off);
      HIP_TRY(hipMemcpyAsync((float*)hv.d_pcm + off * C, hv.h_ring + off * C, n * C * sizeof(float), hipMemcpyHostToDevice, stream));
      a += n;
    }
    // the new count travels BY VALUE (a kernel argument): a word in pinned memory could be overwritten by the next feed before an earlier
    // write's copy of it has run, and tell the device about frames whose data is still on its way
    hipLaunchKernelGGL(pg_store_u64_kernel, dim3(1), dim3(1), 0, stream, (unsigned long long*)((char*)(g->d_voices.d + hv.dev_index) + offsetof(PgVoice, stream_fed)),
                       (unsigned long long)(hv.fed | (hv.ended ? (1ull << 63) : 0ull)));
    HIP_TRY(hipGetLastError());
    hv.sent = hv.fed; hv.ended_sent = hv.ended;
  }
  return PG_OK;
}

// `begin`: this call opens a write of the main mixer (process_messages first, mixed.rs:659-661). The sharded handle renders one write
// as several calls — one per run of frames between two main-mixer events of ANY shard — and opens it itself (graph_begin_write on
// every shard, then begin = false), so that all shards cut their rounds at the same frames as the one main mixer would.
void graph_begin_write(pg_graph* g, uint64_t pos) {
  drain_control_messages(g);  // process_messages (mixed.rs:294-499)
  apply_remove_pending(g, pos);
  g->messages_due = true;     // StopSource messages travel with the call's first launch
}
// "Return early and avoid touching the buffer if there's nothing to do" (mixed.rs:664-670): no playing sources, no effects, no
// sub-mixers, no events. (Sources of the main mixer that ended are dropped after the write they ended in, :715 — the host learns
// their number from the device after every synchronous write, graph_collect_status.)
bool graph_is_empty(const pg_graph* g) {
  bool no_sub_mixers = true;  // self.mixers.is_empty(): sub-mixers that were removed again do not count
  for (size_t m = 1; m < g->mixers.size(); ++m) no_sub_mixers &= g->mixers[m].removed;
  return g->main_active_voices == 0 && g->mixers[0].fx.empty() && no_sub_mixers && g->mixers[0].events.empty();
}
void drain_control_messages_public(pg_graph* g) { drain_control_messages(g); }
uint64_t graph_next_main_event(const pg_graph* g) { return g->mixers[0].events.empty() ? UINT64_MAX : g->mixers[0].events.front().sample_time; }
// The number of main-mixer sources still alive -> the graph's mapped status word, behind everything enqueued on `stream` so far;
// graph_collect_status reads it once the stream has drained.
int graph_enqueue_status(pg_graph* g, hipStream_t stream) {
  if (!g->d_feedback || g->topo_dirty) return PG_OK;
  const int n_main = (int)g->mixers[0].voices.size();  // the main-mixer voices are the last n_main entries of the voice index table
  hipLaunchKernelGGL(pg_status_kernel, dim3(1), dim3(64), 0, stream, g->d_voices.d, g->d_voice_index.d + (g->d_voice_index.n - (size_t)n_main), n_main,
                     (volatile int*)(g->d_feedback + 1));
  HIP_TRY(hipGetLastError());
  g->status_pending = true;
  return PG_OK;
}
void graph_collect_status(pg_graph* g) {
  if (!g->status_pending) return;
  g->status_pending = false;
  g->main_active_voices = *(volatile int*)(g->h_feedback + 1);
}

// The bus chain over blocks [row_block, +n_chunks) of a chunk (or of several whole chunks) whose sum sits in d_dst. Commands (main-mixer
// effect events, due at the chunk's first frame) ride on a launch of the first block alone: a launch that walks several blocks applies nothing.
static int launch_bus(pg_graph* g, float* d_dst, LaunchSpan sp, int audible_slot, hipStream_t stream) {
  if (g->mixers[0].fx.empty()) return PG_OK;
  auto one = [&](const LaunchSpan& q, float* dst, int slot) -> int {
    PgLaunch B;
    fill_launch(g, q, B);
    B.mode = 0;
    B.n_units = 1; B.unit_order = nullptr; B.unit_base = g->mixers[0].unit_slot;
    B.bus = dst; B.bus_audible = g->d_audible + slot; B.audible_tab = nullptr;
    B.sched = nullptr;
    bus_pipeline_setup(g, B);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (q.timed && g->ev_bus_used < g->ev_bus_pool.size()) { e0 = g->ev_bus_pool[g->ev_bus_used].first; e1 = g->ev_bus_pool[g->ev_bus_used].second; g->ev_bus_blocks[g->ev_bus_used] = (uint32_t)q.n_chunks; g->ev_bus_used++; }
    HIP_TRY(pg_launch_units(B, stream, e0, e1));
    return PG_OK;
  };
  if (sp.n_cmds > 0 && sp.n_chunks > 1) {
    LaunchSpan head = sp;
    head.n_chunks = 1;
    int rc = one(head, d_dst, audible_slot);
    if (rc) return rc;
    sp.d_cmds = nullptr; sp.n_cmds = 0;
    sp.t0 += sp.n; sp.grid_off += sp.n; sp.n_chunks -= 1; sp.row_block += 1;
    return one(sp, d_dst + (size_t)sp.n * 2, audible_slot + 1);
  }
  return one(sp, d_dst, audible_slot);
}

// Commands of the launch that renders frames [t0, t0 + n) of the chunk [chunk_t0, chunk_end): `head` (main-mixer events and messages due at
// the chunk's first frame: first piece only), the sub-mixers' own events inside the piece (each sub-mixer splits its block there on the
// device), call boundaries for the descendants of a mixer that splits (CMD_CALL_SPLIT), and the markers that tell a unit where its chunk /
// its parent's call ends when that is inside the main chunk but beyond this piece (CMD_CHUNK_END / CMD_CALL_END).
static void collect_piece_commands(pg_graph* g, std::vector<PgCmd>& cmds, uint64_t t0, uint64_t n, uint64_t chunk_t0, uint64_t chunk_end) {
  const uint64_t t1 = t0 + n;
  const bool first_piece = t0 == chunk_t0;
  // Nested sub-mixers: a mixer that splits its chunk at an event calls its sub-mixers once per segment (mixed.rs:679-712), so
  // every event of a mixer with sub-mixers is also a call boundary for all its descendants.
  if (g->levels.size() > 1) {
    for (size_t m = 1; m < g->mixers.size(); ++m) {
      if (g->mixers[m].children.empty() || g->mixers[m].events.empty()) continue;
      std::vector<int> desc(g->mixers[m].children);
      for (size_t i = 0; i < desc.size(); ++i) for (int c : g->mixers[desc[i]].children) desc.push_back(c);
      uint64_t next_after = UINT64_MAX;
      for (const Event& e : g->mixers[m].events) {
        if (e.sample_time >= t1) { next_after = e.sample_time; break; }
        if (first_piece ? e.sample_time <= t0 : e.sample_time < t0) continue;  // (the chunk's first frame is a call start anyway)
        for (int d : desc) {
          PgCmd c;
          memset(&c, 0, sizeof c);
          c.type = CMD_CALL_SPLIT; c.unit = g->mixers[d].unit_slot; c.frame = (uint32_t)(e.sample_time - t0);
          cmds.push_back(c);
        }
      }
      if (next_after < chunk_end) for (int d : desc) {
        PgCmd c;
        memset(&c, 0, sizeof c);
        c.type = CMD_CALL_END; c.unit = g->mixers[d].unit_slot; c.frame = (uint32_t)n; c.value64 = next_after;
        cmds.push_back(c);
      }
    }
  }
  // sub-mixer events inside [t0, t1): each sub-mixer splits its own block on the device
  for (size_t m = 1; m < g->mixers.size(); ++m) {
    HostMixer& mx = g->mixers[m];
    while (!mx.events.empty() && mx.events.front().sample_time < t1) {
      PgCmd c = mx.events.front().cmd;
      uint64_t t = mx.events.front().sample_time;
      c.frame = t <= t0 ? 0u : (uint32_t)(t - t0);
      c.unit = mx.unit_slot;
      cmds.push_back(c);
      mx.events.erase(mx.events.begin());
    }
    if (!mx.events.empty() && mx.events.front().sample_time < chunk_end && !mx.removed) {
      PgCmd c;
      memset(&c, 0, sizeof c);
      c.type = CMD_CHUNK_END; c.unit = mx.unit_slot; c.frame = (uint32_t)n; c.value64 = mx.events.front().sample_time;
      cmds.push_back(c);
    }
  }
  // everything is sorted by (unit, frame), stable (markers carry the launch's frame count: last)
  std::stable_sort(cmds.begin(), cmds.end(), [](const PgCmd& a, const PgCmd& b) { return a.unit != b.unit ? a.unit < b.unit : a.frame < b.frame; });
  // SetSourceVolume / SetSourcePanning events reach the source through a ONE-slot queue that the mixer force_pushes into (mixed.rs:810-845,
  // amplified.rs:33-35): of several such events that come due in front of the same chunk only the last one is still there when the source runs —
  // and that matters, an exponential smoother snaps to a target that is close enough (smoothing.rs:221-226), so applying the earlier ones too
  // can leave another `current` behind. (Speed and seek messages travel through the file's 128-slot queue and all arrive.)
  for (size_t i = 0; i < cmds.size(); ++i) {
    if (cmds[i].type != CMD_VOICE_VOLUME && cmds[i].type != CMD_VOICE_PAN) continue;
    for (size_t j = i + 1; j < cmds.size() && cmds[j].unit == cmds[i].unit && cmds[j].frame == cmds[i].frame; ++j)
      if (cmds[j].type == cmds[i].type && cmds[j].target == cmds[i].target) { cmds[i].type = CMD_NOP; break; }
  }
}

// Renders frames [pos, pos + n_samples / 2) of the write call into d_out. The call is walked in the reference's chunks — min(remaining,
// PG_MAX_FRAMES) frames from the call's start and from every main-mixer event (mixed.rs:679-712) — whatever max_frames is: a chunk is rendered
// as pieces of at most max_frames frames, every per-chunk decision taken once per chunk on the device (pg_dev.h: PG_MAX_FRAMES).
// cap_frames > 0: stop in front of the first chunk that would end beyond cap_frames (the caller's staging holds that much; >= PG_MAX_FRAMES)
// and return what was rendered; the caller goes on with begin = false. Returns the samples rendered (0: nothing to do, or the graph failed).
size_t graph_write_impl(pg_graph* g, float* d_out, size_t n_samples, uint64_t pos, hipStream_t stream, bool begin, size_t cap_frames) {
  if (g->failed) return 0;
  // a consistency flag the kernels mirrored into the mapped feedback block (a unit left unrendered inside a super-block launch): what was
  // handed out since is not trustworthy — the graph goes silent like a GuardedSource whose source panicked (src/source/guarded.rs:87-107)
  if (g->h_feedback && *(volatile unsigned long long*)(g->h_feedback + 2) != 0) {
    g->failed = true;
    set_error(PG_ERR_DEVICE, "kernel consistency flag %llu raised (see pg_graph_device_errors): the graph is disabled", *(volatile unsigned long long*)(g->h_feedback + 2));
    return 0;
  }
  if (n_samples % 2 != 0) { set_error(PG_ERR_PARAMETER, "n_samples must be a multiple of the channel count"); return 0; }
  (void)hipSetDevice(g->device);
  // a caller that moves from one stream to another without a graph mutation in between: the tables and rings are ordered per stream
  if (g->last_stream && g->last_stream != stream) { if (pg_stream_sync(g->last_stream) != hipSuccess) { g->failed = true; return 0; } g->cmds_since_sync = 0; }
  g->last_stream = stream;
  if (begin) { graph_begin_write(g, pos); g->call_end = pos + n_samples / 2; }
  if (g->topo_dirty) { g->rows_free_fresh = false; if (rebuild_topology(g, stream)) { g->failed = true; return 0; } }
  if (!g->stream_voices.empty()) { g->rows_free_fresh = false; if (flush_stream_feeds(g, stream)) { g->failed = true; return 0; } }
  if (g->overlap_stream != stream) { g->rows_free_fresh = false; g->overlap_stream = stream; }
  g->write_done_attached = false;
  g->audible_valid = false;   // (a write that finds nothing to do launches nothing: the words an earlier call left are not this call's)
  if (graph_is_empty(g)) return 0;
  g->audible_valid = true;
  const uint64_t frames = n_samples / 2, mf = g->max_frames, CH = PG_MAX_FRAMES;
  uint64_t done = 0;
  auto fail = [&]() -> size_t { g->failed = true; return 0; };
  // Deferred bus chain: the call leaves ONE `audible` word per piece it emits, in order (a running cursor — a chunk that an event cut short
  // has pieces of its own, and max_frames need not divide a chunk), and the offsets at which an event restarted the chunk grid:
  // process_bus_impl walks the same grid with the same cursor (round-4 advisor finding: the words used to sit at done / max_frames).
  uint64_t word_cursor = 0;
  int* const aud_words = (g->defer_bus && g->d_audible_out) ? g->d_audible_out : g->d_audible;   // where the mixer sum leaves the `audible` words
  if (g->defer_bus) { g->defer_cuts.clear(); g->defer_pos = pos; g->defer_words = 0; }
  while (done < frames) {
    const uint64_t now = pos + done;
    HostMixer& main = g->mixers[0];
    if (cap_frames && done > 0) {  // would the next chunk still fit the caller's staging? (decided before its events are taken out of the queue)
      uint64_t next_n = std::min<uint64_t>(frames - done, CH);
      for (const Event& e : main.events) if (e.sample_time > now) { next_n = std::min<uint64_t>(next_n, e.sample_time - now); break; }
      if (done + next_n > cap_frames) break;
    }
    // main-mixer events due now apply at frame 0 of the chunk; the next main event bounds it (:679-693)
    std::vector<PgCmd> head;
    while (!main.events.empty() && main.events.front().sample_time <= now) {
      PgCmd c = main.events.front().cmd;
      if (g->defer_bus && (c.type == CMD_FX_PARAM || c.type == CMD_FX_RESET || c.type == CMD_NOP)) {  // the bus chain runs in pg_graph_process_bus_device
        if (main.bus_events.size() >= 65536) main.bus_events.erase(main.bus_events.begin(), main.bus_events.begin() + 32768);  // a shard that never runs the bus
        main.bus_events.push_back(main.events.front());
        main.events.erase(main.events.begin());
        continue;
      }
      c.frame = 0;
      c.unit = (c.type == CMD_FX_PARAM || c.type == CMD_FX_RESET || c.type == CMD_NOP) ? main.unit_slot : g->source_unit_of_voice[c.param];  // voice commands carry the voice id in `param`
      head.push_back(c);
      main.events.erase(main.events.begin());
    }
    // the regular chunk grid holds from here to the end of the call or the next main-mixer event
    uint64_t span = frames - done;
    if (!main.events.empty()) span = std::min<uint64_t>(span, main.events.front().sample_time - now);
    if (span == 0) continue;
    const bool span_ends_at_event = span < frames - done;   // the grid restarts behind this span
    uint64_t chunk_n = std::min<uint64_t>(span, CH);
    // StopSource messages: processed by process_messages at the start of write (:294-499)
    if (g->messages_due) {
      for (size_t m = 0; m < g->mixers.size(); ++m) {
        for (PgCmd c : g->mixers[m].messages) {
          c.frame = 0;
          c.unit = m == 0 ? g->source_unit_of_voice[c.param] : g->mixers[m].unit_slot;
          head.push_back(c);
        }
        g->mixers[m].messages.clear();
      }
      g->messages_due = false;
    }
    // ---- super-block: whole chunks without a command anywhere in them, in steady state, as ONE launch sequence (every workgroup of the staged /
    // fast kernels walks the blocks of its unit; per-chunk decisions are still taken per chunk, on the device) ----
    uint64_t k = 0;
    if (head.empty() && span >= mf && graph_super_ok(g)) {
      uint64_t kmax = std::min<uint64_t>(span / mf, g->max_blocks);
      if (cap_frames) kmax = std::min<uint64_t>(kmax, (cap_frames - std::min<uint64_t>(done, cap_frames)) / mf);
      uint64_t t_next = UINT64_MAX;
      for (const HostMixer& mx : g->mixers) if (!mx.events.empty()) t_next = std::min(t_next, mx.events.front().sample_time);
      if (t_next != UINT64_MAX && t_next < now + kmax * mf) kmax = t_next > now ? (t_next - now) / mf : 0;
      // the launch must end where a chunk ends: at a multiple of the chunk length or at the end of the span
      if (g->defer_bus) kmax = std::min<uint64_t>(kmax, g->audible_slots - std::min<uint64_t>(word_cursor, g->audible_slots));   // (one word per block)
      if (CH % mf == 0) k = (kmax * mf >= span && span % mf == 0) ? span / mf : (kmax / (CH / mf)) * (CH / mf);
      else k = (span % mf == 0 && span <= CH && span / mf <= kmax) ? span / mf : 0;
    }
    // A small unit level in front of a bus chain (BASELINE configs 2 and 4: 64 / 256 voices, then a chain that takes four times as long as they
    // do): launch sequences of a few chunks each, so that the chain of one runs over the unit kernels of the next INSIDE a call too
    // (the first sequence behind a serialisation has no chain to run under: it is one chunk, so that only that chunk's unit kernels are exposed)
    if (CH % mf == 0 && g->overlap_bus && !g->defer_bus && !g->mixers[0].fx.empty() && g->n_graph_units <= 512) {
      const uint64_t per_chunk = CH / mf;
      // (a chain to run under: one was enqueued by the sequence before — in this call, or in the call before and the stream has not drained
      // since: a caller that waited for its last call finds an idle device)
      const bool under_chain = g->rows_free_fresh && (done > 0 || hipStreamQuery(stream) == hipErrorNotReady);
      const uint64_t group = under_chain ? std::max<uint64_t>(per_chunk, (g->bus_group / per_chunk) * per_chunk) : std::max<uint64_t>(per_chunk, 2);
      if (k > group) k = group;
    }
    if (k > 0) {
      LaunchSpan sp;
      sp.n = (uint32_t)mf; sp.t0 = now; sp.n_chunks = (int)k; sp.row_block = 0; sp.grid_off = 0; sp.grid_span = (uint32_t)std::min<uint64_t>(span, 0x7fffffffull);
      sp.round = g->launch_counter++;
      sp.generic_idle = true;  // (graph_super_ok: steady state, one level)
      sp.timed = g->timing_period > 0 && (g->launch_counter % (uint64_t)g->timing_period) == 0 && g->ev_used < g->ev_pool.size() && g->n_graph_units > 0;
      // A bus chain behind the sum: this sequence's unit kernels go to the unit stream, UNDER the bus chain of the sequence before (they wait for
      // its sum to have read the per-unit rows, not for its chain), and this sequence's sum waits for them
      const bool overlap = g->overlap_bus && !g->defer_bus && !g->mixers[0].fx.empty() && k > 1;
      if (overlap) {
        if (!g->rows_free_fresh) HIP_TRY_FAIL(hipEventRecord(g->ev_rows_free, stream));   // (behind whatever the write's stream holds: one serialisation, then the pipeline runs)
        HIP_TRY_FAIL(hipStreamWaitEvent(g->unit_stream, g->ev_rows_free, 0));
        if (launch_level(g, 0, sp, g->unit_stream)) return fail();
        HIP_TRY_FAIL(hipEventRecord(g->ev_units_done, g->unit_stream));
        HIP_TRY_FAIL(hipStreamWaitEvent(stream, g->ev_units_done, 0));
      } else {
        if (launch_level(g, 0, sp, stream)) return fail();
        g->rows_free_fresh = false;
      }
      // where the blocks' `audible` words go: with the bus chain deferred to the caller, word c of the call belongs to its c-th block
      const int slot = g->defer_bus ? (int)word_cursor : 0;
      if (g->defer_bus) { word_cursor += k; g->defer_words = (int)word_cursor; }
      {
        const Level& top = g->levels.back();
        HIP_TRY_FAIL(pg_launch_mix(g->d_unit_out + (size_t)top.off * g->stride, g->stride, top.cnt, g->d_partial, d_out + done * 2, (uint32_t)mf * 2, g->d_audible_tab + top.off, g->unit_out_rows,
                                   aud_words + slot, stream, (int)k, (size_t)g->unit_out_rows * g->stride, (g->write_done_event && done + k * mf == frames) ? g->write_done_event : nullptr));
        if (g->write_done_event && done + k * mf == frames) g->write_done_attached = true;
      }
      if (overlap) { HIP_TRY_FAIL(hipEventRecord(g->ev_rows_free, stream)); g->rows_free_fresh = true; }
      if (!g->defer_bus && launch_bus(g, d_out + done * 2, sp, slot, stream)) return fail();
      done += k * mf;
      if (g->defer_bus && span_ends_at_event && k * mf == span && done < frames) g->defer_cuts.push_back(done);
      continue;
    }
    // ---- one chunk, piece by piece ----
    g->rows_free_fresh = false;
    // A sub-mixer keeps one result bit per call of its parent: the chunk is bounded so that at most PG_MAX_CALLS - 1 call boundaries fall inside it
    if (g->levels.size() > 1) {
      std::vector<uint64_t> cuts;
      for (size_t m = 1; m < g->mixers.size(); ++m) {
        if (g->mixers[m].children.empty()) continue;
        for (const Event& e : g->mixers[m].events) { if (e.sample_time >= now + chunk_n) break; if (e.sample_time > now) cuts.push_back(e.sample_time); }
      }
      std::sort(cuts.begin(), cuts.end());
      cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
      if (cuts.size() > (size_t)(PG_MAX_CALLS - 1)) chunk_n = cuts[PG_MAX_CALLS - 1] - now;
    }
    const uint64_t n_pieces = (chunk_n + mf - 1) / mf, n_full = chunk_n / mf;
    if (g->defer_bus && word_cursor + n_pieces > g->audible_slots) {
      // the call has used up its `audible` words (audible_slots = max(64, pieces of one chunk): only a call longer than 64 pieces, or one cut by
      // many events, gets here): what was rendered is returned, the caller goes on with another call — silent aliasing of words is not an option
      set_error(PG_ERR_PARAMETER, "a deferred-bus write leaves at most %zu `audible` words: %llu frames rendered, call again for the rest", g->audible_slots, (unsigned long long)done);
      break;
    }
    if (n_pieces > g->unit_out_blocks) { set_error(PG_ERR_STATE, "per-unit buffers were not reserved for a chunk's pieces"); return fail(); }
    std::vector<LaunchSpan> spans((size_t)n_pieces);
    const uint64_t round0 = g->launch_counter;
    g->launch_counter += n_pieces;
    const bool steady_now = g->levels.size() == 1 && graph_steady(g);
    const size_t tl = timed_level_of(g);
    for (uint64_t p = 0; p < n_pieces; ++p) {
      LaunchSpan& sp = spans[(size_t)p];
      sp.t0 = now + p * mf; sp.n = (uint32_t)std::min<uint64_t>(mf, chunk_n - p * mf);
      sp.n_chunks = 1; sp.row_block = (int)p; sp.grid_off = (uint32_t)(p * mf); sp.grid_span = (uint32_t)chunk_n;
      sp.round = round0 + p;
      std::vector<PgCmd> cmds;
      if (p == 0) cmds = head;
      collect_piece_commands(g, cmds, sp.t0, sp.n, now, now + chunk_n);
      if (!cmds.empty()) {
        // (a round that will run its decision-scan kernel takes a short list there as the kernel's argument: no copy kernel on the stream)
        sp.n_cmds = (int)cmds.size();
        const bool by_scan = cmds.size() <= PG_CMD_PACK && level_concurrent(g, 0, sp, stream);
        if (stage_commands(g, cmds, stream, &sp.d_cmds, by_scan ? &sp.h_cmds : nullptr)) return fail();
        // Which commands can leave the steady state behind them? A parameter command may start a smoother of an effect (the device knows how
        // long: the next rounds' scans tell), a speed command a glide, markers belong to split chunks. Source volume / panning / stop / seek
        // change nothing the time-parallel kernels do not render (AmplifiedSource / PannedSource smoothers, the fader, a new position): a unit
        // deferred for those alone is back on its kernel in the next block, so the host need not wait for the device to say so — offline
        // calls keep their super-block launches between such commands (notes that stop and start: bench.py --workload dyn --churn).
        bool may_ramp = false;
        for (const PgCmd& c : cmds) may_ramp |= !(c.type == CMD_VOICE_VOLUME || c.type == CMD_VOICE_PAN || c.type == CMD_VOICE_STOP || c.type == CMD_VOICE_SEEK);
        // (the round AFTER this one is the first whose scan sees what the commands left behind: this round's own count of state-deferred units
        // was taken in front of them)
        if (may_ramp) g->last_change_round = sp.round + 1;
      }
      // Steady state: the generic kernel of an earlier round (not older than the last topology change / command / mode switch) found
      // nothing deferred, and units leave the steady state only through those host-visible events -> the generic launch is skipped.
      // (Graphs with nested sub-mixers always launch it: the parents are rendered there.)
      sp.generic_idle = steady_now && cmds.empty() && g->last_change_round < round0;
    }
    for (size_t li = 0; li < g->levels.size(); ++li) {
      for (uint64_t p = 0; p < n_pieces; ++p) {
        LaunchSpan sp = spans[(size_t)p];
        // the event pair costs ~8 us of stream time per round (also when it rides on the dispatch): callers that only need the
        // average can time every n-th round (pg_graph_set_timing_period)
        sp.timed = li == tl && g->timing_period > 0 && ((sp.round + 1) % (uint64_t)g->timing_period) == 0 && g->ev_used < g->ev_pool.size() && g->n_graph_units > 0;
        if (launch_level(g, li, sp, stream)) return fail();
      }
    }
    // the chunk's sum and bus chain: its full pieces in one launch each, a shorter last piece in launches of its own
    const int slot0 = g->defer_bus ? (int)word_cursor : 0;
    if (g->defer_bus) { word_cursor += n_pieces; g->defer_words = (int)word_cursor; }
    const size_t chunk_stride = (size_t)g->unit_out_rows * g->stride;
    const Level& top = g->levels.back();
    float* dst = d_out + done * 2;
    const bool last_chunk_of_call = g->write_done_event != nullptr && done + chunk_n == frames;   // (the event rides on the call's last sum launch)
    if (last_chunk_of_call) g->write_done_attached = true;
    if (n_full > 0) HIP_TRY_FAIL(pg_launch_mix(g->d_unit_out + (size_t)top.off * g->stride, g->stride, top.cnt, g->d_partial, dst, (uint32_t)mf * 2, g->d_audible_tab + top.off, g->unit_out_rows,
                                               aud_words + slot0, stream, (int)n_full, chunk_stride, (last_chunk_of_call && n_pieces == n_full) ? g->write_done_event : nullptr));
    if (n_pieces > n_full) {
      const LaunchSpan& r = spans[(size_t)n_full];
      HIP_TRY_FAIL(pg_launch_mix(g->d_unit_out + (size_t)n_full * chunk_stride + (size_t)top.off * g->stride, g->stride, top.cnt, g->d_partial, dst + n_full * mf * 2, r.n * 2,
                                 g->d_audible_tab + (size_t)n_full * g->unit_out_rows + top.off, g->unit_out_rows, aud_words + slot0 + (int)n_full, stream, 1, chunk_stride, last_chunk_of_call ? g->write_done_event : nullptr));
    }
    if (!g->defer_bus && !g->mixers[0].fx.empty()) {
      // the bus unit's commands of the chunk's first piece (main-mixer effect events) ride on its bus launch
      const bool bus_timed = g->timing_period > 0 && ((spans[0].round + 1) % (uint64_t)g->timing_period) == 0;
      if (n_full > 0) {
        LaunchSpan b = spans[0];
        b.timed = bus_timed;
        b.n_chunks = (int)n_full;
        if (launch_bus(g, dst, b, slot0, stream)) return fail();
      }
      if (n_pieces > n_full) {
        LaunchSpan b = spans[(size_t)n_full];
        b.timed = bus_timed;
        if (n_full > 0) { b.d_cmds = nullptr; b.n_cmds = 0; }
        if (launch_bus(g, dst + n_full * mf * 2, b, slot0 + (int)n_full, stream)) return fail();
      }
    }
    done += chunk_n;
    if (g->defer_bus && span_ends_at_event && chunk_n == span && done < frames) g->defer_cuts.push_back(done);
  }
  return (size_t)done * 2;
}

extern "C" {

size_t pg_graph_write_device(pg_graph* g, float* d_out, size_t n_samples, uint64_t pos_in_frames, void* hip_stream) {
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : g->stream;
  size_t w = graph_write_impl(g, d_out, n_samples, pos_in_frames, s);
  if (!hip_stream && w) { if (pg_stream_sync(g->stream) != hipSuccess) { g->failed = true; return 0; } g->cmds_since_sync = 0; }
  return w;
}

size_t pg_graph_write(pg_graph* g, float* out, size_t n_samples, uint64_t pos_in_frames) {
  if (g->failed) return 0;
  if (n_samples % 2 != 0) { set_error(PG_ERR_PARAMETER, "n_samples must be a multiple of the channel count"); return 0; }
  (void)hipSetDevice(g->device);
  // ONE write call of the main mixer whatever its length (process_messages once, one call end): the staging bus holds whole chunks — at least
  // PG_MAX_FRAMES frames — and the call is rendered in spans of as many chunks as fit, each copied out behind its last launch
  graph_begin_write(g, pos_in_frames);
  g->call_end = pos_in_frames + n_samples / 2;
  const size_t cap_frames = g->bus_frames;
  size_t off = 0, total = 0;
  uint64_t pos = pos_in_frames;
  while (off < n_samples) {
    const size_t w = graph_write_impl(g, g->d_bus, n_samples - off, pos, g->stream, false, cap_frames);
    if (w == 0) {  // nothing to do (from here on): silence for the rest of a call that has produced output, 0 for one that has not
      if (g->failed || off == 0) return 0;
      memset(out + off, 0, (n_samples - off) * sizeof(float));
      break;
    }
    // device feedback: how many main-mixer sources are still alive (transient sources are dropped when exhausted, :715)
    if (graph_enqueue_status(g, g->stream) != PG_OK || hipMemcpyAsync(g->h_pinned, g->d_bus, w * sizeof(float), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        pg_stream_sync(g->stream) != hipSuccess) {
      g->failed = true;
      set_error(PG_ERR_DEVICE, "device failure in write: %s", hipGetErrorString(hipGetLastError()));
      return 0;
    }
    g->cmds_since_sync = 0;
    memcpy(out + off, g->h_pinned, w * sizeof(float));
    graph_collect_status(g);
    off += w; pos += w / 2; total += w;
  }
  return off >= n_samples || total > 0 ? n_samples : 0;
}

}  // extern "C"

// The main mixer's effect chain over a summed bus that the caller holds (pg_graph_process_bus_device, the sharded handle).
// `bus_audible`: device words "the summed input of block c of this call is audible" (audible_input of process_effects, mixed.rs:627-655,
// word c = frames [c * max_frames, +max_frames) of the call); nullptr = audible. Blocks without a bus event at their head are
// rendered by ONE launch that walks them in order (as launch_round does behind a super-block).
int process_bus_impl(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, hipStream_t s, int* bus_audible) {
  if (g->failed) return PG_ERR_DEVICE;
  (void)hipSetDevice(g->device);
  if (g->mixers[0].fx.empty()) return PG_OK;
  if (g->last_stream && g->last_stream != s) { HIP_TRY(pg_stream_sync(g->last_stream)); g->cmds_since_sync = 0; }
  g->last_stream = s;
  if (g->topo_dirty && rebuild_topology(g, s)) return graph_fail(g, PG_ERR_DEVICE);
  const uint64_t frames = n_samples / 2, mf = g->max_frames, CH = PG_MAX_FRAMES;
  uint64_t done = 0;
  HostMixer& main = g->mixers[0];
  // The chunk grid of the write this bus belongs to: it restarts at the bus events (queued below) AND wherever another main-mixer event — a
  // volume / panning / speed / seek event of a main-mixer source — cut the write; those offsets were recorded by this graph's own deferred write
  // of the same position (a root that renders no partial of its own, or ranks whose main-mixer sources take events the root does not see: the
  // caller cuts its calls there, pg_graph_next_main_event). Words are consumed with a running cursor, one per piece, as the write emitted them.
  const std::vector<uint64_t>* cuts = (g->defer_bus && g->defer_pos == pos_in_frames) ? &g->defer_cuts : nullptr;
  size_t cut_i = 0;
  uint64_t word_cursor = 0;
  while (done < frames) {
    const uint64_t now = pos_in_frames + done;
    // effect events of the main mixer (queued by write in defer_bus mode) cut the bus into chunks at their sample times, exactly
    // like the event loop of MixedSource::write (mixed.rs:679-712); between them the chunks are PG_MAX_FRAMES long, rendered in pieces
    std::vector<PgCmd> cmds;
    while (!main.bus_events.empty() && main.bus_events.front().sample_time <= now) {
      PgCmd c = main.bus_events.front().cmd;
      c.frame = 0; c.unit = main.unit_slot;
      cmds.push_back(c);
      main.bus_events.erase(main.bus_events.begin());
    }
    uint64_t span = frames - done;
    if (!main.bus_events.empty()) span = std::min<uint64_t>(span, main.bus_events.front().sample_time - now);
    if (cuts) {
      while (cut_i < cuts->size() && (*cuts)[cut_i] <= done) ++cut_i;
      if (cut_i < cuts->size()) span = std::min<uint64_t>(span, (*cuts)[cut_i] - done);
    }
    if (span == 0) continue;
    // whole blocks of max_frames in one launch (as many chunks as the span holds), a chunk's shorter last piece in a launch of its own
    const PgCmd* d_cmds = nullptr;
    if (!cmds.empty()) { int rc = stage_commands(g, cmds, s, &d_cmds); if (rc) return rc; }
    uint64_t off = 0;  // frames of the span rendered
    while (off < span) {
      const uint64_t chunk_start = (off / CH) * CH, chunk_end = std::min<uint64_t>(chunk_start + CH, span);
      LaunchSpan sp;
      sp.t0 = now + off; sp.grid_off = (uint32_t)off; sp.grid_span = (uint32_t)std::min<uint64_t>(span, 0x7fffffffull);
      sp.round = g->launch_counter;
      uint64_t k;
      if (chunk_end - off < mf) { sp.n = (uint32_t)(chunk_end - off); k = 1; }      // the chunk's last, shorter piece
      else {
        sp.n = (uint32_t)mf;
        // full pieces: to the end of this chunk, and on through the following whole chunks when the pieces tile them
        k = (chunk_end - off) / mf;
        if (CH % mf == 0 && (chunk_end - off) % mf == 0) k = (span - off) / mf;
        k = std::max<uint64_t>(1, std::min<uint64_t>(k, g->audible_slots));
      }
      sp.n_chunks = (int)k;
      if (off == 0 && d_cmds) { sp.d_cmds = d_cmds; sp.n_cmds = (int)cmds.size(); }
      // one word per piece, in the order the write emitted them (a chunk's flag sits in the word of its last piece)
      if (bus_audible && word_cursor + k > g->audible_slots) return set_error(PG_ERR_PARAMETER, "the bus chain ran out of `audible` words (%zu): cut the call as the write was cut", g->audible_slots);
      const uint64_t word = word_cursor;
      word_cursor += k;
      auto one = [&](const LaunchSpan& q, float* dst, int* flags) -> int {
        PgLaunch B;
        fill_launch(g, q, B);
        B.mode = 0; B.n_units = 1; B.unit_order = nullptr; B.unit_base = main.unit_slot;
        B.bus = dst; B.bus_audible = flags; B.audible_tab = nullptr; B.sched = nullptr;
        bus_pipeline_setup(g, B);
        HIP_TRY(pg_launch_units(B, s));
        return PG_OK;
      };
      float* dst = d_bus + (done + off) * 2;
      int* flags = bus_audible ? bus_audible + word : nullptr;
      if (sp.n_cmds > 0 && sp.n_chunks > 1) {  // commands ride on a launch of the first block alone
        LaunchSpan head = sp;
        head.n_chunks = 1;
        int rc = one(head, dst, flags);
        if (rc) return rc;
        LaunchSpan rest = sp;
        rest.d_cmds = nullptr; rest.n_cmds = 0; rest.t0 += sp.n; rest.grid_off += sp.n; rest.n_chunks -= 1;
        rc = one(rest, dst + (size_t)sp.n * 2, flags ? flags + 1 : nullptr);
        if (rc) return rc;
      } else {
        int rc = one(sp, dst, flags);
        if (rc) return rc;
      }
      off += (uint64_t)sp.n * k;
    }
    done += span;
  }
  return PG_OK;
}

// One process per GPU (bench.py over RCCL): the ranks' `audible` words travel WITH their partial buses — as floats behind the samples, one
// sum-reduce for both — and the root's bus chain takes its per-chunk decisions from the summed words as the one mixer would from its own.
__global__ void pg_words_to_float_kernel(const int* __restrict__ w, float* __restrict__ f, int n, int valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = (valid && w[i] != 0) ? 1.0f : 0.0f;
}
__global__ void pg_float_to_words_kernel(const float* __restrict__ f, int* __restrict__ w, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = f[i] > 0.0f ? 1 : 0;
}

extern "C" {

int pg_graph_process_bus_device(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, void* hip_stream) {
  return process_bus_impl(g, d_bus, n_samples, pos_in_frames, hip_stream ? (hipStream_t)hip_stream : g->stream, nullptr);
}
int pg_graph_audible_words(pg_graph* g) { return g->audible_valid ? g->defer_words : 0; }
uint64_t pg_graph_next_main_event(pg_graph* g, uint64_t pos_in_frames) {
  drain_control_messages_public(g);
  for (const Event& e : g->mixers[0].events) if (e.sample_time > pos_in_frames) return e.sample_time;
  return UINT64_MAX;
}
int pg_graph_export_audible(pg_graph* g, float* d_dst, int n_words, void* hip_stream) {
  if (n_words < 0 || (size_t)n_words > g->audible_slots) return set_error(PG_ERR_PARAMETER, "a write leaves at most %zu words", g->audible_slots);
  if (n_words == 0) return PG_OK;
  (void)hipSetDevice(g->device);
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : g->stream;
  hipLaunchKernelGGL(pg_words_to_float_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, g->d_audible, d_dst, n_words, g->audible_valid ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return PG_OK;
}
int pg_graph_process_bus_device_flags(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, void* hip_stream, const float* d_flags, int n_words) {
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : g->stream;
  if (!d_flags) return process_bus_impl(g, d_bus, n_samples, pos_in_frames, s, nullptr);
  const size_t fr = n_samples / 2, per_chunk = (PG_MAX_FRAMES + g->max_frames - 1) / g->max_frames;
  const size_t need = (fr / PG_MAX_FRAMES) * per_chunk + (fr % PG_MAX_FRAMES + g->max_frames - 1) / g->max_frames;   // pieces of an event-free call (events add pieces: pg_graph_audible_words)
  if (n_words < 0 || (size_t)n_words > g->audible_slots || (size_t)n_words < need) return set_error(PG_ERR_PARAMETER, "the bus chain needs one word per block of max_frames (%zu)", need);
  (void)hipSetDevice(g->device);
  hipLaunchKernelGGL(pg_float_to_words_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, d_flags, g->d_audible, n_words);
  HIP_TRY(hipGetLastError());
  return process_bus_impl(g, d_bus, n_samples, pos_in_frames, s, g->d_audible);
}

}  // extern "C"
