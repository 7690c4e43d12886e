// Internals shared by the host-side translation units of libphonic_gpu.so (pg_host.hip: the mixer graph and its write path;
// pg_fxstate.hip: effect construction and parameter descriptors; pg_effect.hip: the standalone `Effect` handle; pg_sharded.hip: the
// multi-GPU handle). Nothing here is part of the ABI (include/phonic_gpu.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/phonic_gpu.h"
#include "pg_ctrl.h"
#include "pg_dev.h"
#include "pg_dsp_dev.h"
#include "pg_params.h"

using namespace pgd;
using namespace pgh;

// ---- kernels' launchers (pg_kernels.hip) ----------------------------------------------------------------
size_t pg_fast_scratch_bytes(uint32_t kind_mask);
size_t pg_unit_lds_bytes(uint32_t n_frames, size_t scratch_bytes = 0);
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames);
size_t pg_stage_lds_bytes(int stage, uint32_t n_frames, bool wide);  // wide: the staged kernel that renders effects in front of the reverb (their state slots)
hipError_t pg_launch_units(const PgLaunch& L, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t pg_launch_defer_scan(const PgLaunch& L, hipStream_t stream, hipEvent_t done = nullptr, const PgCmd* h_cmds = nullptr);
hipError_t pg_launch_stages(const PgLaunch& L, hipStream_t stream, int single_launch, int lean, int wide, int adapt, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, hipEvent_t tail_done = nullptr);
hipError_t pg_launch_mix(const float* unit_out, uint32_t stride, int n_units, float* partial, float* bus, uint32_t n_samples, const int32_t* audible_tab,
                         size_t audible_stride, int* audible_out, hipStream_t stream, int n_chunks = 1, size_t chunk_stride = 0, hipEvent_t done = nullptr);

// ---- errors ---------------------------------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);  // records the thread's last error message (pg_last_error_message) and returns `code`
#define PG_AUDIBLE_SLOTS 64  // >= the largest pg_graph_set_max_blocks_per_launch
#define PG_CMD_RING 65536   // commands in flight between two points at which the host knows the stream drained (2 MB device + 2 MB pinned)
#define PG_CTRL_RING 65536  // control messages waiting for the next write (the reference: 4096 per mixer; here one ring per graph)
#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) return set_error(PG_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// ---- device memory helpers ------------------------------------------------------------------------------
// Every allocation, release and host-blocking HIP call of the library goes through these wrappers and is counted
// (pg_debug_hip_calls): the reference wraps its audio callback in assert_no_alloc (src/output/cpal.rs:712-715); the test-suite
// checks the same property here — a write() on a built graph allocates nothing, frees nothing and (on a caller's stream) never blocks.
hipError_t pg_malloc(void** p, size_t bytes);
hipError_t pg_host_malloc(void** p, size_t bytes, unsigned flags);
hipError_t pg_free(void* p);
hipError_t pg_host_free(void* p);
hipError_t pg_stream_sync(hipStream_t s);
hipError_t pg_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t pg_memset(void* d, int v, size_t n);

// Device array that grows by reallocation + device-to-device copy (device-evolved state survives). New elements are collected in a
// small pinned staging block and travel in batches: a full block is flushed by the (non real-time) call that filled it, the rest by
// flush_async() on the render stream at the next write — one copy per <= STAGE elements instead of one per element.
template <class T>
struct DeviceVec {
  static constexpr size_t STAGE = 256;
  T* d = nullptr;
  size_t n = 0, cap = 0;      // n counts staged elements too
  T* h_stage = nullptr;       // pinned, STAGE elements
  size_t n_staged = 0;        // elements [n - n_staged, n) wait in h_stage
  int reserve(size_t want) {  // allocates: graph construction only
    if (want <= cap) return PG_OK;
    size_t ncap = std::max<size_t>(want, cap ? cap * 2 : 16);
    T* nd = nullptr;
    HIP_TRY(pg_malloc((void**)&nd, ncap * sizeof(T)));
    const size_t on_device = n - n_staged;
    if (d && on_device) HIP_TRY(pg_memcpy(nd, d, on_device * sizeof(T), hipMemcpyDeviceToDevice));
    if (d) (void)pg_free(d);
    d = nd;
    cap = ncap;
    return PG_OK;
  }
  int flush() {  // blocking (graph construction)
    if (n_staged) HIP_TRY(pg_memcpy(d + (n - n_staged), h_stage, n_staged * sizeof(T), hipMemcpyHostToDevice));
    n_staged = 0;
    return PG_OK;
  }
  int flush_async(hipStream_t s) {  // from write(): pinned source, no allocation, no wait; the staging block is not touched again before the
    if (n_staged) HIP_TRY(hipMemcpyAsync(d + (n - n_staged), h_stage, n_staged * sizeof(T), hipMemcpyHostToDevice, s));  // next mutation, which drains the stream first
    n_staged = 0;
    return PG_OK;
  }
  int push(const T& v, int* index) {
    int rc = reserve(n + 1);
    if (rc) return rc;
    if (!h_stage) HIP_TRY(pg_host_malloc((void**)&h_stage, STAGE * sizeof(T), hipHostMallocDefault));
    if (n_staged == STAGE && (rc = flush())) return rc;
    h_stage[n_staged++] = v;
    *index = (int)n++;
    return PG_OK;
  }
  void release() { if (d) (void)pg_free(d); if (h_stage) (void)pg_host_free(h_stage); d = nullptr; h_stage = nullptr; n = cap = n_staged = 0; }
};

// Table rebuilt from the host mirror at every topology change: capacity is reserved by the mutating calls (reserve: may allocate),
// the contents travel with ONE asynchronous copy from pinned staging inside write (upload_async: never allocates).
// A rebuild can also happen inside write WITHOUT a mutating call (and its wait) in front of it — a control message switched a Gain's DC
// filter on, stop_all_voices dropped sources (round-2 advisor finding) — so an upload may follow the previous one while that one's
// asynchronous copy has not run yet: the pinned staging alternates between two halves, and a half is only rewritten once the event
// recorded behind its last copy has passed (waited for in the rare case it has not).
template <class T>
struct DeviceTable {
  T* d = nullptr;
  T* h = nullptr;  // pinned staging: two halves of `cap` elements
  size_t n = 0, cap = 0;
  int turn = 0;
  hipEvent_t copied[2] = {nullptr, nullptr};
  bool in_flight[2] = {false, false};
  int reserve(size_t want) {
    if (want <= cap) return PG_OK;
    size_t ncap = std::max<size_t>(want, cap ? cap * 2 : 64);
    if (d) (void)pg_free(d);
    if (h) (void)pg_host_free(h);
    d = nullptr; h = nullptr; cap = 0;
    HIP_TRY(pg_malloc((void**)&d, ncap * sizeof(T)));
    HIP_TRY(pg_host_malloc((void**)&h, 2 * ncap * sizeof(T), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) { if (!copied[i]) HIP_TRY(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming)); in_flight[i] = false; }  // (the mutating call drained the streams)
    cap = ncap;
    return PG_OK;
  }
  int upload_async(const std::vector<T>& v, hipStream_t s) {
    if (v.size() > cap) return set_error(PG_ERR_STATE, "device table capacity was not reserved by the mutating call");
    if (!v.empty()) {
      T* half = h + (size_t)turn * cap;
      if (in_flight[turn] && hipEventQuery(copied[turn]) != hipSuccess) HIP_TRY(hipEventSynchronize(copied[turn]));
      memcpy(half, v.data(), v.size() * sizeof(T));
      HIP_TRY(hipMemcpyAsync(d, half, v.size() * sizeof(T), hipMemcpyHostToDevice, s));
      HIP_TRY(hipEventRecord(copied[turn], s));
      in_flight[turn] = true;
      turn ^= 1;
    }
    n = v.size();
    return PG_OK;
  }
  void release() {
    if (d) (void)pg_free(d);
    if (h) (void)pg_host_free(h);
    for (int i = 0; i < 2; ++i) { if (copied[i]) (void)hipEventDestroy(copied[i]); copied[i] = nullptr; in_flight[i] = false; }
    d = nullptr; h = nullptr; n = cap = 0;
  }
};

static inline size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

// ---- effect instance: host mirror + construction of the device state (pg_fxstate.hip) -----------------------
struct HostFx {
  int kind = 0;
  std::vector<float> init_raw;   // raw value per parameter after `new()/with_parameters`
  std::vector<float> target;     // shadow of the targets (for nothing on the hot path; introspection only)
  bool with_params = false;
  bool has_seeds = false;
  uint32_t fpd_l = 16386, fpd_r = 16386;
  double vib[16] = {0};
  bool has_lfo_seed = false;     // Delay: explicit Xoshiro256++ state of the LFO's random shapes (pg_effect_init::lfo_rng_state)
  uint64_t lfo_rng[4] = {0, 0, 0, 0};
  void* d_mem = nullptr;         // delay-line memory owned by this effect
  size_t d_mem_bytes = 0;
  int last_mixer = -1;           // graph effects: the mixer the effect belonged to when it was removed (its late events stay that mixer's events)
};

PgSmooth make_smooth(const ParamSpec& p, float value, uint32_t sr);
int host_fx_from_init(int kind, const pg_effect_init* init, HostFx& h);
// State of the effect right after `Effect::initialize(sample_rate, 2, max_frames)`.
int build_fx_device_state(HostFx& h, uint32_t sr, int device, bool standalone, PgFx& fx);
// PgCmd::value64 of a parameter update (time-constant coefficients computed with the host's expf)
uint64_t fx_param_aux(int kind, int param, float raw, uint32_t sr);

// ---- the graph --------------------------------------------------------------------------------------------
struct Event {  // MixerEvent (src/source/mixed.rs:47-109) resolved to a device command
  uint64_t sample_time;
  uint64_t seq;
  PgCmd cmd;  // unit/frame filled per launch
  int mixer;  // owning mixer (0 = main)
};

struct HostVoice {
  int mixer; int dev_index; uint64_t start_time; void* d_pcm; void* d_stage; bool outer;
  // host-fed source (pg_graph_add_stream_voice): pinned ring + word the feeds are staged in, device ring = d_pcm
  bool transient = true;           // PlayingSource::is_transient (pg_voice_options::non_transient == 0)
  bool stream = false, ended = false, ended_sent = false;
  float* h_ring = nullptr;         // pinned: [cap_frames * channels] floats
  uint32_t channels = 0;
  size_t cap_frames = 0;
  uint64_t fed = 0, sent = 0;      // frames accepted from the host / frames whose copy to the device ring has been enqueued
  uint64_t consumed_known = 0;     // frames the device is known to have read (pg_graph_stream_voice_consumed): bounds what may be overwritten
};
struct HostMixer {
  int unit_slot = -1;              // sub-mixer unit; for the main mixer: the bus unit
  std::vector<int> voices;         // voice ids in playing order (sorted by start time, insert-before-equal)
  std::vector<int> fx;             // effect ids in chain order
  std::vector<Event> events;       // sorted by sample_time (stable: insert after equal, event.rs:31-38)
  std::vector<PgCmd> messages;     // StopSource messages: applied at the start of the next write
  std::vector<Event> bus_events;   // main mixer only, defer_bus mode: effect events waiting for pg_graph_process_bus_device
  int parent = 0;                  // Player::add_mixer(parent): 0 = the main mixer
  int depth = 1;                   // main mixer 0, its sub-mixers 1, their sub-mixers 2 ...
  std::vector<int> children;       // nested sub-mixers, in the order they were added
  bool removed = false;            // Player::remove_mixer: gone from its parent (with everything under it)
  bool remove_pending = false;     // MixerMessage::RemoveAllPendingEvents waiting for the next write (it needs that write's position)
  uint64_t remove_event_seq = 0;   // ... it covers the events queued before it (Event::seq below this) and the sources added before it
  size_t remove_voice_limit = 0;   //     (voice ids below this): messages are processed in order (mixed.rs:294-313), what arrives later stays
};
// Launch level: the units of one depth of the mixer tree. A mixer reads its sub-mixers' output rows, so the levels are launched
// deepest first, in stream order; the sub-mixers of the main mixer and its sources form the last level (summed by the mix kernels).
struct Level { int off = 0, cnt = 0, n_staged = 0, n_staged_wide = 0, n_staged_adapt = 0, n_static_defer = 0; };

struct pg_graph {
  int device = 0;
  uint32_t sample_rate = 48000, channels = 2;
  size_t max_frames = 4096;
  hipStream_t stream = nullptr;
  bool failed = false;  // sticky: GuardedSource semantics
  int fast = 1;
  bool wide = false;  // some sub-mixer chain holds Filter / Eq5 / Distortion: use the wide fast-kernel variant
  uint64_t call_end = 0;   // end position of the write call being rendered (the sharded handle sets it for the whole write, across its segments)
  bool any_outer = false;  // some voice sits behind a ResampledSource: the four-per-CU fast kernel does not carry that code (sticky, like `wide`)
  uint32_t fast_kind_mask = 0;  // effect kinds held by the units the fast kernels render: sizes their LDS arena (pg_fast_scratch_bytes)
  int timing_period = 0;   // time every n-th round with a hipEvent pair (0: never, the default); pg_graph_set_timing_period creates the pairs
  int staged_mode = 1;     // [Gain|Panning]* -> Reverb units: 1 = staged single launch (pg_stage_fused_kernel), 2 = one launch per stage, 0 = fused fast kernel
  int n_staged = 0;        // graph units eligible for the staged pipeline (levels 1 and 2)
  int n_staged_wide = 0;   // ... of level 2 (leading effects beyond Gain / Panning)
  int n_staged_adapt = 0;  // ... of level 3 (a voice behind a ResampledSource or a host-fed one)
  int n_static_defer = 0;  // graph units that always run on the generic kernel
  double* d_stage = nullptr;  // [stage_rows][PG_STAGE_BUF_DOUBLES]
  DeviceTable<int4> d_slot_info;  // per launch slot: {unit slot, first voice, last effect, voices}
  DeviceTable<int2> d_slot_fx;    // per launch slot: {first effect, second effect} (device indices, -1: none)
  DeviceTable<int4> d_slot_lead;  // per launch slot, staged units: the first three effects in front of the reverb (device indices, -1: none)
  DeviceTable<int2> d_child_rows; // nested sub-mixers: {output row, unit slot}, indexed by PgUnit::child_off
  DeviceTable<PgUnit> d_topo;     // topology fields of every unit, patched into d_units by pg_patch_units_kernel
  std::vector<Level> levels;    // deepest first
  uint64_t defer_phase = 0;     // one deferral hand-shake per level launch (two counters, alternating)
  int32_t* d_defer = nullptr;  // [2 counters][2 state counters][defer_rows slots]: compact list of the units the fast kernels deferred
  bool defer_dirty = false;    // a round with a generic launch left a count in its counter: the first round that skips the launch clears all four words
  size_t defer_rows = 0;
  size_t stage_rows = 0;
  bool defer_bus = false;
  size_t max_blocks = 1;        // blocks of max_frames one launch sequence may render (pg_graph_set_max_blocks_per_launch); sizes d_unit_out
  size_t unit_out_blocks = 0;   // per-unit output tables as allocated: max(max_blocks, pieces of a chunk)
  size_t bus_frames = 0;        // frames the staging of pg_graph_write holds (whole chunks: >= PG_MAX_FRAMES)
  size_t audible_slots = 0;     // words of d_audible: one per block of a launch sequence / piece of a chunk
  bool audible_valid = false;   // the last write rendered (its `audible` words are in d_audible)
  // the last deferred-bus write: its position, the words it left (one per piece, in order) and the offsets at which a main-mixer event
  // restarted its chunk grid — pg_graph_process_bus_device of the same position walks the same grid
  uint64_t defer_pos = UINT64_MAX;
  int defer_words = 0;
  std::vector<uint64_t> defer_cuts;
  // Graphs with a bus chain, super-block launches: the unit kernels of launch sequence s + 1 run on a stream of their own UNDER the bus chain of
  // sequence s (one workgroup per effect: the machine is all but idle while it walks the blocks). They need the sum of sequence s to have read
  // the per-unit rows (ev_rows_free, recorded behind the mix launch, in front of the bus launch); the sum of s + 1 needs them done (ev_units_done).
  hipStream_t unit_stream = nullptr;
  hipEvent_t ev_units_done = nullptr, ev_rows_free = nullptr;
  hipStream_t overlap_stream = nullptr;  // the write stream ev_rows_free was last recorded on
  bool rows_free_fresh = false;  // ev_rows_free was recorded behind everything the caller's stream holds that the next unit launch must follow.
                                 // INVARIANT: whatever the library enqueues on a write's stream that a unit kernel reads — topology uploads, command
                                 // lists (stage_commands), stream feeds, a change of stream — must clear this flag (all of them do: rebuild_topology,
                                 // the piece path, flush_stream_feeds, graph_quiesce); the flag persists across calls on purpose, so that a call's
                                 // unit kernels run under the bus chain of the call before. A caller's own work on that stream never feeds a unit kernel.
  uint64_t bus_group = 16;       // blocks per launch sequence of a small unit level in front of a bus chain (PHONIC_BUS_GROUP)
  bool overlap_bus = true;       // PHONIC_BUS_OVERLAP=0 (read at create): everything on the caller's stream
  hipEvent_t ev_scan_done = nullptr, ev_generic_done = nullptr;   // the generic kernel beside the fast kernels (launch_level)
  bool concurrent_generic = true;   // PHONIC_CONCURRENT_GENERIC=0 (read at create): the generic kernel behind the fast kernels, on one stream
  bool messages_due = false;    // StopSource messages wait for the first launch of the write call that has begun
  int32_t* d_error = nullptr;   // sticky consistency flags of the kernels (PG_DEVERR_*)
  unsigned long long* d_bus_progress = nullptr;  // progress words of the pipelined bus chain (pg_bus_pipeline)
  uint32_t bus_epoch = 0;       // launch number of the pipelined bus chain, carried by its progress words
  bool bus_pipeline = true;     // PHONIC_BUS_PIPELINE=0 (read at create): the bus chain behind a super-block stays one workgroup
  // host mirrors
  std::vector<HostMixer> mixers;        // [0] = main
  std::vector<HostVoice> voices;
  std::vector<int> stream_voices;       // ids of the host-fed voices (their feeds are flushed at the top of every write)
  std::vector<int> retired_voices;      // removed sources whose removal has not reached the device yet (the next topology upload carries it)
  std::vector<int> retired_ready;       // ... and those it has: their memory is released at the next graph_quiesce
  std::vector<std::unique_ptr<HostFx>> fx;
  std::vector<int> fx_mixer;            // effect id -> mixer id
  std::vector<int> source_unit_of_voice;  // main-mixer voices: unit slot
  uint64_t event_seq = 0;
  int main_active_voices = 0;           // feedback from the device (sync write only)
  bool ever_had_main_voice = false;
  // device tables
  DeviceVec<PgUnit> d_units;
  DeviceVec<PgVoice> d_voices;
  DeviceVec<PgFx> d_fx;
  DeviceTable<int32_t> d_voice_index, d_fx_index, d_order;
  // Command lists of the launch rounds: a device ring fed from a pinned host ring of the same size by asynchronous copies, one region
  // per round. A region is reused only after PG_CMD_RING commands have gone through since the last point at which the host knew the
  // stream to be drained (then it waits once): rounds with parameter automation neither allocate nor block.
  PgCmd* d_cmd_ring = nullptr;
  PgCmd* h_cmd_ring = nullptr;
  PgCmd* d_cmd_overflow = nullptr;
  size_t cmd_head = 0, cmds_since_sync = 0;
  hipStream_t last_stream = nullptr;   // the stream of the last write: mutating calls drain it before they touch device tables
  // control path (pg_ctrl.h): messages from any thread, drained at the top of write like MixedSource::process_messages
  pgc::CtrlRing ctrl{PG_CTRL_RING};
  pgc::ChunkTable<int8_t> fx_kind_tab;     // effect id -> kind, -1 once removed (readable from any thread)
  pgc::ChunkTable<int8_t> voice_alive_tab; // voice id -> 1 while it can take messages
  DeviceVec<PgSchedEntry> d_sched;       // [classes][2 banks]
  std::map<uint32_t, int> sched_class_of_ratio;
  uint64_t launch_counter = 0;
  std::vector<PgUnit> h_units;          // topology part only (kind, offsets); state fields are device-owned
  bool topo_dirty = true;
  std::vector<int32_t> order;           // launch order: sub-mixer units, then main-mixer source units by start time
  int n_graph_units = 0;                // units excluding bus
  // buffers
  float* d_unit_out = nullptr; size_t unit_out_rows = 0;
  float* d_partial = nullptr; size_t partial_rows = 0;
  float* d_bus = nullptr;               // [2*max_frames + 4]
  int* d_audible = nullptr;             // [audible_slots] audible_input of the bus chain, one word per block of a round / of a deferred-bus call
  hipEvent_t write_done_event = nullptr; // set by the sharded handle around a write: rides on the call's LAST mixer-sum launch as its stop event (write_done_attached says it did)
  bool write_done_attached = false;
  int* d_audible_out = nullptr;         // deferred-bus writes only (pg_sharded, direct delivery): the mixer sum leaves the call's words HERE instead of d_audible — the root's table, on the root's device
  int32_t* d_audible_tab = nullptr;     // [max_blocks][unit_out_rows] per-unit `audible` results, block by block (PgLaunch::audible_tab)
  bool status_pending = false;          // graph_enqueue_status ran, graph_collect_status has not
  float* h_pinned = nullptr;
  unsigned long long* h_feedback = nullptr;   // pinned, device-visible: (round << 32 | deferred units) written by the generic kernel
  unsigned long long* d_feedback = nullptr;   // its device address
  uint64_t last_change_round = 0;             // last round that may have left a unit out of steady state (topology, commands, mode switches)
  uint32_t stride = 0;
  unsigned long long* d_diag = nullptr;  // diagnostic builds
  // timing of the dominant kernel
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<uint32_t> ev_blocks;  // max_frames blocks the timed launch rendered (super-block launches: several)
  size_t ev_used = 0;
  // ... and of the main mixer's bus chain (one workgroup per effect: a latency chain — for graphs like BASELINE configs 2 and 4 it is the
  // launch that dominates by GPU time, pg_graph_bus_kernel_stats)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_bus_pool;
  std::vector<uint32_t> ev_bus_blocks;
  size_t ev_bus_used = 0;
  // ... and of the generic kernel's launches, with the counters behind pg_graph_dynamic_stats
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_gen_pool;
  size_t ev_gen_used = 0;
  uint64_t stat_unit_blocks = 0, stat_generic_launches = 0;
};

// ---- graph internals used by the sharded handle (pg_host.hip) -------------------------------------------------
int graph_quiesce(pg_graph* g);
void graph_begin_write(pg_graph* g, uint64_t pos);
void drain_control_messages_public(pg_graph* g);
bool graph_is_empty(const pg_graph* g);
uint64_t graph_next_main_event(const pg_graph* g);
int graph_enqueue_status(pg_graph* g, hipStream_t stream);
void graph_collect_status(pg_graph* g);
size_t graph_write_impl(pg_graph* g, float* d_out, size_t n_samples, uint64_t pos, hipStream_t stream, bool begin = true, size_t cap_frames = 0);
int process_bus_impl(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, hipStream_t s, int* bus_audible);
