// ---- voice-sharded graph: one object, n per-device graphs (SURVEY §8b `n_gpus`, §8e) ----------------------------------------------
// The reference's only parallel axis is independent sub-mixers handed to worker threads, each rendering into a private buffer that the
// caller sums (SubMixerThreadPool, src/source/mixed/submixer/thread_pool.rs:92-121,350-412; src/source/mixed.rs:522-536). Here the
// workers are GPUs: every sub-mixer (with everything under it) and every main-mixer source lives on ONE shard — the least loaded one
// when it is added, the greedy placement of WorkerTaskBatcher — state never migrates, each shard renders a partial master bus on its
// own device and stream, the partials meet on the root device and the main mixer's effect chain runs once, on the root, behind the
// sum. Two ways for the partials to meet (pg_sharded_set_reduce): peer copies + a sum kernel in shard order (default, deterministic),
// or an RCCL ncclReduce(sum) over xGMI on the shards' own streams. One process, one caller thread; the measured multi-GPU path of
// bench.py (one process per GPU, RCCL reduce through torch.distributed) shares everything below the ABI with this one.
//
// The handle is the ONE main MixedSource: every call the reference's mixer takes is routed to the shard that owns its target
// (src/source/mixed.rs:124-145,163-178,422-462), a write is cut at the main mixer's event times of ALL shards (so every shard splits
// its chunks where the one mixer would, mixed.rs:679-712), the bus chain sees one `audible` word per chunk, OR-ed over the shards
// (process_effects, mixed.rs:627-655), and write returns 0 exactly when the one mixer would (mixed.rs:664-670).
#include "pg_host_internal.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>

#include <rccl/rccl.h>  // types and enums only: the entry points are looked up at run time (pg_sharded_set_reduce), the library does not link RCCL

// bus[i] = own[i] + sum of the peers' partials in shard order; audible_out[c] = OR over the shards of their flag for chunk c
__global__ void pg_shard_sum_kernel(float* __restrict__ bus, const float* __restrict__ own, const float* __restrict__ gathered, int n_peers, size_t peer_stride, int n,
                                    const int* __restrict__ flags, int n_shards, int n_chunks, int* __restrict__ audible_out, int flag_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && audible_out) {
    for (int c = (int)threadIdx.x; c < n_chunks; c += (int)blockDim.x) {
      int a = 0;
      for (int k = 0; k < n_shards; ++k) a |= flags[k * flag_stride + c];
      audible_out[c] = a;
    }
  }
  if (i >= n) return;
  float acc = own[i];
  for (int p = 0; p < n_peers; ++p) acc = acc + gathered[(size_t)p * peer_stride + i];  // shard order: deterministic
  bus[i] = acc;
}

// RCCL entry points, resolved from the RCCL the process already holds (soname librccl.so.1: the one PyTorch bundles when the host is
// Python) or from the ROCm installation on the library's run path
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static std::mutex g_rccl_mutex;
static RcclApi g_rccl;
static int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return PG_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return set_error(PG_ERR_DEVICE, "RCCL is not available: %s", dlerror());
  RcclApi a;
  a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
  a.Reduce = (decltype(a.Reduce))dlsym(h, "ncclReduce");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.CommInitAll || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Reduce || !a.GetErrorString) { dlclose(h); return set_error(PG_ERR_DEVICE, "RCCL lacks an entry point"); }
  a.handle = h;
  g_rccl = a;
  return PG_OK;
}

// One issuing thread per shard behind the first. A write enqueues every shard's launch sequence (tens of HIP calls each); from ONE host thread
// that is ~20 us per shard one after the other — 170 us per 1024-frame block at eight shards, several times the block's GPU time on eight
// devices (tools/exp_sharded_host_time.py) — so the caller's thread issues shard 0 and hands the others to their workers, which issue in
// parallel on their own devices (hipSetDevice is per thread) and report back before the root's sum is enqueued. A worker spins for a short
// while after a job (an offline pull loop calls back to back) and then sleeps on its condition variable (a real-time callback comes every
// few milliseconds). PHONIC_SHARD_THREADS=0: the caller's thread issues everything (as before).
struct ShardWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<uint64_t> posted{0}, finished{0};
  std::function<int()> job;
  int rc = 0;
  std::string error;
  bool quit = false;
  void run() {
    uint64_t seen = 0;
    for (;;) {
      int spins = 0;
      while (posted.load(std::memory_order_acquire) == seen && spins < 20000) { ++spins; __builtin_ia32_pause(); }
      if (posted.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return quit || posted.load(std::memory_order_acquire) != seen; });
      }
      if (quit) return;
      seen = posted.load(std::memory_order_acquire);
      rc = job();
      if (rc) error = pg_last_error_message();
      finished.store(seen, std::memory_order_release);
    }
  }
  void post(std::function<int()> f) {
    job = std::move(f);
    { std::lock_guard<std::mutex> lock(m); posted.fetch_add(1, std::memory_order_release); }
    cv.notify_one();
  }
  int wait() {  // (the caller needs the result before it can go on: it spins — with a bound: a job that never returns fails the handle)
    const uint64_t want = posted.load(std::memory_order_acquire);
    static const double limit_s = [] { const char* e = getenv("PHONIC_SHARD_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 30.0; }();
    auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (finished.load(std::memory_order_acquire) != want) {
      __builtin_ia32_pause();
      if ((++spins & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
        return set_error(PG_ERR_DEVICE, "a shard's issuing thread did not report back within %.0f s: the sharded graph is disabled", limit_s);
    }
    if (rc) set_error(rc, "%s", error.c_str());
    return rc;
  }
  void stop() {
    { std::lock_guard<std::mutex> lock(m); quit = true; }
    cv.notify_one();
    if (th.joinable()) th.join();
  }
};

constexpr size_t PG_FLAG_BANKS = 4;
struct pg_sharded_graph {
  std::vector<std::unique_ptr<ShardWorker>> workers;   // [shard - 1]; empty: the caller's thread issues every shard
  std::vector<pg_graph*> shards;
  std::vector<int> load;                 // sub-mixers + main-mixer sources placed on each shard
  uint32_t sample_rate = 48000;
  size_t max_frames = 0, max_blocks = 1, stride = 0;
  size_t stage_frames = 0;               // frames the partial / gather / bus staging holds: max(max_blocks x max_frames, one chunk of the reference's grid)
  size_t flag_stride = PG_AUDIBLE_SLOTS; // words per shard in d_flags (= the shards' audible_slots)
  std::vector<float*> d_partial;         // per shard, on its device: [max_blocks * stride] partial master bus
  float* d_gather = nullptr;             // root device: the peers' partials [(n - 1)][max_blocks * stride]
  int* d_flags = nullptr;                // root device: the shards' `audible` words [PG_FLAG_BANKS][n][PG_AUDIBLE_SLOTS]: one bank per segment in flight
  // Direct delivery (peer-copy mode, round 5): a shard's mixer sum writes its partial bus and its words straight into the root's gather buffer
  // and word table (peer access across devices) — no copy launches behind the render; the segments of consecutive calls land in a ring of
  // regions / banks, and the shards' streams wait for the root's last sum only when a region comes round again.
  bool direct = true;
  uint64_t seg_counter = 0;
  std::vector<hipEvent_t> summed_bank;   // [PG_FLAG_BANKS] root: the sum of the last segment that used bank b has read its region and its words
  std::vector<char> bank_recorded;
  hipEvent_t last_sum = nullptr;         // the sum event recorded last (behind every earlier one on the root's stream)
  bool last_banked = false;
  float* d_bus = nullptr;                // root device: the summed bus of a write with a host buffer
  float* h_pinned = nullptr;
  std::vector<hipEvent_t> done;          // per shard: partial (and flags) arrived on the root
  hipEvent_t summed = nullptr;           // root: the sum kernel has read d_gather / d_flags (the peers' next copies wait for it)
  bool summed_recorded = false;
  bool failed = false;
  int reduce_mode = PG_REDUCE_PEER_COPY;
  std::vector<ncclComm_t> comms;         // PG_REDUCE_RCCL: one communicator per shard (ncclCommInitAll over the shards' devices)
  // global id -> shard << 24 | local id; append-only, readable from any thread (control calls)
  pgc::ChunkTable<int32_t> mixer_map, fx_map, voice_map;
};
static inline int shard_of(int32_t packed) { return (int)((uint32_t)packed >> 24); }
static inline int local_of(int32_t packed) { return (int)((uint32_t)packed & 0xffffffu); }

static int sharded_alloc_buffers(pg_sharded_graph* s) {
  s->stage_frames = std::max<size_t>(s->max_frames * s->max_blocks, PG_MAX_FRAMES);
  const size_t words = 2 * s->stage_frames + 4;
  const size_t n = s->shards.size();
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(s->shards[i]->device));
    if (s->d_partial[i]) (void)pg_free(s->d_partial[i]);
    s->d_partial[i] = nullptr;
    HIP_TRY(pg_malloc((void**)&s->d_partial[i], words * sizeof(float)));
  }
  HIP_TRY(hipSetDevice(s->shards[0]->device));
  if (s->d_gather) (void)pg_free(s->d_gather);
  if (s->d_bus) (void)pg_free(s->d_bus);
  if (s->h_pinned) (void)pg_host_free(s->h_pinned);
  s->d_gather = nullptr; s->d_bus = nullptr; s->h_pinned = nullptr;
  HIP_TRY(pg_malloc((void**)&s->d_gather, std::max<size_t>(n - 1, 1) * words * sizeof(float)));
  HIP_TRY(pg_malloc((void**)&s->d_bus, words * sizeof(float)));
  HIP_TRY(pg_host_malloc((void**)&s->h_pinned, words * sizeof(float), hipHostMallocDefault));
  return PG_OK;
}
static void sharded_drop_comms(pg_sharded_graph* s) {
  for (ncclComm_t c : s->comms) if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
  s->comms.clear();
}
static int sharded_wait_all(pg_sharded_graph* s) {
  for (pg_graph* g : s->shards) { (void)hipSetDevice(g->device); HIP_TRY(pg_stream_sync(g->stream)); g->cmds_since_sync = 0; }
  return PG_OK;
}

extern "C" {

pg_sharded_graph* pg_sharded_create(uint32_t sample_rate, uint32_t channel_count, size_t max_frames, const int* devices, int n_devices) {
  if (n_devices < 1 || n_devices > 64 || !devices) { set_error(PG_ERR_PARAMETER, "1..=64 shards"); return nullptr; }
  std::unique_ptr<pg_sharded_graph> s(new pg_sharded_graph());
  s->sample_rate = sample_rate; s->max_frames = max_frames; s->stride = 2 * max_frames;
  for (int i = 0; i < n_devices; ++i) {
    pg_graph* g = pg_graph_create(sample_rate, channel_count, max_frames, devices[i]);
    if (!g) { for (pg_graph* h : s->shards) pg_graph_destroy(h); return nullptr; }
    g->defer_bus = true;  // the main mixer's chain runs once, behind the sum of all shards
    s->shards.push_back(g);
    s->load.push_back(0);
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error(PG_ERR_DEVICE, "hipEventCreate failed"); for (pg_graph* h : s->shards) pg_graph_destroy(h); return nullptr; }
    s->done.push_back(e);
  }
  s->d_partial.assign((size_t)n_devices, nullptr);
  (void)hipSetDevice(devices[0]);
  s->flag_stride = s->shards[0]->audible_slots;
  const size_t flag_bytes = PG_FLAG_BANKS * (size_t)n_devices * s->flag_stride * sizeof(int);
  if (hipEventCreateWithFlags(&s->summed, hipEventDisableTiming) != hipSuccess || pg_malloc((void**)&s->d_flags, flag_bytes) != hipSuccess || sharded_alloc_buffers(s.get())) {
    set_error(PG_ERR_DEVICE, "device allocation failed");
    pg_sharded_destroy(s.release());   // (shards, events and what was allocated so far)
    return nullptr;
  }
  (void)pg_memset(s->d_flags, 0, flag_bytes);
  s->bank_recorded.assign(PG_FLAG_BANKS, 0);
  for (size_t b = 0; b < PG_FLAG_BANKS; ++b) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error(PG_ERR_DEVICE, "hipEventCreate failed"); pg_sharded_destroy(s.release()); return nullptr; }
    s->summed_bank.push_back(e);
  }
  s->mixer_map.append(0);  // global mixer 0 = the main mixer (its chain lives on the root shard)
  // direct delivery needs the root's memory mapped on every shard's device (same device: it is); PHONIC_SHARD_DIRECT=0 keeps the copies
  { const char* d = getenv("PHONIC_SHARD_DIRECT"); if (d && d[0] == '0') s->direct = false; }
  for (int i = 1; i < n_devices && s->direct; ++i) {
    if (devices[i] == devices[0]) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) != hipSuccess || !can) { s->direct = false; break; }
    (void)hipSetDevice(devices[i]);
    const hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) s->direct = false;
    (void)hipGetLastError();
  }
  (void)hipSetDevice(devices[0]);
  const char* e = getenv("PHONIC_SHARD_THREADS");
  if (n_devices > 1 && !(e && e[0] == '0')) {
    for (int i = 1; i < n_devices; ++i) {
      s->workers.emplace_back(new ShardWorker());
      ShardWorker* w = s->workers.back().get();
      w->th = std::thread([w] { w->run(); });
    }
  }
  return s.release();
}
void pg_sharded_destroy(pg_sharded_graph* s) {
  if (!s) return;
  for (auto& w : s->workers) w->stop();
  s->workers.clear();
  (void)sharded_wait_all(s);
  sharded_drop_comms(s);
  for (size_t i = 0; i < s->shards.size(); ++i) {
    (void)hipSetDevice(s->shards[i]->device);
    if (i < s->d_partial.size() && s->d_partial[i]) (void)pg_free(s->d_partial[i]);
    if (i < s->done.size()) (void)hipEventDestroy(s->done[i]);
  }
  (void)hipSetDevice(s->shards[0]->device);
  if (s->summed) (void)hipEventDestroy(s->summed);
  for (hipEvent_t e : s->summed_bank) (void)hipEventDestroy(e);
  if (s->d_gather) (void)pg_free(s->d_gather);
  if (s->d_bus) (void)pg_free(s->d_bus);
  if (s->d_flags) (void)pg_free(s->d_flags);
  if (s->h_pinned) (void)pg_host_free(s->h_pinned);
  for (pg_graph* g : s->shards) pg_graph_destroy(g);
  delete s;
}
int pg_sharded_shard_count(pg_sharded_graph* s) { return (int)s->shards.size(); }
int pg_sharded_set_max_blocks_per_launch(pg_sharded_graph* s, int n_blocks) {
  { int rc = sharded_wait_all(s); if (rc) return rc; }
  for (pg_graph* g : s->shards) { int rc = pg_graph_set_max_blocks_per_launch(g, n_blocks); if (rc) return rc; }
  s->max_blocks = (size_t)n_blocks;
  return sharded_alloc_buffers(s);
}
// How the partial buses meet on the root. PG_REDUCE_RCCL creates one communicator per shard (ncclCommInitAll over the shards' devices:
// every device may then be listed once only) and fails with PG_ERR_DEVICE + the RCCL error text when that is impossible — the mode
// stays what it was; there is no silent fallback.
int pg_sharded_set_reduce(pg_sharded_graph* s, int mode) {
  if (mode != PG_REDUCE_PEER_COPY && mode != PG_REDUCE_RCCL) return set_error(PG_ERR_PARAMETER, "unknown reduce mode %d", mode);
  if (mode == s->reduce_mode) return PG_OK;
  { int rc = sharded_wait_all(s); if (rc) return rc; }
  if (mode == PG_REDUCE_PEER_COPY) { sharded_drop_comms(s); s->reduce_mode = mode; return PG_OK; }
  { int rc = rccl_load(); if (rc) return rc; }
  std::vector<int> devs;
  for (pg_graph* g : s->shards) devs.push_back(g->device);
  for (size_t i = 0; i < devs.size(); ++i) for (size_t j = i + 1; j < devs.size(); ++j)
    if (devs[i] == devs[j]) return set_error(PG_ERR_PARAMETER, "PG_REDUCE_RCCL needs one device per shard (device %d is listed twice)", devs[i]);
  std::vector<ncclComm_t> comms(devs.size(), nullptr);
  const ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
  if (r != ncclSuccess) return set_error(PG_ERR_DEVICE, "ncclCommInitAll over %d device(s) failed: %s", (int)devs.size(), g_rccl.GetErrorString(r));
  s->comms = comms;
  s->reduce_mode = mode;
  return PG_OK;
}
int pg_sharded_reduce_mode(pg_sharded_graph* s) { return s->reduce_mode; }

static int sharded_least_loaded(const pg_sharded_graph* s) {
  int best = 0;
  for (size_t i = 1; i < s->load.size(); ++i) if (s->load[i] < s->load[best]) best = (int)i;
  return best;
}
static bool sharded_mixer(pg_sharded_graph* s, int mixer_id, int32_t& packed) {
  if (mixer_id < 0 || (size_t)mixer_id >= s->mixer_map.size()) { set_error(PG_ERR_NOT_FOUND, "Mixer with id %d not found", mixer_id); return false; }
  packed = s->mixer_map.get((size_t)mixer_id);
  return true;
}
int pg_sharded_add_mixer_to(pg_sharded_graph* s, int parent_mixer_id) {
  int32_t pk;
  if (!sharded_mixer(s, parent_mixer_id, pk)) return -PG_ERR_NOT_FOUND;
  const int shard = parent_mixer_id == 0 ? sharded_least_loaded(s) : shard_of(pk);  // a nested sub-mixer lives with its parent
  const int local = pg_graph_add_mixer_to(s->shards[shard], parent_mixer_id == 0 ? 0 : local_of(pk));
  if (local < 0) return local;
  if (parent_mixer_id == 0) s->load[shard] += 1;
  const int id = (int)s->mixer_map.size();
  if (local > 0xffffff || !s->mixer_map.append((int32_t)(((uint32_t)shard << 24) | (uint32_t)local))) return -set_error(PG_ERR_STATE, "too many mixers");
  return id;
}
int pg_sharded_add_mixer(pg_sharded_graph* s) { return pg_sharded_add_mixer_to(s, 0); }
int pg_sharded_add_effect(pg_sharded_graph* s, int mixer_id, int kind, const pg_effect_init* init) {
  int32_t pk;
  if (!sharded_mixer(s, mixer_id, pk)) return -PG_ERR_NOT_FOUND;
  const int shard = mixer_id == 0 ? 0 : shard_of(pk);  // main-mixer effects: the bus chain on the root
  const int local = pg_graph_add_effect(s->shards[shard], mixer_id == 0 ? 0 : local_of(pk), kind, init);
  if (local < 0) return local;
  const int id = (int)s->fx_map.size();
  if (local > 0xffffff || !s->fx_map.append((int32_t)(((uint32_t)shard << 24) | (uint32_t)local))) return -set_error(PG_ERR_STATE, "too many effects");
  return id;
}
int pg_sharded_add_voice(pg_sharded_graph* s, int mixer_id, const float* pcm, size_t n_frames, uint32_t src_channels, uint32_t src_rate, const pg_voice_options* opt) {
  int32_t pk;
  if (!sharded_mixer(s, mixer_id, pk)) return -PG_ERR_NOT_FOUND;
  const int shard = mixer_id == 0 ? sharded_least_loaded(s) : shard_of(pk);
  const int local = pg_graph_add_voice(s->shards[shard], mixer_id == 0 ? 0 : local_of(pk), pcm, n_frames, src_channels, src_rate, opt);
  if (local < 0) return local;
  if (mixer_id == 0) s->load[shard] += 1;
  const int id = (int)s->voice_map.size();
  if (local > 0xffffff || !s->voice_map.append((int32_t)(((uint32_t)shard << 24) | (uint32_t)local))) return -set_error(PG_ERR_STATE, "too many voices");
  return id;
}
// host-fed sources (pg_graph_add_stream_voice): placed like any other source, fed through the shard that owns them
int pg_sharded_add_stream_voice(pg_sharded_graph* s, int mixer_id, uint32_t channels, uint32_t rate, size_t capacity_frames, const pg_voice_options* opt) {
  int32_t pk;
  if (!sharded_mixer(s, mixer_id, pk)) return -PG_ERR_NOT_FOUND;
  const int shard = mixer_id == 0 ? sharded_least_loaded(s) : shard_of(pk);
  const int local = pg_graph_add_stream_voice(s->shards[shard], mixer_id == 0 ? 0 : local_of(pk), channels, rate, capacity_frames, opt);
  if (local < 0) return local;
  if (mixer_id == 0) s->load[shard] += 1;
  const int id = (int)s->voice_map.size();
  if (local > 0xffffff || !s->voice_map.append((int32_t)(((uint32_t)shard << 24) | (uint32_t)local))) return -set_error(PG_ERR_STATE, "too many voices");
  return id;
}
int pg_sharded_feed_voice(pg_sharded_graph* s, int voice_id, const float* frames, size_t n_frames) {
  if (voice_id < 0 || (size_t)voice_id >= s->voice_map.size()) return set_error(PG_ERR_NOT_FOUND, "Source with id %d not found", voice_id);
  const int32_t pk = s->voice_map.get((size_t)voice_id);
  return pg_graph_feed_voice(s->shards[shard_of(pk)], local_of(pk), frames, n_frames);
}
int pg_sharded_end_stream_voice(pg_sharded_graph* s, int voice_id) {
  if (voice_id < 0 || (size_t)voice_id >= s->voice_map.size()) return set_error(PG_ERR_NOT_FOUND, "Source with id %d not found", voice_id);
  const int32_t pk = s->voice_map.get((size_t)voice_id);
  return pg_graph_end_stream_voice(s->shards[shard_of(pk)], local_of(pk));
}
int64_t pg_sharded_stream_voice_consumed(pg_sharded_graph* s, int voice_id) {
  if (voice_id < 0 || (size_t)voice_id >= s->voice_map.size()) { set_error(PG_ERR_NOT_FOUND, "Source with id %d not found", voice_id); return -1; }
  const int32_t pk = s->voice_map.get((size_t)voice_id);
  return pg_graph_stream_voice_consumed(s->shards[shard_of(pk)], local_of(pk));
}
int pg_sharded_shard_of_mixer(pg_sharded_graph* s, int mixer_id) {
  int32_t pk;
  if (!sharded_mixer(s, mixer_id, pk)) return -PG_ERR_NOT_FOUND;
  return mixer_id == 0 ? 0 : shard_of(pk);
}
// Player::remove_mixer / remove_effect / move_effect (src/player.rs:825-867,942-990 -> MixerMessage::RemoveMixer / RemoveEffect / MoveEffect,
// src/source/mixed.rs:422-462) on the shard that owns the target; ids stay global and are never reused.
int pg_sharded_remove_mixer(pg_sharded_graph* s, int mixer_id) {
  if (mixer_id == 0) return set_error(PG_ERR_PARAMETER, "Cannot remove the main mixer");
  int32_t pk;
  if (!sharded_mixer(s, mixer_id, pk)) return PG_ERR_NOT_FOUND;
  pg_graph* g = s->shards[shard_of(pk)];
  const int local = local_of(pk);
  const bool top_level = local > 0 && local < (int)g->mixers.size() && !g->mixers[local].removed && g->mixers[local].parent == 0;
  const int rc = pg_graph_remove_mixer(g, local);
  if (rc == PG_OK && top_level && s->load[shard_of(pk)] > 0) s->load[shard_of(pk)] -= 1;  // the shard has room for the next sub-mixer again
  return rc;
}
int pg_sharded_remove_effect(pg_sharded_graph* s, int effect_id) {
  if (effect_id < 0 || (size_t)effect_id >= s->fx_map.size()) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  const int32_t pk = s->fx_map.get((size_t)effect_id);
  return pg_graph_remove_effect(s->shards[shard_of(pk)], local_of(pk));
}
int pg_sharded_move_effect(pg_sharded_graph* s, int effect_id, int mixer_id, int movement, int offset) {
  if (effect_id < 0 || (size_t)effect_id >= s->fx_map.size()) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  int32_t mk;
  if (!sharded_mixer(s, mixer_id, mk)) return PG_ERR_NOT_FOUND;
  const int32_t pk = s->fx_map.get((size_t)effect_id);
  const int m_shard = mixer_id == 0 ? 0 : shard_of(mk), m_local = mixer_id == 0 ? 0 : local_of(mk);
  if (m_shard != shard_of(pk)) return set_error(PG_ERR_PARAMETER, "Effect %d does not belong to mixer %d", effect_id, mixer_id);
  return pg_graph_move_effect(s->shards[m_shard], local_of(pk), m_local, movement, offset);
}
// control calls (any thread): routed to the owning shard's message ring
#define SHARDED_FX(s, effect_id, pk) \
  if ((effect_id) < 0 || (size_t)(effect_id) >= (s)->fx_map.size()) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", (effect_id)); \
  const int32_t pk = (s)->fx_map.get((size_t)(effect_id))
#define SHARDED_VOICE(s, voice_id, pk) \
  if ((voice_id) < 0 || (size_t)(voice_id) >= (s)->voice_map.size()) return set_error(PG_ERR_NOT_FOUND, "Source with id %d not found", (voice_id)); \
  const int32_t pk = (s)->voice_map.get((size_t)(voice_id))
int pg_sharded_schedule_param(pg_sharded_graph* s, int effect_id, uint32_t fourcc, float value, int is_normalized, uint64_t sample_time) {
  SHARDED_FX(s, effect_id, pk);
  return pg_graph_schedule_param(s->shards[shard_of(pk)], local_of(pk), fourcc, value, is_normalized, sample_time);
}
int pg_sharded_schedule_reset(pg_sharded_graph* s, int effect_id, uint64_t sample_time) {
  SHARDED_FX(s, effect_id, pk);
  return pg_graph_schedule_reset(s->shards[shard_of(pk)], local_of(pk), sample_time);
}
int pg_sharded_set_voice_volume(pg_sharded_graph* s, int voice_id, float volume, uint64_t sample_time) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_set_voice_volume(s->shards[shard_of(pk)], local_of(pk), volume, sample_time);
}
int pg_sharded_set_voice_panning(pg_sharded_graph* s, int voice_id, float panning, uint64_t sample_time) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_set_voice_panning(s->shards[shard_of(pk)], local_of(pk), panning, sample_time);
}
int pg_sharded_set_voice_speed(pg_sharded_graph* s, int voice_id, double speed, float glide_semitones_per_second, uint64_t sample_time) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_set_voice_speed(s->shards[shard_of(pk)], local_of(pk), speed, glide_semitones_per_second, sample_time);
}
int pg_sharded_seek_voice(pg_sharded_graph* s, int voice_id, double position_seconds, uint64_t sample_time) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_seek_voice(s->shards[shard_of(pk)], local_of(pk), position_seconds, sample_time);
}
int pg_sharded_stop_voice(pg_sharded_graph* s, int voice_id, uint64_t sample_time) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_stop_voice(s->shards[shard_of(pk)], local_of(pk), sample_time);
}
int pg_sharded_remove_voice(pg_sharded_graph* s, int voice_id) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_remove_voice(s->shards[shard_of(pk)], local_of(pk));
}
int pg_sharded_stop_all_voices(pg_sharded_graph* s) {
  for (pg_graph* g : s->shards) { int rc = pg_graph_stop_all_voices(g); if (rc) return rc; }
  return PG_OK;
}
int pg_sharded_is_voice_playing(pg_sharded_graph* s, int voice_id) {
  if (voice_id < 0 || (size_t)voice_id >= s->voice_map.size()) return 0;
  const int32_t pk = s->voice_map.get((size_t)voice_id);
  return pg_graph_is_voice_playing(s->shards[shard_of(pk)], local_of(pk));
}

}  // extern "C"

// One span of a write: frames without a main-mixer event of any shard inside, starting on the chunk grid (at the call's start, at an event,
// or a whole number of chunks behind either). Every shard renders its partial bus (asynchronously, on its own device and stream) into its
// staging at `stage_off` and leaves one `audible` word per piece; partials and words meet on the root; the bus chain runs behind the sum with
// the OR-ed words.
static int sharded_render_segment(pg_sharded_graph* s, float* d_dst, size_t stage_off, size_t n_samples, uint64_t pos) {
  const size_t cap = 2 * s->stage_frames;
  const size_t n = s->shards.size();
  pg_graph* root = s->shards[0];
  // words of the span: one per piece (whole chunks of ceil(PG_MAX_FRAMES / max_frames) pieces, then the pieces of the last, shorter chunk)
  const size_t fr = n_samples / 2, per_chunk = (PG_MAX_FRAMES + s->max_frames - 1) / s->max_frames;
  const int n_chunks = (int)std::min<size_t>((fr / PG_MAX_FRAMES) * per_chunk + (fr % PG_MAX_FRAMES + s->max_frames - 1) / s->max_frames, s->flag_stride);
  const bool rccl = s->reduce_mode == PG_REDUCE_RCCL;
  const bool direct = s->direct && !rccl;
  // Direct delivery in BANKS: a segment of at most stage_frames / PG_FLAG_BANKS frames (a real-time call) takes region b of the partial /
  // gather buffers and bank b of the word table, b = its number mod PG_FLAG_BANKS, and the shards' streams wait — in front of their render:
  // its sum kernel is what writes there — for the sum of the segment that used bank b LAST, PG_FLAG_BANKS segments ago: long done, so shards
  // on different devices do not meet once per call. Longer segments (and the copy / RCCL modes) use the caller's region and bank 0 and wait
  // for the last recorded sum, which is behind every earlier one.
  const bool banked = direct && fr * PG_FLAG_BANKS <= s->stage_frames;
  const size_t bank = banked ? (size_t)(s->seg_counter % PG_FLAG_BANKS) : 0;
  if (banked) stage_off = bank * 2 * (s->stage_frames / PG_FLAG_BANKS);
  int* const flags = s->d_flags + bank * n * s->flag_stride;
  hipEvent_t wait_ev = nullptr;
  if (banked && s->last_banked) wait_ev = s->bank_recorded[bank] ? s->summed_bank[bank] : nullptr;
  else wait_ev = s->last_sum;
  s->seg_counter += 1;
  const bool wait_sum = wait_ev != nullptr;
  // every shard's launch sequence, its `audible` words and (peer-copy mode) the copies that carry both to the root: issued by the shard's own
  // thread when the handle has workers (shard 0 by the caller's), else one after the other here
  auto issue = [s, stage_off, n_samples, pos, n_chunks, rccl, direct, wait_sum, wait_ev, flags, cap, root](size_t i) -> int {
    pg_graph* g = s->shards[i];
    HIP_TRY(hipSetDevice(g->device));
    // direct delivery: a peer's partial bus goes straight to its slot of the root's gather ring, every shard's words to its row of the bank
    float* part = (direct && i > 0) ? s->d_gather + (i - 1) * (cap + 4) + stage_off : s->d_partial[i] + stage_off;
    g->d_audible_out = direct ? flags + i * s->flag_stride : nullptr;
    int* const words = direct ? g->d_audible_out : g->d_audible;
    if (direct && wait_sum) HIP_TRY(hipStreamWaitEvent(g->stream, wait_ev, 0));   // (in front of the render: its sum kernel is what writes there)
    g->write_done_event = (direct && i > 0) ? s->done[i] : nullptr;   // "this shard's partial has arrived" rides on its last sum launch
    const size_t w = graph_write_impl(g, part, n_samples, pos, g->stream, false);
    const bool done_rode = g->write_done_event != nullptr && g->write_done_attached;
    g->write_done_event = nullptr;
    if (g->failed) return PG_ERR_DEVICE;
    if (w != 0 && w != n_samples) return set_error(PG_ERR_STATE, "shard %d rendered %zu of %zu samples of a span", (int)i, w, n_samples);
    if (w == 0) {  // nothing on this shard: a silent partial and silent flags (not the words an earlier call left there)
      HIP_TRY(hipMemsetAsync(part, 0, n_samples * sizeof(float), g->stream));
      HIP_TRY(hipMemsetAsync(words, 0, (direct ? (size_t)n_chunks : g->audible_slots) * sizeof(int), g->stream));
    }
    if (rccl) return PG_OK;
    if (direct) { if (i > 0 && (!done_rode || w == 0)) HIP_TRY(hipEventRecord(s->done[i], g->stream)); return PG_OK; }
    // the root's sum of the previous segment (or call) must have read the gather buffers before this shard overwrites them
    if (wait_sum) HIP_TRY(hipStreamWaitEvent(g->stream, wait_ev, 0));
    if (i > 0) {
      HIP_TRY(hipMemcpyPeerAsync(s->d_gather + (i - 1) * (cap + 4) + stage_off, root->device, part, g->device, n_samples * sizeof(float), g->stream));
      HIP_TRY(hipMemcpyPeerAsync(flags + i * s->flag_stride, root->device, g->d_audible, g->device, (size_t)n_chunks * sizeof(int), g->stream));
      HIP_TRY(hipEventRecord(s->done[i], g->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(flags, g->d_audible, (size_t)n_chunks * sizeof(int), hipMemcpyDeviceToDevice, g->stream));
    }
    return PG_OK;
  };
  if (!s->workers.empty()) {
    for (size_t i = 1; i < n; ++i) s->workers[i - 1]->post([issue, i] { return issue(i); });
    int rc = issue(0);
    for (size_t i = 1; i < n; ++i) { const int r = s->workers[i - 1]->wait(); if (!rc) rc = r; }   // (every done[i] is recorded before the root waits for it)
    if (rc) return rc;
  } else {
    for (size_t i = 0; i < n; ++i) { const int rc = issue(i); if (rc) return rc; }
  }
  HIP_TRY(hipSetDevice(root->device));
  if (rccl) {
    // ncclReduce(sum) of the partial buses and ncclReduce(max) of the `audible` words (0 / 1: max = OR), root = shard 0, every shard's
    // operation on its own stream behind its render; the root's stream then holds the summed bus and the words
    ncclResult_t r = g_rccl.GroupStart();
    for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
      pg_graph* g = s->shards[i];
      r = g_rccl.Reduce(s->d_partial[i] + stage_off, i == 0 ? d_dst : nullptr, n_samples, ncclFloat32, ncclSum, 0, s->comms[i], g->stream);
      if (r == ncclSuccess) r = g_rccl.Reduce(g->d_audible, i == 0 ? (void*)s->d_flags : nullptr, (size_t)n_chunks, ncclInt32, ncclMax, 0, s->comms[i], g->stream);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return set_error(PG_ERR_DEVICE, "ncclReduce of the master bus failed: %s", g_rccl.GetErrorString(r));
    HIP_TRY(hipSetDevice(root->device));
    HIP_TRY(hipMemcpyAsync(root->d_audible, s->d_flags, (size_t)n_chunks * sizeof(int), hipMemcpyDeviceToDevice, root->stream));
  } else {
    for (size_t i = 1; i < n; ++i) HIP_TRY(hipStreamWaitEvent(root->stream, s->done[i], 0));
    hipEvent_t ev = banked ? s->summed_bank[bank] : s->summed;   // rides on the sum's launch as its stop event
    hipExtLaunchKernelGGL(pg_shard_sum_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, root->stream, nullptr, ev, 0, d_dst, s->d_partial[0] + stage_off,
                          s->d_gather + stage_off, (int)n - 1, cap + 4, (int)n_samples, flags, (int)n, n_chunks, root->d_audible, (int)s->flag_stride);
    HIP_TRY(hipGetLastError());
    if (banked) s->bank_recorded[bank] = 1;
    s->last_sum = ev; s->last_banked = banked;
    s->summed_recorded = true;
  }
  root->defer_pos = UINT64_MAX;   // (the handle cuts the spans itself: the root's bus chain takes no recorded cuts)
  return process_bus_impl(root, d_dst, n_samples, pos, root->stream, root->d_audible);
}

// Source::write of the sharded main mixer: ONE write call of the one mixer whatever its length (process_messages once on every shard, one
// call end — src/source/mixed.rs:659-719), walked on the reference's chunk grid: min(remaining, PG_MAX_FRAMES) frames from the call's start
// and from every main-mixer event of ANY shard. A segment between two events is rendered in spans of whole chunks, as many as the staging
// holds (stage_frames >= one chunk), so every shard takes its per-chunk decisions on the chunks the one mixer would see — whatever
// max_blocks x max_frames is (round-4 advisor finding: the host-buffer write used to open a fresh call per max_blocks x max_frames frames).
// `contiguous`: d_out holds the whole call (device writes); else every span lands at d_out + 0 (the staging bus of host writes) and
// `span_done(offset, samples)` takes it away. Returns the samples written, 0 when the one main mixer would return 0: no playing source, no
// sub-mixer and no pending event on any shard, and no effect on mixer 0.
static size_t sharded_write_walk(pg_sharded_graph* s, float* d_out, bool contiguous, size_t n_samples, uint64_t pos, const std::function<int(size_t, size_t)>& span_done) {
  if (s->failed) return 0;
  if (n_samples % 2 != 0) { set_error(PG_ERR_PARAMETER, "n_samples must be a multiple of the channel count"); return 0; }
  // process_messages of the one mixer: every shard takes its messages now, none later in this call
  bool empty = true;
  for (pg_graph* g : s->shards) {
    if (g->failed) { s->failed = true; return 0; }
    graph_begin_write(g, pos);
    g->call_end = pos + n_samples / 2;   // (one write of the one mixer, whatever segments it is cut into)
    empty &= graph_is_empty(g);
  }
  if (empty) return 0;
  const uint64_t frames = n_samples / 2, CH = PG_MAX_FRAMES;
  // spans hold whole chunks; the words of a span must fit the shards' word tables (flag_stride = max(64, pieces of one chunk))
  const uint64_t per_chunk = (CH + s->max_frames - 1) / s->max_frames;
  const uint64_t span_chunks = std::max<uint64_t>(1, std::min<uint64_t>(s->stage_frames / CH, s->flag_stride / per_chunk));
  uint64_t done = 0;
  size_t stage_off = 0;
  while (done < frames) {
    // the segment ends where the next main-mixer event of any shard comes due (events at or before `now` apply at the segment's head)
    const uint64_t now = pos + done;
    uint64_t seg = frames - done;
    for (pg_graph* g : s->shards)
      for (const Event& e : g->mixers[0].events) { if (e.sample_time > now) { seg = std::min<uint64_t>(seg, e.sample_time - now); break; } }
    uint64_t off = 0;
    while (off < seg) {
      const uint64_t n = std::min<uint64_t>(seg - off, span_chunks * CH);
      if (stage_off + n * 2 > 2 * s->stage_frames) stage_off = 0;   // (the staging is a ring of spans: reuse is ordered by the `summed` event and the shards' own streams)
      float* dst = contiguous ? d_out + (size_t)(done + off) * 2 : d_out;
      if (sharded_render_segment(s, dst, contiguous ? stage_off : 0, (size_t)n * 2, now + off)) { s->failed = true; return 0; }
      if (span_done && span_done((size_t)(done + off) * 2, (size_t)n * 2)) { s->failed = true; return 0; }
      stage_off += (size_t)n * 2;
      off += n;
    }
    done += seg;
  }
  return n_samples;
}

extern "C" {

size_t pg_sharded_write_device(pg_sharded_graph* s, float* d_out, size_t n_samples, uint64_t pos_in_frames) { return sharded_write_walk(s, d_out, true, n_samples, pos_in_frames, nullptr); }
int pg_sharded_synchronize(pg_sharded_graph* s) { return sharded_wait_all(s); }
size_t pg_sharded_write(pg_sharded_graph* s, float* out, size_t n_samples, uint64_t pos_in_frames) {
  // every span: the status words of every shard, the span's samples to the pinned staging, one wait, then out of the way for the next span
  auto span_done = [s, out](size_t off, size_t n) -> int {
    pg_graph* root = s->shards[0];
    // device feedback per shard: how many of its main-mixer sources are still alive (transient sources are dropped when exhausted, mixed.rs:715)
    for (pg_graph* g : s->shards) { (void)hipSetDevice(g->device); if (graph_enqueue_status(g, g->stream) != PG_OK) return PG_ERR_DEVICE; }
    (void)hipSetDevice(root->device);
    if (hipMemcpyAsync(s->h_pinned, s->d_bus, n * sizeof(float), hipMemcpyDeviceToHost, root->stream) != hipSuccess || sharded_wait_all(s) != PG_OK) return PG_ERR_DEVICE;
    for (pg_graph* g : s->shards) graph_collect_status(g);
    memcpy(out + off, s->h_pinned, n * sizeof(float));
    return PG_OK;
  };
  return sharded_write_walk(s, s->d_bus, false, n_samples, pos_in_frames, span_done);
}
int pg_sharded_device_errors(pg_sharded_graph* s) {
  int e = 0;
  for (pg_graph* g : s->shards) { const int r = pg_graph_device_errors(g); if (r < 0) return r; e |= r; }
  return e;
}

}  // extern "C"
