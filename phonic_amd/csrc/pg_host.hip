// Host side of libphonic_gpu.so: the C ABI of include/phonic_gpu.h above the gfx950 kernels.
//
// Mirrors the reference's control plane only as far as the hot path needs it: graph construction as
// `Player` does it (src/player.rs:519-602,773-822,893-939), the mixer's message/event bookkeeping
// (src/source/mixed.rs:294-499,679-712, src/utils/event.rs) — all integer sample-time arithmetic, done
// here on the host and handed to the kernels as per-launch command lists — and the parameter descriptor
// logic (pg_params.h). All per-sample work happens in pg_kernels.hip.
#include <hip/hip_runtime.h>

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

size_t pg_fast_scratch_bytes(uint32_t kind_mask);
hipError_t pg_launch_units(const PgLaunch& L, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t pg_launch_stages(const PgLaunch& L, hipStream_t stream, int single_launch, int lean, int wide, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t pg_launch_mix(const float* unit_out, uint32_t stride, int n_units, float* partial, float* bus, uint32_t n_samples, const PgUnit* units,
                         const int32_t* order, int* audible_out, hipStream_t stream, int n_chunks = 1, size_t chunk_stride = 0);

// ---- errors ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
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
static std::atomic<uint64_t> g_n_alloc{0}, g_n_free{0}, g_n_sync{0}, g_n_blocking_copy{0};
static hipError_t pg_malloc(void** p, size_t bytes) { g_n_alloc++; return hipMalloc(p, bytes); }
static hipError_t pg_host_malloc(void** p, size_t bytes, unsigned flags) { g_n_alloc++; return hipHostMalloc(p, bytes, flags); }
static hipError_t pg_free(void* p) { g_n_free++; return hipFree(p); }
static hipError_t pg_host_free(void* p) { g_n_free++; return hipHostFree(p); }
static hipError_t pg_stream_sync(hipStream_t s) { g_n_sync++; return hipStreamSynchronize(s); }
static hipError_t pg_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { g_n_blocking_copy++; return hipMemcpy(d, s, n, k); }
static hipError_t pg_memset(void* d, int v, size_t n) { g_n_blocking_copy++; return hipMemset(d, v, n); }

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
template <class T>
struct DeviceTable {
  T* d = nullptr;
  T* h = nullptr;  // pinned staging, same capacity
  size_t n = 0, cap = 0;
  int reserve(size_t want) {
    if (want <= cap) return PG_OK;
    size_t ncap = std::max<size_t>(want, cap ? cap * 2 : 64);
    if (d) (void)pg_free(d);
    if (h) (void)pg_host_free(h);
    d = nullptr; h = nullptr; cap = 0;
    HIP_TRY(pg_malloc((void**)&d, ncap * sizeof(T)));
    HIP_TRY(pg_host_malloc((void**)&h, ncap * sizeof(T), hipHostMallocDefault));
    cap = ncap;
    return PG_OK;
  }
  int upload_async(const std::vector<T>& v, hipStream_t s) {
    if (v.size() > cap) return set_error(PG_ERR_STATE, "device table capacity was not reserved by the mutating call");
    if (!v.empty()) {
      memcpy(h, v.data(), v.size() * sizeof(T));
      HIP_TRY(hipMemcpyAsync(d, h, v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    n = v.size();
    return PG_OK;
  }
  void release() { if (d) (void)pg_free(d); if (h) (void)pg_host_free(h); d = nullptr; h = nullptr; n = cap = 0; }
};

static size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

// ---- smoother construction (constructors of src/utils/smoothing.rs + SmoothedParameterValue) --------------
static PgSmooth make_smooth(const ParamSpec& p, float value, uint32_t sr) {
  PgSmooth s;
  memset(&s, 0, sizeof s);
  s.comp = 44100.0f / (float)sr;  // set_sample_rate
  s.current = s.target = value;
  switch (p.smooth) {
    case S_LIN:  // LinearSmoothedValue::default().with_step(step); init(v); set_sample_rate(sr)  :257-310,394-401
      s.kind = SM_LIN; s.a = p.smooth_arg; s.b = s.a * s.comp; s.pending = 0; break;
    case S_SPRING:  // SpringSmoothedValue::default().with_duration(d)  :434-474
      s.kind = SM_SPRING; s.a = 5.5f / (float)(size_t)p.smooth_arg; s.b = 0.0f; break;
    default:  // ExponentialSmoothedValue (inertia 1/256 unless stated)  :139,156-170
      s.kind = SM_EXP; s.a = p.smooth_arg > 0.0f ? p.smooth_arg : 1.0f / 256.0f; break;
  }
  return s;
}

// distortion LUT (DistortionType::rms_compensation, src/effect/distortion.rs:88-122): pure function of the
// shaper, built once per device on the host with the same f32 arithmetic as the reference's `new()`.
static float h_dist_shape(int type, float sample, float drive) {
  const float MAX_DRIVE = 4.0f, PI32 = 3.14159274101257324f;
  float t = drive / MAX_DRIVE;
  switch (type) {
    case 0: { float gain = 1.0f + (t * t) * 14.0f; float x = sample * gain; if (x >= 1.0f) return 1.0f; if (x > -1.0f) { if (gain <= 1.0f) return sample; return (3.0f / 2.0f) * (x - (x * x * x) / 3.0f); } return -1.0f; }
    case 1: { float gain = 1.0f + (t * t) * 24.0f; float th = 1.0f / gain; return h_clamp(sample, -th, th) * gain; }
    case 2: { float curve = 0.6f * (t * t) + 0.4f * t; float gain = 1.0f + curve * 19.0f; float dc = std::exp((0.1f * sample) / (0.0253f * 1.68f)) - 1.0f; return 2.0f / PI32 * std::atan(dc * gain); }
    case 3: { float gain = 1.0f + (1.0f - std::exp(-3.0f * t)) * 29.0f; float a = sample * gain; float s = (a < 0.0f) ? -1.0f * (1.0f - std::exp(-std::fabs(a))) : 1.0f * (1.0f - std::exp(-std::fabs(a))); return 1.5f * (s + std::fabs(s)); }
    default: { float gain = 1.0f + (t * t) * 3.0f; float x = sample * gain; float th = 1.0f / gain; if (x > th || x < -th) return std::fabs(std::fmod(std::fabs(x - th), th * 4.0f) - th * 2.0f) - th; return x; }
  }
}
static void build_dist_luts(float* luts /*[5][256]*/) {
  const int N = 256;
  static const float PARTIALS[5][2] = {{1.0f, 0.60f}, {2.7f, 0.25f}, {5.3f, 0.10f}, {9.1f, 0.03f}, {14.6f, 0.02f}};
  float partials_peak = 0.0f;
  for (int p = 0; p < 5; ++p) partials_peak += PARTIALS[p][1];
  for (int type = 0; type < 5; ++type)
    for (int li = 0; li < 256; ++li) {
      float drive = (float)li / 255.0f * 4.0f;
      float in_sq = 0.0f, out_sq = 0.0f;
      for (int i = 0; i < N; ++i) {
        float t = 6.28318548202514648f * ((float)i + 0.5f) / (float)N;
        float s = 0.0f;
        for (int p = 0; p < 5; ++p) s += PARTIALS[p][1] * std::sin(PARTIALS[p][0] * t);
        float sample = s / partials_peak;
        in_sq += sample * sample;
        float o = h_dist_shape(type, sample, drive);
        out_sq += o * o;
      }
      float in_rms = std::sqrt(in_sq / (float)N), out_rms = std::sqrt(out_sq / (float)N);
      luts[type * 256 + li] = (out_rms > 1e-10f) ? in_rms / out_rms : 1.0f;
    }
}
// vibrato rotation table of the reverb fast path: cos/sin(j * depth_i * vib_speed), j = 0..128, for the eight lines
// (depths: src/effect/reverb.rs:137-144; increment depth*speed: reverb.rs:601-603). Read-only, shared by all instances.
static std::mutex g_tables_mutex;  // the shared read-only tables are built once per device, from whichever thread gets there first;
static std::map<int, double*> g_vib_tabs;  // they live until the process ends (a few KB per device)
static int get_vib_tab(int device, const double** out) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  auto it = g_vib_tabs.find(device);
  if (it == g_vib_tabs.end()) {
    static const double depths[8] = {0.003251, 0.002999, 0.002917, 0.002749, 0.002503, 0.002423, 0.002146, 0.002088};
    std::vector<double> h(8 * 129 * 2);
    for (int i = 0; i < 8; ++i) {
      const double d = depths[i] * 0.1;
      for (int j = 0; j <= 128; ++j) { h[(i * 129 + j) * 2] = std::cos((double)j * d); h[(i * 129 + j) * 2 + 1] = std::sin((double)j * d); }
    }
    double* dp = nullptr;
    HIP_TRY(pg_malloc((void**)&dp, h.size() * 8));
    HIP_TRY(pg_memcpy(dp, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    it = g_vib_tabs.emplace(device, dp).first;
  }
  *out = it->second;
  return PG_OK;
}
static std::map<int, float*> g_dist_luts;  // per device
static int get_dist_luts(int device, const float** out) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  auto it = g_dist_luts.find(device);
  if (it == g_dist_luts.end()) {
    std::vector<float> h(5 * 256);
    build_dist_luts(h.data());
    float* d = nullptr;
    HIP_TRY(pg_malloc((void**)&d, h.size() * 4));
    HIP_TRY(pg_memcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    it = g_dist_luts.emplace(device, d).first;
  }
  *out = it->second;
  return PG_OK;
}

// ---- effect instance: host mirror + construction of the device state ---------------------------------------
struct HostFx {
  int kind = 0;
  std::vector<float> init_raw;   // raw value per parameter after `new()/with_parameters`
  std::vector<float> target;     // shadow of the targets (for nothing on the hot path; introspection only)
  bool with_params = false;
  bool has_seeds = false;
  uint32_t fpd_l = 16386, fpd_r = 16386;
  double vib[16] = {0};
  void* d_mem = nullptr;         // delay-line memory owned by this effect
  size_t d_mem_bytes = 0;
  int last_mixer = -1;           // graph effects: the mixer the effect belonged to when it was removed (its late events stay that mixer's events)
};

static int host_fx_from_init(int kind, const pg_effect_init* init, HostFx& h) {
  if (kind < 0 || kind >= PG_FX_KIND_COUNT) return set_error(PG_ERR_PARAMETER, "unknown effect kind %d", kind);
  const KindInfo& k = KINDS[kind];
  h.kind = kind;
  h.init_raw.resize(k.n_params);
  for (int i = 0; i < k.n_params; ++i) h.init_raw[i] = k.params[i].def;
  if (init) {
    if (init->n_params > PG_MAX_INIT_PARAMS) return set_error(PG_ERR_PARAMETER, "too many init parameters");
    for (uint32_t i = 0; i < init->n_params; ++i) {
      int pi = find_param(kind, init->fourcc[i]);
      if (pi < 0) return set_error(PG_ERR_PARAMETER, "Unknown parameter: 0x%08x for effect '%s'", init->fourcc[i], k.name);
      const ParamSpec& p = k.params[pi];
      float v = init->value[i];
      if (p.type == PG_PARAM_FLOAT && !(v >= p.min && v <= p.max)) return set_error(PG_ERR_PARAMETER, "Value out of bounds for '%s'", p.name);
      if (p.type == PG_PARAM_ENUM && !((int)v >= 0 && (int)v < p.n_values)) return set_error(PG_ERR_PARAMETER, "Invalid enum index for '%s'", p.name);
      if (kind == PG_FX_DELAY && pi == P_DELAY_LFO_SHAPE && (int)v >= 5)
        return set_error(PG_ERR_PARAMETER, "LFO shapes Random/Smooth Random draw from an OS-seeded RNG in the reference and are not supported");
      h.init_raw[pi] = v;
      h.with_params = true;
    }
    if (init->has_reverb_seeds) {
      h.has_seeds = true;
      h.fpd_l = init->reverb_fpd_l; h.fpd_r = init->reverb_fpd_r;
      memcpy(h.vib, init->reverb_vib_phase, sizeof h.vib);
    }
  }
  h.target = h.init_raw;
  return PG_OK;
}

// State of the effect right after `Effect::initialize(sample_rate, 2, max_frames)`.
static int build_fx_device_state(HostFx& h, uint32_t sr, int device, bool standalone, PgFx& fx) {
  memset(&fx, 0, sizeof fx);
  const KindInfo& k = KINDS[h.kind];
  const ParamSpec* P = k.params;
  const std::vector<float>& v = h.init_raw;
  fx.kind = h.kind;
  fx.sample_rate = sr;
  fx.bypassed = 1;                         // EffectProcessor::new  effect.rs:26-33
  fx.tail_counter = 0;
  fx.silence_counter = PG_USIZE_MAX;
  fx.standalone = standalone ? 1 : 0;
  auto alloc = [&](size_t bytes) -> int {
    h.d_mem_bytes = bytes;
    HIP_TRY(pg_malloc(&h.d_mem, bytes));
    HIP_TRY(pg_memset(h.d_mem, 0, bytes));
    return PG_OK;
  };
  switch (h.kind) {
    case PG_FX_GAIN: {  // gain.rs:123-141
      PgGain& g = fx.u.gain;
      g.gain = make_smooth(P[0], v[0], sr);
      g.dc_mode = (int)v[1];
      double hz = g.dc_mode == 1 ? 1.0 : (g.dc_mode == 3 ? 20.0 : 5.0);  // unwrap_or(Default)
      for (int c = 0; c < 2; ++c) { g.dc[c].x1 = g.dc[c].y1 = 0.0; g.dc[c].r = dc_r(hz, sr); }
    } break;
    case PG_FX_PANNING: {
      PgPan& p = fx.u.pan;
      p.pan = make_smooth(P[0], v[0], sr);
      p.width = make_smooth(P[1], v[1], sr);
      p.invert_l = v[2] != 0.0f; p.invert_r = v[3] != 0.0f;
    } break;
    case PG_FX_FILTER: {  // filter.rs:87-115,141-164
      PgFilter& f = fx.u.filter;
      f.type = (int)v[0];
      f.cutoff = make_smooth(P[1], v[1], sr);
      f.q = make_smooth(P[2], v[2], sr);
      memset(&f.coef, 0, sizeof f.coef);
      biquad_set(f.coef, 0, 44100, 22050.0f, 0.707f, 0.0f);  // new(): coefficients for 44100 Hz (!)
      if (h.with_params) {
        float c = clampf(v[1], 20.0f, 44100.0f / 2.0f);
        if (!biquad_set(f.coef, filter_to_biquad(f.type), 44100, c, v[2], 0.0f)) return set_error(PG_ERR_PARAMETER, "Invalid filter parameters");
      }
      float c = clampf(f.coef.cutoff, 20.0f, (float)sr / 2.0f);  // initialize(): set_cutoff only
      if (f.coef.cutoff != c) { f.coef.cutoff = c; biquad_apply(f.coef); }
    } break;
    case PG_FX_EQ5: {  // eq5.rs:152-170,268-294
      PgEq5& e = fx.u.eq5;
      for (int i = 0; i < 5; ++i) {
        e.gains[i] = make_smooth(P[i * 3], v[i * 3], sr);
        e.freqs[i] = make_smooth(P[i * 3 + 1], v[i * 3 + 1], sr);
        e.bws[i] = make_smooth(P[i * 3 + 2], v[i * 3 + 2], sr);
        memset(&e.coef[i], 0, sizeof e.coef[i]);
        float c = clampf(e.freqs[i].current, 20.0f, (float)sr / 2.0f);
        int bt = i == 0 ? 7 : (i == 4 ? 8 : 6);
        if (!biquad_set(e.coef[i], bt, sr, c, e.bws[i].current, e.gains[i].current)) return set_error(PG_ERR_PARAMETER, "Invalid EQ parameters");
      }
    } break;
    case PG_FX_DELAY: {  // delay.rs:273-332
      PgDelay& d = fx.u.delay;
      d.mode = (int)v[P_DELAY_MODE]; d.filter_type = (int)v[P_DELAY_FTYPE]; d.lfo_shape = (int)v[P_DELAY_LFO_SHAPE];
      d.delay_time = make_smooth(P[P_DELAY_TIME], v[P_DELAY_TIME], sr);
      d.feedback = make_smooth(P[P_DELAY_FEEDBACK], v[P_DELAY_FEEDBACK], sr);
      d.cutoff = make_smooth(P[P_DELAY_CUTOFF], v[P_DELAY_CUTOFF], sr);
      d.drive = make_smooth(P[P_DELAY_DRIVE], v[P_DELAY_DRIVE], sr);
      d.wet = make_smooth(P[P_DELAY_WET], v[P_DELAY_WET], sr);
      d.width = make_smooth(P[P_DELAY_WIDTH], v[P_DELAY_WIDTH], sr);
      d.lfo_rate = make_smooth(P[P_DELAY_LFO_RATE], v[P_DELAY_LFO_RATE], sr);
      d.d_time = make_smooth(P[P_DELAY_D_TIME], v[P_DELAY_D_TIME], sr);
      d.d_feedback = make_smooth(P[P_DELAY_D_FEEDBACK], v[P_DELAY_D_FEEDBACK], sr);
      d.d_filter = make_smooth(P[P_DELAY_D_FILTER], v[P_DELAY_D_FILTER], sr);
      size_t max_delay_samples = (size_t)std::ceil((4000.0f + 50.0f) * (float)sr / 1000.0f);
      size_t frames = next_pow2(max_delay_samples + 4);
      int rc = alloc(frames * 8 * 2);
      if (rc) return rc;
      d.line[0] = (double*)h.d_mem; d.line[1] = d.line[0] + frames;
      d.mask = (uint32_t)(frames - 1);
      memset(&d.coef, 0, sizeof d.coef);
      if (!svf_set(d.coef, d.filter_type, sr, clampf(d.cutoff.target, 20.0f, (float)sr / 2.0f), 0.302f)) return set_error(PG_ERR_PARAMETER, "Invalid delay filter");
      d.lfo.phase = 0.0f; d.lfo.phase_inc = (float)((double)d.lfo_rate.target / (double)sr); d.lfo.waveform = d.lfo_shape;
      for (int c = 0; c < 2; ++c) { d.dc[c].x1 = d.dc[c].y1 = 0.0; d.dc[c].r = dc_r(5.0, sr); }
    } break;
    case PG_FX_REVERB: {  // reverb.rs:94-151,391-407
      PgReverb& r = fx.u.reverb;
      r.room = make_smooth(P[0], v[0], sr);
      r.wet = make_smooth(P[1], v[1], sr);
      r.fpd_l = h.fpd_l; r.fpd_r = h.fpd_r;
      static const size_t sizes[8] = {8111, 7511, 7311, 6911, 6311, 6111, 5511, 4911};
      static const double depths[8] = {0.003251, 0.002999, 0.002917, 0.002749, 0.002503, 0.002423, 0.002146, 0.002088};
      static const size_t apsizes[4] = {4511, 4311, 3911, 3311};
      size_t total = 0;
      for (int i = 0; i < 8; ++i) total += (sizes[i] + 1) * 2;
      for (int i = 0; i < 4; ++i) total += apsizes[i] * 2;
      total += 4096 * 2;  // DelayLine::new(3111) -> next_power_of_two
      int rc = alloc(total * 8);
      if (rc) return rc;
      double* p = (double*)h.d_mem;
      for (int i = 0; i < 8; ++i) {
        PgReverbLine& l = r.line[i];
        l.buf = p; p += (sizes[i] + 1) * 2;
        l.frames = (uint32_t)(sizes[i] + 1); l.count = 1; l.delay = 1;
        l.depth = depths[i];
        l.vib_phase[0] = h.vib[i * 2]; l.vib_phase[1] = h.vib[i * 2 + 1];
      }
      for (int i = 0; i < 4; ++i) { r.ap[i].buf = p; p += apsizes[i] * 2; r.ap[i].frames = (uint32_t)apsizes[i]; r.ap[i].delay = 0; r.ap[i].write_pos = 0; }
      r.pre = p; r.pre_mask = 4095; r.pre_write_pos = 0;
      rc = get_vib_tab(device, &r.vib_tab);
      if (rc) return rc;
    } break;
    case PG_FX_CHORUS: {  // chorus.rs:263-309
      PgChorus& c = fx.u.chorus;
      c.rate = make_smooth(P[P_CHORUS_RATE], v[P_CHORUS_RATE], sr);
      c.depth = make_smooth(P[P_CHORUS_DEPTH], v[P_CHORUS_DEPTH], sr);
      c.feedback = make_smooth(P[P_CHORUS_FEEDBACK], v[P_CHORUS_FEEDBACK], sr);
      c.delay = make_smooth(P[P_CHORUS_DELAY], v[P_CHORUS_DELAY], sr);
      c.wet = make_smooth(P[P_CHORUS_WET], v[P_CHORUS_WET], sr);
      c.phase = make_smooth(P[P_CHORUS_PHASE], v[P_CHORUS_PHASE], sr);
      c.filter_type = (int)v[P_CHORUS_FTYPE];
      c.freq = make_smooth(P[P_CHORUS_FREQ], v[P_CHORUS_FREQ], sr);
      c.res = make_smooth(P[P_CHORUS_RES], v[P_CHORUS_RES], sr);
      c.lfo_range = 256.0f * ((float)sr / 44100.0f);
      size_t max_depth = (size_t)std::ceil(c.lfo_range);
      size_t max_delay = (size_t)std::ceil(100.0f * (float)sr / 1000.0f);
      size_t frames = next_pow2(2 + max_delay + 2 * max_depth + 1);
      int rc = alloc(frames * 8 * 2);
      if (rc) return rc;
      c.line[0] = (double*)h.d_mem; c.line[1] = c.line[0] + frames;
      c.mask = (uint32_t)(frames - 1);
      memset(&c.coef, 0, sizeof c.coef);
      if (!svf_set(c.coef, c.filter_type, sr, clampf(c.freq.target, 20.0f, (float)sr / 2.0f), c.res.target)) return set_error(PG_ERR_PARAMETER, "Invalid chorus filter");
      c.current_phase = 0.0;  // reset() :201-221
      for (int i = 0; i < 2; ++i) { c.osc[i].phase = 0.0f; c.osc[i].waveform = 0; lfo_set_rate(c.osc[i], sr, (double)c.rate.current); }
      lfo_set_phase_degrees(c.osc[0], (float)c.current_phase);
      lfo_set_phase_degrees(c.osc[1], (float)(c.current_phase + (double)c.phase.current));
    } break;
    case PG_FX_COMPRESSOR: {  // compressor.rs:196-228
      PgComp& c = fx.u.comp;
      c.threshold = v[0]; c.ratio = v[1]; c.knee = v[2]; c.attack = v[3]; c.release = v[4];
      c.makeup = make_smooth(P[5], v[5], sr);
      c.lookahead = v[6];
      c.env_attack = env_coeff(c.attack, sr); c.env_release = env_coeff(c.release, sr);
      c.env_current = c.ratio >= 20.0f ? -120.0f : 0.0f;
      size_t maxf = next_pow2((size_t)std::ceil(0.2f * (float)sr) + 1);
      int rc = alloc(maxf * 2 * 8);
      if (rc) return rc;
      c.line = (double*)h.d_mem; c.line_frames = (uint32_t)maxf;
      c.delay_frames = (uint32_t)f2u64(std::ceil(c.lookahead * (float)sr));
      c.mask = c.delay_frames > 0 ? (uint32_t)(next_pow2(c.delay_frames) - 1) : 0;
      c.write_pos = 0; c.peak_pos = 0; c.peak_value = 0.0;
    } break;
    case PG_FX_GATE: {  // gate.rs:122-145
      PgGate& g = fx.u.gate;
      g.threshold = v[0]; g.attack = v[1]; g.hold = v[2]; g.release = v[3]; g.range = v[4];
      g.env_attack = env_coeff(g.attack, sr); g.env_release = env_coeff(g.release, sr);
      g.env_current = -120.0f; g.hold_counter = 0; g.gate_gain_db = g.range;
      g.attack_coeff = std::exp(-1.0f / (g.attack * (float)sr));
      g.release_coeff = std::exp(-1.0f / (g.release * (float)sr));
    } break;
    default: {  // distortion.rs:232-256,314-324
      PgDist& d = fx.u.dist;
      d.type = (int)v[0];
      d.drive = make_smooth(P[1], v[1], sr);
      d.mix = make_smooth(P[2], v[2], sr);
      int rc = get_dist_luts(device, &d.luts);
      if (rc) return rc;
    } break;
  }
  return PG_OK;
}

// ---- the graph --------------------------------------------------------------------------------------------
struct Event {  // MixerEvent (src/source/mixed.rs:47-109) resolved to a device command
  uint64_t sample_time;
  uint64_t seq;
  PgCmd cmd;  // unit/frame filled per launch
  int mixer;  // owning mixer (0 = main)
};

struct HostVoice { int mixer; int dev_index; uint64_t start_time; void* d_pcm; void* d_stage; bool outer; };
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
struct Level { int off = 0, cnt = 0, n_staged = 0, n_staged_wide = 0, n_static_defer = 0; };

struct pg_graph {
  int device = 0;
  uint32_t sample_rate = 48000, channels = 2;
  size_t max_frames = 4096;
  hipStream_t stream = nullptr;
  bool failed = false;  // sticky: GuardedSource semantics
  int fast = 1;
  bool wide = false;  // some sub-mixer chain holds Filter / Eq5 / Distortion: use the wide fast-kernel variant
  uint32_t fast_kind_mask = 0;  // effect kinds held by the units the fast kernels render: sizes their LDS arena (pg_fast_scratch_bytes)
  int timing_period = 0;   // time every n-th round with a hipEvent pair (0: never, the default); pg_graph_set_timing_period creates the pairs
  int staged_mode = 1;     // [Gain|Panning]* -> Reverb units: 1 = staged single launch (pg_stage_fused_kernel), 2 = one launch per stage, 0 = fused fast kernel
  int n_staged = 0;        // graph units eligible for the staged pipeline (levels 1 and 2)
  int n_staged_wide = 0;   // ... of level 2 (leading effects beyond Gain / Panning)
  int n_static_defer = 0;  // graph units that always run on the generic kernel
  double* d_stage = nullptr;  // [stage_rows][PG_STAGE_BUF_DOUBLES]
  DeviceTable<int4> d_slot_info;  // per launch slot: {unit slot, first voice, last effect, voices}
  DeviceTable<int2> d_child_rows; // nested sub-mixers: {output row, unit slot}, indexed by PgUnit::child_off
  DeviceTable<PgUnit> d_topo;     // topology fields of every unit, patched into d_units by pg_patch_units_kernel
  std::vector<Level> levels;    // deepest first
  uint64_t defer_phase = 0;     // one deferral hand-shake per level launch (two counters, alternating)
  int32_t* d_defer = nullptr;  // [2 counters][defer_rows slots]: compact list of the units the fast kernels deferred
  size_t defer_rows = 0;
  size_t stage_rows = 0;
  bool defer_bus = false;
  size_t max_blocks = 1;        // blocks of max_frames one launch sequence may render (pg_graph_set_max_blocks_per_launch); sizes d_unit_out
  size_t unit_out_blocks = 0;   // ... as allocated
  int32_t* d_error = nullptr;   // sticky consistency flags of the kernels (PG_DEVERR_*)
  // host mirrors
  std::vector<HostMixer> mixers;        // [0] = main
  std::vector<HostVoice> voices;
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
  int* d_audible = nullptr;
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
};

static int graph_fail(pg_graph* g, int code) { g->failed = true; return code; }

// Calls that change the graph (add_* / remove_* / move_* / mode switches) are not real-time calls: they first wait for everything the
// last write enqueued — on the graph's own stream and on the caller's stream the last write used — so that no launch in flight reads a
// table that is about to be re-uploaded, re-allocated or patched.
static int graph_quiesce(pg_graph* g) {
  (void)hipSetDevice(g->device);
  HIP_TRY(pg_stream_sync(g->stream));
  if (g->last_stream && g->last_stream != g->stream) HIP_TRY(pg_stream_sync(g->last_stream));
  g->cmds_since_sync = 0;
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
__global__ void pg_status_kernel(const PgVoice* voices, const int32_t* idx, int n, float* status) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int active = 0;
    for (int i = 0; i < n; ++i) active += voices[idx[i]].active ? 1 : 0;
    ((int*)status)[0] = active;
  }
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
    if (!mx.children.empty()) { u.static_defer = 1; u.staged = 0; }  // sums its sub-mixers' rows first: exact serial kernel
    for (int v : mx.voices) if (g->voices[v].outer) { u.static_defer = 1; u.staged = 0; }  // ResampledSource staging: exact serial kernel
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
    u.static_defer = g->voices[v].outer ? 1 : 0;
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
  }
  g->n_staged = 0; g->n_staged_wide = 0; g->n_static_defer = 0;
  for (Level& lv : g->levels) {
    lv.n_staged = lv.n_staged_wide = lv.n_static_defer = 0;
    for (int i = lv.off; i < lv.off + lv.cnt; ++i) {
      const PgUnit& u = topo[g->order[i]];
      lv.n_staged += u.staged ? 1 : 0; lv.n_staged_wide += u.staged == 2 ? 1 : 0; lv.n_static_defer += u.static_defer ? 1 : 0;
    }
    g->n_staged += lv.n_staged; g->n_staged_wide += lv.n_staged_wide; g->n_static_defer += lv.n_static_defer;
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
  if ((size_t)std::max(g->n_graph_units, 1) > g->unit_out_rows || g->max_blocks != g->unit_out_blocks)
    return set_error(PG_ERR_STATE, "per-unit buffers were not reserved by the mutating call");
  g->topo_dirty = false;
  g->last_change_round = g->launch_counter;  // the patch kernel marked every unit: the generic kernel must look at them again
  return PG_OK;
}

// Device capacity for the graph as the host mirror describes it now. Called at the end of every mutating call (after graph_quiesce):
// this is where the library allocates — grow-by-doubling — so that rebuild_topology and everything else inside write never does.
static int graph_reserve(pg_graph* g) {
  int rc;
  const size_t n_units = g->h_units.size(), n_voices = g->voices.size(), n_fx = g->fx.size(), n_mixers = g->mixers.size();
  if ((rc = g->d_topo.reserve(n_units)) || (rc = g->d_order.reserve(n_units)) || (rc = g->d_slot_info.reserve(n_units)) ||
      (rc = g->d_voice_index.reserve(n_voices)) || (rc = g->d_fx_index.reserve(n_fx)) || (rc = g->d_child_rows.reserve(n_mixers)))
    return rc;
  if (!g->d_cmd_ring) {
    HIP_TRY(pg_malloc((void**)&g->d_cmd_ring, PG_CMD_RING * sizeof(PgCmd)));
    HIP_TRY(pg_host_malloc((void**)&g->h_cmd_ring, PG_CMD_RING * sizeof(PgCmd), hipHostMallocDefault));
  }
  // per-unit output rows (one table of rows per block of a super-block launch), deferral list, stage hand-over, mixer partials
  const size_t rows = std::max<size_t>(n_units, 1);
  if (rows > g->unit_out_rows || g->max_blocks != g->unit_out_blocks) {
    if (g->d_unit_out) (void)pg_free(g->d_unit_out);
    g->d_unit_out = nullptr;
    size_t nr = rows > g->unit_out_rows ? std::max(rows, g->unit_out_rows * 2) : g->unit_out_rows;
    HIP_TRY(pg_malloc((void**)&g->d_unit_out, (nr * g->stride * g->max_blocks + 4) * sizeof(float)));  // +4: the mixer sum reads whole float4s (odd max_frames)
    g->unit_out_rows = nr;
    g->unit_out_blocks = g->max_blocks;
  }
  if (rows > g->defer_rows) {
    if (g->d_defer) (void)pg_free(g->d_defer);
    g->d_defer = nullptr;
    const size_t nr = std::max(rows, g->defer_rows * 2);
    HIP_TRY(pg_malloc((void**)&g->d_defer, (2 + nr) * sizeof(int32_t)));
    HIP_TRY(pg_memset(g->d_defer, 0, (2 + nr) * sizeof(int32_t)));
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
const char* pg_effect_kind_name(int kind) { return (kind >= 0 && kind < PG_FX_KIND_COUNT) ? KINDS[kind].name : nullptr; }
int pg_effect_kind_weight(int kind) { return (kind >= 0 && kind < PG_FX_KIND_COUNT) ? KINDS[kind].weight : -1; }
int pg_effect_kind_param_count(int kind) { return (kind >= 0 && kind < PG_FX_KIND_COUNT) ? KINDS[kind].n_params : -1; }
int pg_effect_kind_param(int kind, int index, pg_param_desc* out) {
  if (kind < 0 || kind >= PG_FX_KIND_COUNT || index < 0 || index >= KINDS[kind].n_params) return set_error(PG_ERR_NOT_FOUND, "no such parameter");
  const ParamSpec& p = KINDS[kind].params[index];
  out->fourcc = p.fourcc; out->type = p.type; out->min = p.min; out->max = p.max; out->default_value = p.def;
  out->scaling = p.scaling; out->scaling_arg0 = p.sa; out->scaling_arg1 = p.sb; out->n_values = p.n_values; out->name = p.name;
  return PG_OK;
}
void pg_voice_options_default(pg_voice_options* o) {  // FilePlaybackOptions::default()  file.rs:94-112
  memset(o, 0, sizeof *o);
  o->volume = 1.0f; o->panning = 0.0f; o->speed = 1.0;
  o->fade_in_seconds = 0.0f; o->fade_out_seconds = 0.05f;
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
  if (pg_malloc((void**)&g->d_bus, (g->stride + 4) * sizeof(float)) != hipSuccess || pg_malloc((void**)&g->d_audible, 16) != hipSuccess ||
      pg_host_malloc((void**)&g->h_pinned, (g->stride + 4) * sizeof(float), hipHostMallocDefault) != hipSuccess) {
    set_error(PG_ERR_DEVICE, "device allocation failed");
    return nullptr;
  }
  (void)pg_memset(g->d_audible, 0, 16);
  if (pg_malloc((void**)&g->d_error, 16) == hipSuccess) (void)pg_memset(g->d_error, 0, 16); else g->d_error = nullptr;
  if (pg_host_malloc((void**)&g->h_feedback, 64, hipHostMallocMapped) == hipSuccess) {
    *g->h_feedback = ~0ull;  // nothing reported yet
    if (hipHostGetDevicePointer((void**)&g->d_feedback, g->h_feedback, 0) != hipSuccess) g->d_feedback = nullptr;
  }
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
  for (auto& v : g->voices) { if (v.d_pcm) (void)pg_free(v.d_pcm); if (v.d_stage) (void)pg_free(v.d_stage); }
  for (auto& f : g->fx) if (f->d_mem) (void)pg_free(f->d_mem);
  g->d_units.release(); g->d_voices.release(); g->d_fx.release(); g->d_voice_index.release(); g->d_fx_index.release(); g->d_order.release();
  g->d_sched.release(); g->d_slot_info.release(); g->d_child_rows.release(); g->d_topo.release();
  if (g->d_cmd_ring) (void)pg_free(g->d_cmd_ring);
  if (g->d_cmd_overflow) (void)pg_free(g->d_cmd_overflow);
  if (g->h_cmd_ring) (void)pg_host_free(g->h_cmd_ring);
  if (g->d_unit_out) (void)pg_free(g->d_unit_out);
  if (g->d_partial) (void)pg_free(g->d_partial);
  if (g->d_stage) (void)pg_free(g->d_stage);
  if (g->d_defer) (void)pg_free(g->d_defer);
  if (g->d_bus) (void)pg_free(g->d_bus);
  if (g->d_audible) (void)pg_free(g->d_audible);
  if (g->d_error) (void)pg_free(g->d_error);
  if (g->h_pinned) (void)pg_host_free(g->h_pinned);
  if (g->h_feedback) (void)pg_host_free(g->h_feedback);
  for (auto& e : g->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
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
    if (g->voices[v].mixer < 0) continue;
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
      if ((size_t)v < mx.remove_voice_limit && g->voices[v].start_time > pos) {
        mx.messages.erase(std::remove_if(mx.messages.begin(), mx.messages.end(), [v](const PgCmd& c) { return c.param == v; }), mx.messages.end());
        g->voices[v].mixer = -1;
        g->voice_alive_tab.set((size_t)v, 0);
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
    for (int v : mx.voices) { g->voices[v].mixer = -1; g->voice_alive_tab.set((size_t)v, 0); }
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
  g->voices.push_back(HostVoice{mixer_id, dev_index, opt->start_time, d_pcm, d_stage, inner_rate != g->sample_rate});
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

int pg_graph_schedule_param(pg_graph* g, int effect_id, uint32_t fourcc, float value, int is_normalized, uint64_t sample_time) {
  const int kind = fx_kind_of(g, effect_id);
  if (kind < 0) return set_error(PG_ERR_NOT_FOUND, "Effect with id %d not found", effect_id);
  int pi = find_param(kind, fourcc);
  if (pi < 0) return set_error(PG_ERR_PARAMETER, "Unknown parameter: 0x%08x for effect '%s'", fourcc, KINDS[kind].name);
  float raw;
  if (!resolve_update(KINDS[kind].params[pi], value, is_normalized != 0, raw)) return PG_OK;  // logged + ignored in the reference
  if (kind == PG_FX_DELAY && pi == P_DELAY_LFO_SHAPE && (int)raw >= 5)
    return set_error(PG_ERR_PARAMETER, "LFO shapes Random/Smooth Random are not supported (OS-seeded RNG in the reference)");
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
  return voice_message(g, voice_id, pgc::CT_VOICE_SPEED, glide, speed, sample_time);
}
int pg_graph_seek_voice(pg_graph* g, int voice_id, double position_seconds, uint64_t sample_time) {
  if (!(position_seconds >= 0.0)) return set_error(PG_ERR_PARAMETER, "seek position must be >= 0");
  return voice_message(g, voice_id, pgc::CT_VOICE_SEEK, 0.0f, position_seconds, sample_time);
}
int pg_graph_stop_voice(pg_graph* g, int voice_id, uint64_t sample_time) {  // MixerMessage::StopSource (mixed.rs:389-400): not an event
  return voice_message(g, voice_id, pgc::CT_VOICE_STOP, 0.0f, 0.0, sample_time);
}

// insert_event (src/utils/event.rs:31-38): sorted by sample time, after the events of the same time
// PgCmd::value64 of a parameter update: the coefficients that follow from a new attack / release time of the Compressor's and the Gate's envelope
// follower (EnvelopeFollower::set_attack_time / set_release_time, envelope.rs:27-42) and of the Gate's gain smoothing (gate.rs:80-90), computed here
// with the host's expf — the same call the effect's initial state was built with, and the one the reference makes.
static uint64_t fx_param_aux(int kind, int param, float raw, uint32_t sr) {
  float lo = 0.0f, hi = 0.0f;
  if (kind == PG_FX_COMPRESSOR && (param == P_COMP_ATTACK || param == P_COMP_RELEASE)) lo = env_coeff(raw, sr);
  else if (kind == PG_FX_GATE && (param == P_GATE_ATTACK || param == P_GATE_RELEASE)) { lo = env_coeff(raw, sr); hi = std::exp(-1.0f / (raw * (float)sr)); }
  uint32_t l, h;
  memcpy(&l, &lo, 4); memcpy(&h, &hi, 4);
  return (uint64_t)l | ((uint64_t)h << 32);
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
  const size_t words = (size_t)g->stride * (size_t)n_blocks + 4;
  HIP_TRY(pg_malloc((void**)&nb, words * sizeof(float)));
  if (pg_host_malloc((void**)&np, words * sizeof(float), hipHostMallocDefault) != hipSuccess) { (void)pg_free(nb); return set_error(PG_ERR_DEVICE, "pinned allocation failed"); }
  (void)pg_free(g->d_bus); (void)pg_host_free(g->h_pinned);
  g->d_bus = nb; g->h_pinned = np;
  g->max_blocks = (size_t)n_blocks;
  g->topo_dirty = true;
  return graph_reserve(g);  // the per-unit output table grows here, never inside write
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
  const int n_lean = g->n_staged - g->n_staged_wide;
  const int n_handled = g->staged_mode == 1 ? g->n_staged : n_lean;
  if (g->staged_mode && n_handled > 0) {
    if (g->staged_mode == 2) return n_handled < g->n_graph_units ? "pg_stage1_kernel + pg_stage2_kernel + pg_stage3_kernel + pg_unit_kernel_fast" : "pg_stage1_kernel + pg_stage2_kernel + pg_stage3_kernel";
    if (n_handled < g->n_graph_units) return "pg_stage_fused_kernel + pg_unit_kernel_fast";
    if (n_lean > 0 && g->n_staged_wide > 0) return "pg_stage_fused_kernel + pg_stage_fused_wide_kernel";
    return g->n_staged_wide > 0 ? "pg_stage_fused_wide_kernel" : "pg_stage_fused_kernel";
  }
  return g->wide ? ((g->fast_kind_mask & ((1u << PG_FX_REVERB) | (1u << PG_FX_COMPRESSOR))) ? "pg_unit_kernel_fast_wide" : "pg_unit_kernel_fast_mid") : "pg_unit_kernel_fast";
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
  const unsigned long long fb = *(volatile unsigned long long*)g->h_feedback;
  return fb != ~0ull && (uint32_t)fb == 0u && (int32_t)((uint32_t)(fb >> 32) - (uint32_t)g->last_change_round) >= 0;
}
// A super-block launch sequence renders several blocks of max_frames per workgroup: only in steady state (nobody would render the
// later blocks of a unit that defers itself), in the single-launch kernels, and with no bus chain behind the sum (it needs the
// `audible` result of every block).
static bool graph_super_ok(const pg_graph* g) {
  return g->max_blocks > 1 && g->staged_mode != 2 && (g->defer_bus || g->mixers[0].fx.empty()) && graph_steady(g);
}

// The command list of one round -> a fresh region of the device ring (asynchronous copy from the pinned ring on the round's stream).
static int stage_commands(pg_graph* g, const std::vector<PgCmd>& cmds, hipStream_t stream, const PgCmd** d_out) {
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
  HIP_TRY(hipMemcpyAsync(g->d_cmd_ring + g->cmd_head, g->h_cmd_ring + g->cmd_head, n * sizeof(PgCmd), hipMemcpyHostToDevice, stream));
  *d_out = g->d_cmd_ring + g->cmd_head;
  g->cmd_head += n;
  g->cmds_since_sync += skipped + n;
  return PG_OK;
}

// One launch round: all graph units for frames [t0, t0 + n_chunks * n) -> per-unit rows -> tree sum -> (bus chain) -> d_dst.
// n_chunks > 1 (super-block): n == max_frames, no commands, graph_super_ok().
static int launch_round(pg_graph* g, float* d_dst, uint32_t n, uint64_t t0, hipStream_t stream, bool run_bus, const std::vector<PgCmd>& cmds, int n_chunks = 1) {
  const PgCmd* d_cmds = nullptr;
  if (!cmds.empty()) { int rc = stage_commands(g, cmds, stream, &d_cmds); if (rc) return rc; }
  PgLaunch L;
  memset(&L, 0, sizeof L);
  L.units = g->d_units.d; L.voices = g->d_voices.d; L.fx = g->d_fx.d;
  L.voice_index = g->d_voice_index.d; L.fx_index = g->d_fx_index.d;
  L.cmds = d_cmds; L.n_cmds = (int)cmds.size();
  L.n_frames = n; L.pos = t0; L.sample_rate = g->sample_rate; L.fast = g->fast;
  L.out_stride = g->stride;
  L.rows_base = g->d_unit_out; L.child_rows = g->d_child_rows.d;
  L.diag = g->d_diag;
  L.n_chunks = n_chunks; L.chunk_stride = (uint64_t)g->unit_out_rows * g->stride; L.error_word = g->d_error;
  L.fast_scratch_bytes = (uint32_t)pg_fast_scratch_bytes(g->fast_kind_mask);
  const uint64_t round = g->launch_counter;
  if (!cmds.empty()) g->last_change_round = round;
  L.round = (uint32_t)round; L.host_feedback = g->d_feedback;
  const bool nested = g->levels.size() > 1;
  // Steady state: the generic kernel of an earlier round (not older than the last topology change / command / mode switch) found
  // nothing deferred, and units leave the steady state only through those host-visible events -> the generic launch is skipped.
  // (Graphs with nested sub-mixers always launch it: the parents are rendered there.)
  const bool generic_idle = !nested && cmds.empty() && graph_steady(g);
  // (a super-block runs without the resampler schedule cache: its banks alternate per launch, not per block; voices of a cached
  // class replay their schedule serially, and the cache re-validates itself by key when single-block rounds resume)
  L.sched = n_chunks > 1 ? nullptr : g->d_sched.d; L.sched_bank = (int)(g->launch_counter & 1);
  g->launch_counter++;
  // the event pair costs ~8 us of stream time per round (also when it rides on the dispatch): callers that only need the
  // average can time every n-th round (pg_graph_set_timing_period)
  const bool timed = g->timing_period > 0 && (g->launch_counter % (uint64_t)g->timing_period) == 0 && g->ev_used < g->ev_pool.size() && g->n_graph_units > 0;
  size_t timed_level = 0;  // the level holding most units carries the timing events
  for (size_t li = 1; li < g->levels.size(); ++li) if (g->levels[li].cnt > g->levels[timed_level].cnt) timed_level = li;
  for (size_t li = 0; li < g->levels.size(); ++li) {
    const Level& lv = g->levels[li];
    if (lv.cnt == 0) continue;
    // this level's slice of the per-slot tables: launch slot b of the level = row lv.off + b
    L.n_units = lv.cnt; L.unit_order = g->d_order.d + lv.off;
    L.unit_out = g->d_unit_out + (size_t)lv.off * g->stride;
    L.slot_info = g->d_slot_info.d + lv.off;
    if (g->d_defer) { L.defer_count = g->d_defer + (g->defer_phase & 1); L.defer_reset = g->d_defer + ((g->defer_phase & 1) ^ 1); L.defer_list = g->d_defer + 2; }
    g->defer_phase++;
    const bool timed_here = timed && li == timed_level;
    // The event pair times the launch(es) that do the bulk of this graph's work: the fast / staged kernels, or — when most units
    // hold an effect without a time-parallel path — the generic kernel. When that is a single launch the events ride on the
    // dispatch itself (hipExtLaunchKernel: no marker packets in the stream); several launches are bracketed by event records.
    const bool time_generic = g->fast && lv.n_static_defer * 2 > lv.cnt;
    hipEvent_t e0 = timed_here ? g->ev_pool[g->ev_used].first : nullptr, e1 = timed_here ? g->ev_pool[g->ev_used].second : nullptr;
    if (g->fast) {
      // fast kernel; units it cannot run (ramping parameters, effects without a fast path) are deferred ...
      L.mode = 1; L.wide = g->wide ? ((g->fast_kind_mask & ((1u << PG_FX_REVERB) | (1u << PG_FX_COMPRESSOR))) ? 1 : 2) : 0;
      // reverb-terminated sub-mixers go through the staged kernels; level 2 (wide leading effects) only in the single-launch mode
      const int n_lean = lv.n_staged - lv.n_staged_wide;
      const int n_handled = g->staged_mode == 1 ? lv.n_staged : n_lean;
      const bool staged = g->staged_mode && n_handled > 0 && g->d_stage && n <= 1024;
      const bool lean = staged && n_lean > 0, wide = staged && g->staged_mode == 1 && lv.n_staged_wide > 0;
      const bool fused = !staged || n_handled < lv.cnt;
      const int n_launches = (staged ? (g->staged_mode == 1 ? (int)lean + (int)wide : 3) : 0) + (int)fused;
      const bool ride = timed_here && !time_generic && n_launches == 1;       // one dominant launch: timestamps from its dispatch
      const bool bracket = timed_here && !time_generic && n_launches > 1;
      if (bracket) HIP_TRY(hipEventRecord(e0, stream));
      L.stage_buf = nullptr; L.staged_on = 0;
      if (staged) {
        L.stage_buf = g->d_stage + (size_t)lv.off * PG_STAGE_BUF_DOUBLES; L.staged_on = g->staged_mode == 1 ? 2 : 1;
        HIP_TRY(pg_launch_stages(L, stream, g->staged_mode == 1 ? 1 : 0, lean, wide, ride ? e0 : nullptr, ride ? e1 : nullptr));
      }
      if (fused) HIP_TRY(pg_launch_units(L, stream, ride && !staged ? e0 : nullptr, ride && !staged ? e1 : nullptr));
      if (bracket) HIP_TRY(hipEventRecord(e1, stream));
      L.mode = 2;  // ... to the generic kernel, which walks the list of deferred units (skipped while the host knows the list is empty)
      if (!generic_idle) HIP_TRY(pg_launch_units(L, stream, timed_here && time_generic ? e0 : nullptr, timed_here && time_generic ? e1 : nullptr));
    } else {
      L.mode = 0;
      if (g->d_defer) HIP_TRY(hipMemsetAsync(g->d_defer, 0, 2 * sizeof(int32_t), stream));  // no deferral protocol this round: keep both counters clean
      HIP_TRY(pg_launch_units(L, stream, e0, e1));
    }
  }
  if (timed) { g->ev_blocks[g->ev_used] = (uint32_t)n_chunks; g->ev_used++; }
  L.mode = 0;
  // the main mixer sums the rows of its own sub-mixers and sources: the last level
  const Level& top = g->levels.back();
  HIP_TRY(pg_launch_mix(g->d_unit_out + (size_t)top.off * g->stride, g->stride, top.cnt, g->d_partial, d_dst, n * 2, g->d_units.d, g->d_order.d + top.off, g->d_audible, stream,
                        n_chunks, (size_t)L.chunk_stride));
  if (run_bus && !g->mixers[0].fx.empty()) {
    PgLaunch B = L;
    B.n_units = 1; B.unit_order = nullptr; B.unit_base = g->mixers[0].unit_slot;
    B.bus = d_dst; B.bus_audible = g->d_audible;
    HIP_TRY(pg_launch_units(B, stream));
  }
  return PG_OK;
}

// MixedSource::write of the main mixer (src/source/mixed.rs:659-719)
static size_t graph_write_impl(pg_graph* g, float* d_out, size_t n_samples, uint64_t pos, hipStream_t stream) {
  if (g->failed) return 0;
  if (n_samples % 2 != 0) { set_error(PG_ERR_PARAMETER, "n_samples must be a multiple of the channel count"); return 0; }
  (void)hipSetDevice(g->device);
  // a caller that moves from one stream to another without a graph mutation in between: the tables and rings are ordered per stream
  if (g->last_stream && g->last_stream != stream) { if (pg_stream_sync(g->last_stream) != hipSuccess) { g->failed = true; return 0; } g->cmds_since_sync = 0; }
  g->last_stream = stream;
  drain_control_messages(g);  // process_messages (mixed.rs:294-499)
  apply_remove_pending(g, pos);
  if (g->topo_dirty && rebuild_topology(g, stream)) { g->failed = true; return 0; }
  // "Return early and avoid touching the buffer if there's nothing to do" (:664-670)
  bool any_events = false;
  for (auto& m : g->mixers) any_events |= !m.events.empty();
  bool main_sources_empty = g->main_active_voices == 0;
  bool no_sub_mixers = true;  // self.mixers.is_empty(): sub-mixers that were removed again do not count
  for (size_t m = 1; m < g->mixers.size(); ++m) no_sub_mixers &= g->mixers[m].removed;
  if (main_sources_empty && g->mixers[0].fx.empty() && no_sub_mixers && g->mixers[0].events.empty()) return 0;
  (void)any_events;
  const uint64_t frames = n_samples / 2;
  uint64_t done = 0;
  bool first = true;
  while (done < frames) {
    const uint64_t now = pos + done;
    // main-mixer events due now apply at frame 0 of this round; the next main event bounds the round (:679-693)
    std::vector<PgCmd> cmds;
    HostMixer& main = g->mixers[0];
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
      cmds.push_back(c);
      main.events.erase(main.events.begin());
    }
    uint64_t n = std::min<uint64_t>(frames - done, g->max_frames);
    if (!main.events.empty()) n = std::min<uint64_t>(n, main.events.front().sample_time - now);
    if (n == 0) continue;
    // StopSource messages: processed by process_messages at the start of write (:294-499)
    if (first) {
      for (size_t m = 0; m < g->mixers.size(); ++m) {
        for (PgCmd c : g->mixers[m].messages) {
          c.frame = 0;
          c.unit = m == 0 ? g->source_unit_of_voice[c.param] : g->mixers[m].unit_slot;
          cmds.push_back(c);
        }
        g->mixers[m].messages.clear();
      }
      first = false;
    }
    // Nested sub-mixers: a mixer that splits its block at an event calls its sub-mixers once per segment (mixed.rs:679-712), so
    // every event of a mixer with sub-mixers is also a call boundary (CMD_CALL_SPLIT) for all its descendants. A sub-mixer keeps
    // one result bit per call: the round is bounded so that at most PG_MAX_CALLS - 1 boundaries fall inside it.
    if (g->levels.size() > 1) {
      std::vector<uint64_t> cuts;
      for (size_t m = 1; m < g->mixers.size(); ++m) {
        if (g->mixers[m].children.empty()) continue;
        for (const Event& e : g->mixers[m].events) { if (e.sample_time >= now + n) break; if (e.sample_time > now) cuts.push_back(e.sample_time); }
      }
      std::sort(cuts.begin(), cuts.end());
      cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
      if (cuts.size() > (size_t)(PG_MAX_CALLS - 1)) n = cuts[PG_MAX_CALLS - 1] - now;
      for (size_t m = 1; m < g->mixers.size(); ++m) {
        if (g->mixers[m].children.empty()) continue;
        std::vector<int> desc(g->mixers[m].children);
        for (size_t i = 0; i < desc.size(); ++i) for (int c : g->mixers[desc[i]].children) desc.push_back(c);
        for (const Event& e : g->mixers[m].events) {
          if (e.sample_time >= now + n) break;
          if (e.sample_time <= now) continue;
          for (int d : desc) {
            PgCmd c;
            memset(&c, 0, sizeof c);
            c.type = CMD_CALL_SPLIT; c.unit = g->mixers[d].unit_slot; c.frame = (uint32_t)(e.sample_time - now);
            cmds.push_back(c);
          }
        }
      }
    }
    // sub-mixer events inside [now, now+n): each sub-mixer splits its own block on the device
    for (size_t m = 1; m < g->mixers.size(); ++m) {
      HostMixer& mx = g->mixers[m];
      while (!mx.events.empty() && mx.events.front().sample_time < now + n) {
        PgCmd c = mx.events.front().cmd;
        uint64_t t = mx.events.front().sample_time;
        c.frame = t <= now ? 0u : (uint32_t)(t - now);
        c.unit = mx.unit_slot;
        cmds.push_back(c);
        mx.events.erase(mx.events.begin());
      }
    }
    // bus commands run in the bus launch; everything is sorted by (unit, frame), stable
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
    // Super-block: when the call still spans several whole blocks of max_frames, nothing is scheduled inside them and the graph is in
    // steady state, ONE launch sequence renders them all (the reference's MixedSource::write walks its <= 4096-frame chunks in one call
    // the same way, mixed.rs:679-712); every per-block decision is still taken per block, on the device.
    uint64_t k = 1;
    if (cmds.empty() && n == g->max_frames && graph_super_ok(g)) {
      k = std::min<uint64_t>((frames - done) / g->max_frames, g->max_blocks);
      uint64_t t_next = UINT64_MAX;
      for (const HostMixer& mx : g->mixers) if (!mx.events.empty()) t_next = std::min(t_next, mx.events.front().sample_time);
      if (t_next != UINT64_MAX && t_next < now + k * n) k = (t_next - now) / n;  // (t_next >= now + n: no command fell into this block)
      if (k < 1) k = 1;
    }
    if (launch_round(g, d_out + done * 2, (uint32_t)n, now, stream, !g->defer_bus, cmds, (int)k)) { g->failed = true; return 0; }
    done += n * k;
  }
  return n_samples;
}

size_t pg_graph_write_device(pg_graph* g, float* d_out, size_t n_samples, uint64_t pos_in_frames, void* hip_stream) {
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : g->stream;
  size_t w = graph_write_impl(g, d_out, n_samples, pos_in_frames, s);
  if (!hip_stream && w) { if (pg_stream_sync(g->stream) != hipSuccess) { g->failed = true; return 0; } g->cmds_since_sync = 0; }
  return w;
}

size_t pg_graph_write(pg_graph* g, float* out, size_t n_samples, uint64_t pos_in_frames) {
  if (g->failed) return 0;
  (void)hipSetDevice(g->device);
  size_t total = 0;
  // the staging bus holds max_blocks chunks of max_frames (one super-block); larger writes are split like the reference's mix_buffer loop
  const size_t cap = (size_t)g->stride * g->max_blocks;
  size_t off = 0;
  uint64_t pos = pos_in_frames;
  while (off < n_samples) {
    size_t n = std::min(cap, n_samples - off);
    size_t w = graph_write_impl(g, g->d_bus, n, pos, g->stream);
    if (w == 0) { if (g->failed) return 0; if (off == 0) return 0; memset(out + off, 0, n * sizeof(float)); off += n; pos += n / 2; continue; }
    // device feedback: how many main-mixer sources are still alive (transient sources are dropped when exhausted, :715)
    int n_main = (int)g->mixers[0].voices.size();  // the main-mixer voices are the last n_main entries of the voice index table
    hipLaunchKernelGGL(pg_status_kernel, dim3(1), dim3(64), 0, g->stream, g->d_voices.d, g->d_voice_index.d + (g->d_voice_index.n - (size_t)n_main), n_main,
                       g->d_bus + cap);
    if (hipMemcpyAsync(g->h_pinned, g->d_bus, n * sizeof(float), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        hipMemcpyAsync(g->h_pinned + cap, g->d_bus + cap, 4 * sizeof(float), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        pg_stream_sync(g->stream) != hipSuccess) {
      g->failed = true;
      set_error(PG_ERR_DEVICE, "device failure in write: %s", hipGetErrorString(hipGetLastError()));
      return 0;
    }
    g->cmds_since_sync = 0;
    memcpy(out + off, g->h_pinned, n * sizeof(float));
    g->main_active_voices = ((int*)(g->h_pinned + cap))[0];
    off += n; pos += n / 2; total += n;
  }
  return total;
}

// `bus_audible`: device flag "the summed input is audible" (audible_input of process_effects, mixed.rs:627-655); nullptr = audible
static int process_bus_impl(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, hipStream_t s, int* bus_audible) {
  if (g->failed) return PG_ERR_DEVICE;
  (void)hipSetDevice(g->device);
  if (g->mixers[0].fx.empty()) return PG_OK;
  if (g->last_stream && g->last_stream != s) { HIP_TRY(pg_stream_sync(g->last_stream)); g->cmds_since_sync = 0; }
  g->last_stream = s;
  if (g->topo_dirty && rebuild_topology(g, s)) return graph_fail(g, PG_ERR_DEVICE);
  size_t frames = n_samples / 2, done = 0;
  HostMixer& main = g->mixers[0];
  while (done < frames) {
    const uint64_t now = pos_in_frames + done;
    // effect events of the main mixer (queued by write in defer_bus mode) split the bus block at their sample times, exactly
    // like the event loop of MixedSource::write (mixed.rs:679-712)
    std::vector<PgCmd> cmds;
    while (!main.bus_events.empty() && main.bus_events.front().sample_time <= now) {
      PgCmd c = main.bus_events.front().cmd;
      c.frame = 0; c.unit = main.unit_slot;
      cmds.push_back(c);
      main.bus_events.erase(main.bus_events.begin());
    }
    uint64_t n64 = std::min<uint64_t>(frames - done, g->max_frames);
    if (!main.bus_events.empty()) n64 = std::min<uint64_t>(n64, main.bus_events.front().sample_time - now);
    if (n64 == 0) continue;
    const uint32_t n = (uint32_t)n64;
    const PgCmd* d_cmds = nullptr;
    if (!cmds.empty()) { int rc = stage_commands(g, cmds, s, &d_cmds); if (rc) return rc; }
    PgLaunch B;
    memset(&B, 0, sizeof B);
    B.units = g->d_units.d; B.voices = g->d_voices.d; B.fx = g->d_fx.d;
    B.voice_index = g->d_voice_index.d; B.fx_index = g->d_fx_index.d;
    B.cmds = d_cmds; B.n_cmds = (int)cmds.size(); B.error_word = g->d_error;
    B.n_frames = n; B.pos = now; B.sample_rate = g->sample_rate; B.fast = g->fast;
    B.n_units = 1; B.unit_base = main.unit_slot;
    B.bus = d_bus + done * 2; B.bus_audible = bus_audible;
    HIP_TRY(pg_launch_units(B, s));
    done += n;
  }
  return PG_OK;
}
int pg_graph_process_bus_device(pg_graph* g, float* d_bus, size_t n_samples, uint64_t pos_in_frames, void* hip_stream) {
  return process_bus_impl(g, d_bus, n_samples, pos_in_frames, hip_stream ? (hipStream_t)hip_stream : g->stream, nullptr);
}

// ---- voice-sharded graph: one object, n per-device graphs (SURVEY §8b `n_gpus`, §8e) ----------------------------------------------
// The reference's only parallel axis is independent sub-mixers handed to worker threads, each rendering into a private buffer that the
// caller sums (SubMixerThreadPool, src/source/mixed/submixer/thread_pool.rs:92-121,350-412; src/source/mixed.rs:522-536). Here the
// workers are GPUs: every sub-mixer (with everything under it) and every main-mixer source lives on ONE shard — the least loaded one
// when it is added, the greedy placement of WorkerTaskBatcher — state never migrates, each shard renders a partial master bus on its
// own device and stream, the partials travel to the root device (peer copies over xGMI) and are summed there in shard order, and the
// main mixer's effect chain runs once, on the root, behind the sum. One process, one caller thread; the measured multi-GPU path of
// bench.py (one process per GPU, RCCL reduce) shares everything below the ABI with this one.
__global__ void pg_shard_sum_kernel(float* __restrict__ bus, const float* __restrict__ own, const float* __restrict__ gathered, int n_peers, size_t peer_stride, int n,
                                    const int* __restrict__ flags, int n_flags, int* __restrict__ audible_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && audible_out) { int a = 0; for (int k = 0; k < n_flags; ++k) a |= flags[k]; *audible_out = a; }
  if (i >= n) return;
  float acc = own[i];
  for (int p = 0; p < n_peers; ++p) acc = acc + gathered[(size_t)p * peer_stride + i];  // shard order: deterministic
  bus[i] = acc;
}

struct pg_sharded_graph {
  std::vector<pg_graph*> shards;
  std::vector<int> load;                 // sub-mixers + main-mixer sources placed on each shard
  uint32_t sample_rate = 48000;
  size_t max_frames = 0, max_blocks = 1, stride = 0;
  std::vector<float*> d_partial;         // per shard, on its device: [max_blocks * stride] partial master bus
  float* d_gather = nullptr;             // root device: the peers' partials [(n - 1)][max_blocks * stride]
  int* d_flags = nullptr;                // root device: the shards' "audible" flags [n]
  float* d_bus = nullptr;                // root device: the summed bus of a write with a host buffer
  float* h_pinned = nullptr;
  std::vector<hipEvent_t> done;          // per shard: partial (and flag) arrived on the root
  bool failed = false;
  // global id -> shard << 24 | local id; append-only, readable from any thread (control calls)
  pgc::ChunkTable<int32_t> mixer_map, fx_map, voice_map;
};
static inline int shard_of(int32_t packed) { return (int)((uint32_t)packed >> 24); }
static inline int local_of(int32_t packed) { return (int)((uint32_t)packed & 0xffffffu); }

static int sharded_alloc_buffers(pg_sharded_graph* s) {
  const size_t words = s->stride * s->max_blocks + 4;
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
  if (pg_malloc((void**)&s->d_flags, 64 * sizeof(int)) != hipSuccess || sharded_alloc_buffers(s.get())) { set_error(PG_ERR_DEVICE, "device allocation failed"); return nullptr; }
  (void)pg_memset(s->d_flags, 0, 64 * sizeof(int));
  s->mixer_map.append(0);  // global mixer 0 = the main mixer (its chain lives on the root shard)
  return s.release();
}
void pg_sharded_destroy(pg_sharded_graph* s) {
  if (!s) return;
  for (size_t i = 0; i < s->shards.size(); ++i) {
    (void)hipSetDevice(s->shards[i]->device);
    (void)pg_stream_sync(s->shards[i]->stream);
    if (s->d_partial[i]) (void)pg_free(s->d_partial[i]);
    (void)hipEventDestroy(s->done[i]);
  }
  (void)hipSetDevice(s->shards[0]->device);
  if (s->d_gather) (void)pg_free(s->d_gather);
  if (s->d_bus) (void)pg_free(s->d_bus);
  if (s->d_flags) (void)pg_free(s->d_flags);
  if (s->h_pinned) (void)pg_host_free(s->h_pinned);
  for (pg_graph* g : s->shards) pg_graph_destroy(g);
  delete s;
}
int pg_sharded_shard_count(pg_sharded_graph* s) { return (int)s->shards.size(); }
int pg_sharded_set_max_blocks_per_launch(pg_sharded_graph* s, int n_blocks) {
  for (pg_graph* g : s->shards) { int rc = pg_graph_set_max_blocks_per_launch(g, n_blocks); if (rc) return rc; }
  s->max_blocks = (size_t)n_blocks;
  return sharded_alloc_buffers(s);
}
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
int pg_sharded_shard_of_mixer(pg_sharded_graph* s, int mixer_id) {
  int32_t pk;
  if (!sharded_mixer(s, mixer_id, pk)) return -PG_ERR_NOT_FOUND;
  return mixer_id == 0 ? 0 : shard_of(pk);
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
int pg_sharded_stop_voice(pg_sharded_graph* s, int voice_id, uint64_t sample_time) {
  SHARDED_VOICE(s, voice_id, pk);
  return pg_graph_stop_voice(s->shards[shard_of(pk)], local_of(pk), sample_time);
}
int pg_sharded_stop_all_voices(pg_sharded_graph* s) {
  for (pg_graph* g : s->shards) { int rc = pg_graph_stop_all_voices(g); if (rc) return rc; }
  return PG_OK;
}

// Source::write of the sharded main mixer: n asynchronous renders, partials -> root, sum in shard order, bus chain, result in d_out
// (root device) on the root shard's stream. Returns the samples written, 0 when every shard is empty and the main mixer has no chain.
static size_t sharded_write_impl(pg_sharded_graph* s, float* d_out, size_t n_samples, uint64_t pos) {
  if (s->failed) return 0;
  const size_t cap = s->stride * s->max_blocks;
  if (n_samples > cap || n_samples % 2 != 0) { set_error(PG_ERR_PARAMETER, "a sharded write holds at most max_blocks x max_frames stereo frames"); return 0; }
  const size_t n = s->shards.size();
  pg_graph* root = s->shards[0];
  bool any = false;
  for (size_t i = 0; i < n; ++i) {
    pg_graph* g = s->shards[i];
    (void)hipSetDevice(g->device);
    const size_t w = graph_write_impl(g, s->d_partial[i], n_samples, pos, g->stream);
    if (g->failed) { s->failed = true; return 0; }
    if (w == 0) { if (hipMemsetAsync(s->d_partial[i], 0, n_samples * sizeof(float), g->stream) != hipSuccess) { s->failed = true; return 0; } }
    else any = true;
    if (i > 0) {  // partial bus and the shard's `audible` flag -> the root device; the root's stream waits for their arrival only
      if (hipMemcpyPeerAsync(s->d_gather + (i - 1) * (cap + 4), root->device, s->d_partial[i], g->device, n_samples * sizeof(float), g->stream) != hipSuccess ||
          hipMemcpyPeerAsync(s->d_flags + i, root->device, g->d_audible, g->device, sizeof(int), g->stream) != hipSuccess ||
          hipEventRecord(s->done[i], g->stream) != hipSuccess) { s->failed = true; return 0; }
    }
  }
  if (!any && root->mixers[0].fx.empty()) return 0;
  (void)hipSetDevice(root->device);
  for (size_t i = 1; i < n; ++i) if (hipStreamWaitEvent(root->stream, s->done[i], 0) != hipSuccess) { s->failed = true; return 0; }
  if (hipMemcpyAsync(s->d_flags, root->d_audible, sizeof(int), hipMemcpyDeviceToDevice, root->stream) != hipSuccess) { s->failed = true; return 0; }
  hipLaunchKernelGGL(pg_shard_sum_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, root->stream, d_out, s->d_partial[0], s->d_gather, (int)n - 1, cap + 4,
                     (int)n_samples, s->d_flags, (int)n, root->d_audible);
  if (hipGetLastError() != hipSuccess) { s->failed = true; return 0; }
  if (process_bus_impl(root, d_out, n_samples, pos, root->stream, root->d_audible)) { s->failed = true; return 0; }
  return n_samples;
}
size_t pg_sharded_write_device(pg_sharded_graph* s, float* d_out, size_t n_samples, uint64_t pos_in_frames) { return sharded_write_impl(s, d_out, n_samples, pos_in_frames); }
int pg_sharded_synchronize(pg_sharded_graph* s) {
  for (pg_graph* g : s->shards) { (void)hipSetDevice(g->device); HIP_TRY(pg_stream_sync(g->stream)); g->cmds_since_sync = 0; }
  return PG_OK;
}
size_t pg_sharded_write(pg_sharded_graph* s, float* out, size_t n_samples, uint64_t pos_in_frames) {
  const size_t cap = s->stride * s->max_blocks;
  size_t off = 0, total = 0;
  uint64_t pos = pos_in_frames;
  while (off < n_samples) {
    const size_t n = std::min(cap, n_samples - off);
    const size_t w = sharded_write_impl(s, s->d_bus, n, pos);
    if (w == 0) { if (s->failed || off == 0) return 0; memset(out + off, 0, n * sizeof(float)); off += n; pos += n / 2; continue; }
    pg_graph* root = s->shards[0];
    (void)hipSetDevice(root->device);
    if (hipMemcpyAsync(s->h_pinned, s->d_bus, n * sizeof(float), hipMemcpyDeviceToHost, root->stream) != hipSuccess || pg_sharded_synchronize(s) != PG_OK) { s->failed = true; return 0; }
    memcpy(out + off, s->h_pinned, n * sizeof(float));
    off += n; pos += n / 2; total += n;
  }
  return total;
}
int pg_sharded_device_errors(pg_sharded_graph* s) {
  int e = 0;
  for (pg_graph* g : s->shards) { const int r = pg_graph_device_errors(g); if (r < 0) return r; e |= r; }
  return e;
}

// ---- standalone effect: a one-unit graph whose unit is UNIT_EFFECT -------------------------------------------
struct pg_effect {
  int kind = 0, device = 0;
  HostFx host;
  bool initialized = false;
  uint32_t sample_rate = 0;
  size_t max_frames = 0;
  hipStream_t stream = nullptr;
  PgUnit* d_unit = nullptr;
  PgFx* d_fx = nullptr;
  int32_t* d_fx_index = nullptr;
  PgCmd* d_cmds = nullptr;
  float* d_buf = nullptr;
  int32_t* d_idx_log = nullptr;  // test hook (pg_effect_debug_index_log)
  size_t idx_log_words = 0;
  std::vector<PgCmd> pending;
  size_t cmd_cap = 64;           // commands d_cmds holds
};

pg_effect* pg_effect_create(int kind, const pg_effect_init* init, int device) {
  std::unique_ptr<pg_effect> e(new pg_effect());
  e->kind = kind; e->device = device;
  if (host_fx_from_init(kind, init, e->host)) return nullptr;
  return e.release();
}
void pg_effect_destroy(pg_effect* e) {
  if (!e) return;
  if (e->initialized) {
    (void)hipSetDevice(e->device);
    (void)pg_stream_sync(e->stream);
    (void)pg_free(e->d_unit); (void)pg_free(e->d_fx); (void)pg_free(e->d_fx_index); (void)pg_free(e->d_cmds); (void)pg_free(e->d_buf);
    if (e->host.d_mem) (void)pg_free(e->host.d_mem);
    if (e->d_idx_log) (void)pg_free(e->d_idx_log);
    (void)hipStreamDestroy(e->stream);
  }
  delete e;
}
int pg_effect_initialize(pg_effect* e, uint32_t sample_rate, size_t channel_count, size_t max_frames) {
  if (e->initialized) return set_error(PG_ERR_STATE, "effect is already initialized");
  if (channel_count != 2) return set_error(PG_ERR_PARAMETER, "%sEffect only supports stereo I/O", KINDS[e->kind].name);
  if (sample_rate == 0 || max_frames == 0 || max_frames > PG_MAX_FRAMES) return set_error(PG_ERR_PARAMETER, "max_frames must be in 1..=%d", PG_MAX_FRAMES);
  HIP_TRY(hipSetDevice(e->device));
  PgFx fx;
  int rc = build_fx_device_state(e->host, sample_rate, e->device, true, fx);
  if (rc) return rc;
  HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  PgUnit u;
  memset(&u, 0, sizeof u);
  u.kind = UNIT_EFFECT; u.n_fx = 1; u.fx_off = 0; u.effects_bypassed = 0;
  int32_t zero = 0;
  HIP_TRY(pg_malloc((void**)&e->d_unit, sizeof u));
  HIP_TRY(pg_malloc((void**)&e->d_fx, sizeof fx));
  HIP_TRY(pg_malloc((void**)&e->d_fx_index, 4));
  HIP_TRY(pg_malloc((void**)&e->d_cmds, sizeof(PgCmd) * 64));
  e->cmd_cap = 64;
  HIP_TRY(pg_malloc((void**)&e->d_buf, max_frames * 2 * sizeof(float)));
  HIP_TRY(pg_memcpy(e->d_unit, &u, sizeof u, hipMemcpyHostToDevice));
  HIP_TRY(pg_memcpy(e->d_fx, &fx, sizeof fx, hipMemcpyHostToDevice));
  HIP_TRY(pg_memcpy(e->d_fx_index, &zero, 4, hipMemcpyHostToDevice));
  e->sample_rate = sample_rate; e->max_frames = max_frames; e->initialized = true;
  return PG_OK;
}
// Test hook: collect the floor()-derived read indices of the effect's delay lines during the following process calls (time-parallel
// paths of Reverb: ((frame * 8 + line) * 2 + channel) -> read_1 of ReverbDelayLine::get; Delay / Chorus: (frame * 2 + channel) ->
// read_idx1 of InterpolatedDelayLine::process), `words` slots, -1 = not written. out == nullptr: (re)arm the log; else copy it out.
int pg_effect_debug_index_log(pg_effect* e, int32_t* out, size_t words) {
  if (!e->initialized) return set_error(PG_ERR_STATE, "effect is not initialized");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(pg_stream_sync(e->stream));
  if (!out) {
    if (words > e->idx_log_words) {
      if (e->d_idx_log) (void)pg_free(e->d_idx_log);
      e->d_idx_log = nullptr;
      HIP_TRY(pg_malloc((void**)&e->d_idx_log, words * sizeof(int32_t)));
      e->idx_log_words = words;
    }
    if (e->d_idx_log) HIP_TRY(pg_memset(e->d_idx_log, 0xff, e->idx_log_words * sizeof(int32_t)));
    return PG_OK;
  }
  if (!e->d_idx_log || words > e->idx_log_words) return set_error(PG_ERR_PARAMETER, "index log is not armed for %zu words", words);
  HIP_TRY(pg_memcpy(out, e->d_idx_log, words * sizeof(int32_t), hipMemcpyDeviceToHost));
  return PG_OK;
}
int pg_effect_process_started(pg_effect*) { return PG_OK; }  // no-ops for all stock effects (src/effect.rs:127-139)
int pg_effect_process_stopped(pg_effect*) { return PG_OK; }

// room for one more queued command (a launch applies commands at the head of the frames it renders, so a launch of no frames cannot flush them:
// the queue grows instead)
static int effect_reserve_cmd(pg_effect* e) {
  if (e->pending.size() < e->cmd_cap) return PG_OK;
  HIP_TRY(hipSetDevice(e->device));
  PgCmd* bigger = nullptr;
  HIP_TRY(pg_malloc((void**)&bigger, sizeof(PgCmd) * e->cmd_cap * 2));
  HIP_TRY(pg_stream_sync(e->stream));
  (void)pg_free(e->d_cmds);
  e->d_cmds = bigger;
  e->cmd_cap *= 2;
  return PG_OK;
}
static int effect_run(pg_effect* e, float* host_buf, size_t n_samples, uint64_t pos) {
  HIP_TRY(hipSetDevice(e->device));
  if (n_samples) HIP_TRY(hipMemcpyAsync(e->d_buf, host_buf, n_samples * sizeof(float), hipMemcpyHostToDevice, e->stream));
  if (!e->pending.empty()) HIP_TRY(hipMemcpyAsync(e->d_cmds, e->pending.data(), e->pending.size() * sizeof(PgCmd), hipMemcpyHostToDevice, e->stream));
  PgLaunch L;
  memset(&L, 0, sizeof L);
  L.units = e->d_unit; L.fx = e->d_fx; L.fx_index = e->d_fx_index;
  L.cmds = e->d_cmds; L.n_cmds = (int)e->pending.size();
  L.n_units = 1; L.unit_base = 0; L.n_frames = (uint32_t)(n_samples / 2); L.pos = pos; L.sample_rate = e->sample_rate; L.fast = 1;
  L.bus = e->d_buf;
  L.index_log = e->d_idx_log;
  HIP_TRY(pg_launch_units(L, e->stream));
  if (n_samples) HIP_TRY(hipMemcpyAsync(host_buf, e->d_buf, n_samples * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(pg_stream_sync(e->stream));
  e->pending.clear();
  return PG_OK;
}
int pg_effect_process(pg_effect* e, float* interleaved, size_t n_samples, uint64_t pos_in_frames) {
  if (!e->initialized) return set_error(PG_ERR_STATE, "effect is not initialized");
  if (n_samples % 2 != 0 || n_samples / 2 > e->max_frames) return set_error(PG_ERR_PARAMETER, "buffer must hold <= max_frames stereo frames");
  if (n_samples == 0) return PG_OK;  // nothing to render: parameter updates and messages received so far stay queued, in order, for the next call that does
  return effect_run(e, interleaved, n_samples, pos_in_frames);
}
int pg_effect_set_parameter(pg_effect* e, uint32_t fourcc, float value, int is_normalized) {
  int pi = find_param(e->kind, fourcc);
  if (pi < 0) return set_error(PG_ERR_PARAMETER, "Unknown parameter: 0x%08x for effect '%s'", fourcc, KINDS[e->kind].name);
  float raw;
  if (!resolve_update(KINDS[e->kind].params[pi], value, is_normalized != 0, raw)) return PG_OK;
  if (e->kind == PG_FX_DELAY && pi == P_DELAY_LFO_SHAPE && (int)raw >= 5)
    return set_error(PG_ERR_PARAMETER, "LFO shapes Random/Smooth Random are not supported (OS-seeded RNG in the reference)");
  e->host.target[pi] = raw;
  if (!e->initialized) { e->host.init_raw[pi] = raw; return PG_OK; }  // before initialize: plain value update
  PgCmd c;
  memset(&c, 0, sizeof c);
  c.type = CMD_FX_PARAM; c.unit = 0; c.target = 0; c.param = pi; c.value = raw; c.frame = 0; c.value64 = fx_param_aux(e->kind, pi, raw, e->sample_rate);
  int rc = effect_reserve_cmd(e);
  if (rc) return rc;
  e->pending.push_back(c);
  return PG_OK;
}
int pg_effect_message_reset(pg_effect* e) {
  if (e->kind != PG_FX_DELAY && e->kind != PG_FX_REVERB && e->kind != PG_FX_CHORUS)
    return set_error(PG_ERR_PARAMETER, "%sEffect: Invalid/unknown message payload", KINDS[e->kind].name);
  if (!e->initialized) return PG_OK;
  PgCmd c;
  memset(&c, 0, sizeof c);
  c.type = CMD_FX_RESET; c.unit = 0; c.target = 0;
  int rc = effect_reserve_cmd(e);
  if (rc) return rc;
  e->pending.push_back(c);
  return PG_OK;
}
int64_t pg_effect_tail(pg_effect* e) {  // Effect::process_tail from the target values (host shadow)
  const std::vector<float>& t = e->host.target;
  double sr = (double)e->sample_rate;
  switch (e->kind) {
    case PG_FX_GAIN: { int m = (int)t[1]; return m == 0 ? 0 : (int64_t)((uint64_t)e->sample_rate / (uint64_t)(m == 1 ? 1 : (m == 2 ? 5 : 20))); }
    case PG_FX_PANNING: return 0;
    case PG_FX_FILTER: return e->sample_rate / 10;
    case PG_FX_EQ5: return e->sample_rate / 5;
    case PG_FX_DELAY: {
      if (t[P_DELAY_DRIVE] > 0.0f) return -1;
      double delay_ms = (double)(t[P_DELAY_TIME] + 50.0f);
      double fb = (double)std::fabs(t[P_DELAY_FEEDBACK]);
      if (fb >= 0.9999) return INT64_MAX;
      if (fb < 0.001) return (int64_t)d2u64(std::ceil(delay_ms * sr / 1000.0));
      double ds = delay_ms * sr / 1000.0;
      return (int64_t)std::max<uint64_t>(d2u64(std::ceil(ds + ds * std::log10(0.001) / std::log10(fb))), 1);
    }
    case PG_FX_REVERB: {
      double rs = (double)t[0];
      double size = (rs * rs * 75.0) + 25.0;
      uint64_t max_delay = d2u64(79.0 * size);
      double tt = 1.0 - (0.82 - (((1.0 - rs) * 0.7) + (size * 0.002)));
      double fb = 1.0 - (tt * tt) * (tt * tt);
      if (fb >= 1.0) return INT64_MAX;
      if (fb == 0.0) return (int64_t)max_delay;
      return (int64_t)(max_delay + d2u64((double)max_delay * std::log10(0.001) / std::log10(fb)));
    }
    case PG_FX_CHORUS: {
      float srf = (float)e->sample_rate;
      float total_ms = t[P_CHORUS_DELAY] + 256.0f * 1000.0f / srf;
      float fb = std::fabs(t[P_CHORUS_FEEDBACK]);
      if (fb >= 1.0f) return INT64_MAX;
      if (fb < 0.001f) return (int64_t)f2u64(std::ceil(total_ms * srf / 1000.0f));
      float total = total_ms * srf / 1000.0f;
      float decay = total + (float)((double)total * std::log10(0.001) / std::log10((double)fb));
      return (int64_t)f2u64(std::ceil(decay));
    }
    case PG_FX_COMPRESSOR: return (int64_t)(f2u64(std::ceil(t[P_COMP_LOOKAHEAD] * (float)e->sample_rate)) + f2u64(std::ceil(t[P_COMP_RELEASE] * (float)e->sample_rate)));
    case PG_FX_GATE: return (int64_t)(f2u64(std::ceil(t[P_GATE_HOLD] * (float)e->sample_rate)) + f2u64(std::ceil(t[P_GATE_RELEASE] * (float)e->sample_rate)));
    default: return 0;
  }
}

}  // extern "C"
