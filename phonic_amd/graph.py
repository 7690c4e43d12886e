"""GPU-resident mixer graph: Python mirror of the reference's `Player` calls that populate the main
`MixedSource` (src/player.rs:519-602,773-822,893-939) and of `Source::write` (src/source.rs:95).
Everything forwards to the C ABI of include/phonic_gpu.h; there is no CPU fallback."""
import ctypes as C

import numpy as np

from . import _capi
from ._wrap import GraphHandle, PhonicError


class Graph(GraphHandle):
    def __init__(self, sample_rate=48000, channels=2, max_frames=4096, device=0):
        super().__init__(_capi.load(), "pg_", sample_rate, channels, max_frames, device)
        self.max_frames = max_frames

    # -- device-side output (multi-GPU master-bus reduce, bench.py) ------------------------------------
    def write_device(self, d_out_ptr, n_samples, pos_in_frames, stream=None):
        """`Source::write` into device memory (`d_out_ptr` = device pointer as int). Asynchronous on `stream` (a hipStream_t as int);
        without one — and that includes the default stream, whose handle is 0 — the call renders on the graph's own stream and
        returns when the block is complete."""
        return self._lib.pg_graph_write_device(self._h, C.c_void_p(d_out_ptr), n_samples, pos_in_frames, C.c_void_p(stream or 0))

    # -- host-fed sources (any `dyn Source` the host pulls itself) -------------------------------------
    def add_stream_voice(self, mixer_id, channels, rate, capacity_frames, **opts):
        o = _capi.default_voice_options(**opts)
        v = self._id(self._lib.pg_graph_add_stream_voice(self._h, mixer_id, channels, rate, capacity_frames, C.byref(o)))
        if not hasattr(self, "_stream_channels"):
            self._stream_channels = {}
        self._stream_channels[v] = channels
        return v

    def feed_voice(self, voice, frames):
        """Interleaved float32 frames of the host's source; raises SendError (PG_ERR_QUEUE_FULL) when the ring has no room for all of them."""
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        ch = self._stream_channels.get(voice) if hasattr(self, "_stream_channels") else None
        n = frames.size // (ch or 1)
        self._check(self._lib.pg_graph_feed_voice(self._h, voice, frames.ctypes.data_as(C.POINTER(C.c_float)), n))

    def end_stream_voice(self, voice):
        self._check(self._lib.pg_graph_end_stream_voice(self._h, voice))

    def stream_voice_consumed(self, voice):
        return self._id(self._lib.pg_graph_stream_voice_consumed(self._h, voice))

    def set_max_blocks_per_launch(self, n_blocks):
        """Offline rendering: let one write call render up to `n_blocks` blocks of max_frames per launch sequence (steady state only)."""
        self._check(self._lib.pg_graph_set_max_blocks_per_launch(self._h, int(n_blocks)))

    def device_errors(self):
        """Sticky consistency flags raised by the kernels (0 = none)."""
        return self._id(self._lib.pg_graph_device_errors(self._h))

    def set_defer_bus(self, defer):
        self._check(self._lib.pg_graph_set_defer_bus(self._h, 1 if defer else 0))

    def process_bus_device(self, d_bus_ptr, n_samples, pos_in_frames, stream=None, flags_ptr=None, n_words=0):
        """The main mixer's chain over a summed bus the caller holds; flags_ptr: the ranks' summed `audible` words (export_audible), one per block of max_frames."""
        if flags_ptr:
            self._check(self._lib.pg_graph_process_bus_device_flags(self._h, C.c_void_p(d_bus_ptr), n_samples, pos_in_frames, C.c_void_p(stream or 0), C.c_void_p(flags_ptr), n_words))
        else:
            self._check(self._lib.pg_graph_process_bus_device(self._h, C.c_void_p(d_bus_ptr), n_samples, pos_in_frames, C.c_void_p(stream or 0)))

    def export_audible(self, d_dst_ptr, n_words, stream=None):
        """The `audible` words of the last write_device call of a deferred-bus graph as floats (0 / 1): they ride with the partial bus in one reduce."""
        self._check(self._lib.pg_graph_export_audible(self._h, C.c_void_p(d_dst_ptr), n_words, C.c_void_p(stream or 0)))

    def audible_words(self):
        """Words the last deferred-bus write_device call left: one per piece it was rendered in (events add pieces)."""
        return int(self._lib.pg_graph_audible_words(self._h))

    def next_main_event(self, pos_in_frames):
        """Sample time of the first main-mixer event behind pos_in_frames (None: none pending). Writing thread only (drains the control ring)."""
        t = int(self._lib.pg_graph_next_main_event(self._h, int(pos_in_frames)))
        return None if t == 0xFFFFFFFFFFFFFFFF else t

    def synchronize(self):
        self._check(self._lib.pg_graph_synchronize(self._h))

    def voice_count(self):
        return self._lib.pg_graph_voice_count(self._h)

    def deferred_units(self):
        """Units the time-parallel kernels handed to the exact serial kernel in the last block (0 in steady state)."""
        return self._id(self._lib.pg_graph_deferred_units(self._h))

    def is_voice_playing(self, voice):
        return bool(self._lib.pg_graph_is_voice_playing(self._h, voice))

    def kernel_ms(self, reset=True):
        """(average ms of the unit kernel per launch, launches) since the last reset, from hipEvents on the graph's stream."""
        n = C.c_uint64(0)
        ms = self._lib.pg_graph_kernel_ms(self._h, 1 if reset else 0, C.byref(n))
        return ms, n.value

    def kernel_stats(self, reset=True):
        """(average ms per timed launch of the dominant kernel, timed launches, blocks those launches rendered): a super-block launch
        renders several max_frames blocks per unit."""
        ms, n, b = C.c_double(0.0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.pg_graph_kernel_stats(self._h, 1 if reset else 0, C.byref(ms), C.byref(n), C.byref(b)))
        return (ms.value / n.value if n.value else 0.0), n.value, b.value

    def bus_kernel_stats(self, reset=True):
        """The same for the launches of the main mixer's effect chain: (average ms per timed bus launch, timed launches, blocks they walked)."""
        ms, n, b = C.c_double(0.0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.pg_graph_bus_kernel_stats(self._h, 1 if reset else 0, C.byref(ms), C.byref(n), C.byref(b)))
        return (ms.value / n.value if n.value else 0.0), n.value, b.value

    def dynamic_stats(self, reset=True):
        """{unit_blocks, deferred_unit_blocks, generic_launches, generic_launches_with_work, generic_ms, generic_timed}: what leaving the
        steady state cost since the last reset (pg_graph_dynamic_stats; waits for the stream)."""
        out = (C.c_uint64 * 4)()
        ms, n = C.c_double(0.0), C.c_uint64(0)
        self._check(self._lib.pg_graph_dynamic_stats(self._h, 1 if reset else 0, out, C.byref(ms), C.byref(n)))
        return {"unit_blocks": int(out[0]), "deferred_unit_blocks": int(out[1]), "generic_launches": int(out[2]), "generic_launches_with_work": int(out[3]),
                "generic_ms": ms.value, "generic_timed": int(n.value)}

    def bus_kernel(self):
        return self._lib.pg_graph_bus_kernel(self._h).decode()

    def set_timing_period(self, every_n_rounds):
        """Time every n-th round with a hipEvent pair (the pair costs ~8 us of stream time); 0 = never."""
        self._check(self._lib.pg_graph_set_timing_period(self._h, int(every_n_rounds)))

    def dominant_kernel(self):
        """Name(s) of the kernel launch(es) that `kernel_ms` brackets for this graph."""
        return self._lib.pg_graph_dominant_kernel(self._h).decode()

    def set_fast_math(self, level):
        self._check(self._lib.pg_graph_set_fast_math(self._h, int(level)))

    def set_staged(self, mode):
        """[Gain|Panning]* -> Reverb sub-mixers: 1/True = staged single launch (default), 2 = one launch per stage, 0/False = fused fast kernel."""
        self._check(self._lib.pg_graph_set_staged(self._h, int(mode)))


class ShardedGraph:
    """The main mixer spread over several devices behind ONE handle (pg_sharded_*, include/phonic_gpu.h): same calls as `Graph` for
    building, automation and `write`; sub-mixers and main-mixer sources are placed on the least loaded shard, the main mixer's effects
    run on the root behind the sum of the shards' partial buses."""

    def __init__(self, devices, sample_rate=48000, channels=2, max_frames=4096):
        self._lib = _capi.load()
        self.sample_rate, self.channels, self.max_frames = sample_rate, channels, max_frames
        arr = (C.c_int * len(devices))(*devices)
        self._h = self._lib.pg_sharded_create(sample_rate, channels, max_frames, arr, len(devices))
        if not self._h:
            raise PhonicError(_capi.PG_ERR_DEVICE, (self._lib.pg_last_error_message() or b"").decode())

    def _check(self, code):
        if code != 0:
            raise PhonicError(code, (self._lib.pg_last_error_message() or b"").decode())

    def _id(self, v):
        if v < 0:
            raise PhonicError(-v, (self._lib.pg_last_error_message() or b"").decode())
        return v

    def shard_count(self):
        return self._lib.pg_sharded_shard_count(self._h)

    def set_max_blocks_per_launch(self, n_blocks):
        self._check(self._lib.pg_sharded_set_max_blocks_per_launch(self._h, int(n_blocks)))

    def add_mixer(self, parent=None):
        return self._id(self._lib.pg_sharded_add_mixer_to(self._h, parent or 0))

    def add_effect(self, mixer_id, kind, params=None, reverb_seeds=None, lfo_seed=None):
        init = _capi.make_init(params, reverb_seeds, lfo_seed)
        return self._id(self._lib.pg_sharded_add_effect(self._h, mixer_id, kind, C.byref(init)))

    def add_voice(self, mixer_id, pcm, src_channels, src_rate, **opts):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        o = _capi.default_voice_options(**opts)
        return self._id(self._lib.pg_sharded_add_voice(self._h, mixer_id, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size // src_channels, src_channels, src_rate, C.byref(o)))

    def add_stream_voice(self, mixer_id, channels, rate, capacity_frames, **opts):
        o = _capi.default_voice_options(**opts)
        v = self._id(self._lib.pg_sharded_add_stream_voice(self._h, mixer_id, channels, rate, capacity_frames, C.byref(o)))
        if not hasattr(self, "_stream_channels"):
            self._stream_channels = {}
        self._stream_channels[v] = channels
        return v

    def feed_voice(self, voice, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        self._check(self._lib.pg_sharded_feed_voice(self._h, voice, frames.ctypes.data_as(C.POINTER(C.c_float)), frames.size // self._stream_channels[voice]))

    def end_stream_voice(self, voice):
        self._check(self._lib.pg_sharded_end_stream_voice(self._h, voice))

    def stream_voice_consumed(self, voice):
        return self._id(self._lib.pg_sharded_stream_voice_consumed(self._h, voice))

    def shard_of_mixer(self, mixer_id):
        return self._id(self._lib.pg_sharded_shard_of_mixer(self._h, mixer_id))

    def schedule_param(self, effect_id, id4, value, sample_time, normalized=False):
        self._check(self._lib.pg_sharded_schedule_param(self._h, effect_id, _capi.fourcc(id4), float(value), 1 if normalized else 0, sample_time))

    def schedule_reset(self, effect_id, sample_time):
        self._check(self._lib.pg_sharded_schedule_reset(self._h, effect_id, sample_time))

    def set_voice_volume(self, voice, volume, sample_time):
        self._check(self._lib.pg_sharded_set_voice_volume(self._h, voice, float(volume), sample_time))

    def set_voice_panning(self, voice, panning, sample_time):
        self._check(self._lib.pg_sharded_set_voice_panning(self._h, voice, float(panning), sample_time))

    def stop_voice(self, voice, sample_time):
        self._check(self._lib.pg_sharded_stop_voice(self._h, voice, sample_time))

    def remove_voice(self, voice):
        self._check(self._lib.pg_sharded_remove_voice(self._h, voice))

    def set_voice_speed(self, voice, speed, sample_time, glide=None):
        self._check(self._lib.pg_sharded_set_voice_speed(self._h, voice, float(speed), float(glide) if glide else 0.0, sample_time))

    def seek_voice(self, voice, seconds, sample_time):
        self._check(self._lib.pg_sharded_seek_voice(self._h, voice, float(seconds), sample_time))

    def remove_mixer(self, mixer_id):
        self._check(self._lib.pg_sharded_remove_mixer(self._h, mixer_id))

    def remove_effect(self, effect_id):
        self._check(self._lib.pg_sharded_remove_effect(self._h, effect_id))

    def move_effect(self, effect_id, mixer_id, movement, offset=0):
        self._check(self._lib.pg_sharded_move_effect(self._h, effect_id, mixer_id, movement, offset))

    def set_reduce(self, mode):
        """REDUCE_PEER_COPY (default) or REDUCE_RCCL (ncclReduce over xGMI; one device per shard). Raises with RCCL's error text on failure."""
        if int(mode) == _capi.REDUCE_RCCL:
            _capi.preload_rccl()
        self._check(self._lib.pg_sharded_set_reduce(self._h, int(mode)))

    def reduce_mode(self):
        return self._lib.pg_sharded_reduce_mode(self._h)

    def is_voice_playing(self, voice):
        return bool(self._lib.pg_sharded_is_voice_playing(self._h, voice))

    def stop_all_voices(self):
        self._check(self._lib.pg_sharded_stop_all_voices(self._h))

    def write(self, out, pos_in_frames):
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        return self._lib.pg_sharded_write(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, pos_in_frames)

    def write_device(self, d_out_ptr, n_samples, pos_in_frames):
        return self._lib.pg_sharded_write_device(self._h, C.c_void_p(d_out_ptr), n_samples, pos_in_frames)

    def synchronize(self):
        self._check(self._lib.pg_sharded_synchronize(self._h))

    def device_errors(self):
        return self._id(self._lib.pg_sharded_device_errors(self._h))

    def render(self, n_blocks, block_frames=1024, start_pos=0):
        out = np.zeros((n_blocks, block_frames * self.channels), dtype=np.float32)
        pos = start_pos
        for b in range(n_blocks):
            self.write(out[b], pos)
            pos += block_frames
        return out.reshape(-1)

    def close(self):
        if self._h:
            self._lib.pg_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hip_calls():
    """Process-wide counters of the library's own HIP calls: dict(alloc, free, sync, blocking_copy) — see pg_debug_hip_calls."""
    out = (C.c_uint64 * 4)()
    _capi.load().pg_debug_hip_calls(out)
    return dict(alloc=out[0], free=out[1], sync=out[2], blocking_copy=out[3])


def effect_parameters(kind):
    """`Effect::parameters()` descriptors of an effect kind."""
    lib = _capi.load()
    out = []
    for i in range(lib.pg_effect_kind_param_count(kind)):
        d = _capi.ParamDesc()
        if lib.pg_effect_kind_param(kind, i, C.byref(d)) != 0:
            raise PhonicError(_capi.PG_ERR_NOT_FOUND, "parameter")
        out.append(dict(fourcc=d.fourcc, type=d.type, min=d.min, max=d.max, default=d.default_value, scaling=d.scaling,
                        scaling_args=(d.scaling_arg0, d.scaling_arg1), n_values=d.n_values, name=d.name.decode()))
    return out
