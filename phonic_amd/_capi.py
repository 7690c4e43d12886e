"""ctypes view of include/phonic_gpu.h: struct layouts, constants and the loader of the HIP library.

The product library (phonic_amd/csrc/libphonic_gpu.so) is built by `__graft_entry__.build()`.
Loading fails loudly when it is missing: there is no CPU fallback for the product path.
"""
import ctypes as C
import os

PG_OK, PG_ERR_PARAMETER, PG_ERR_NOT_FOUND, PG_ERR_QUEUE_FULL, PG_ERR_DEVICE, PG_ERR_STATE = range(6)

FX_GAIN, FX_PANNING, FX_FILTER, FX_EQ5, FX_DELAY, FX_REVERB, FX_CHORUS, FX_COMPRESSOR, FX_GATE, FX_DISTORTION = range(10)
FX_NAMES = ["Gain", "Panning", "Filter", "Eq5", "Delay", "Reverb", "Chorus", "Compressor", "Gate", "Distortion"]

PG_MAX_INIT_PARAMS = 16
MOVE_DIRECTION, MOVE_START, MOVE_END = 0, 1, 2  # EffectMovement (src/player.rs:75-82)
REDUCE_PEER_COPY, REDUCE_RCCL = 0, 1  # pg_sharded_set_reduce
PG_REPEAT_FOREVER = 2**64 - 1
INT64_MAX = 2**63 - 1


def fourcc(s):
    """FourCC(*b"room") -> u32 (big endian, as in include/phonic_gpu.h PG_FOURCC)."""
    b = s.encode() if isinstance(s, str) else bytes(s)
    assert len(b) == 4, s
    return b[0] << 24 | b[1] << 16 | b[2] << 8 | b[3]


class EffectInit(C.Structure):
    _fields_ = [
        ("n_params", C.c_uint32),
        ("fourcc", C.c_uint32 * PG_MAX_INIT_PARAMS),
        ("value", C.c_float * PG_MAX_INIT_PARAMS),
        ("has_reverb_seeds", C.c_uint32),
        ("reverb_fpd_l", C.c_uint32),
        ("reverb_fpd_r", C.c_uint32),
        ("reverb_vib_phase", C.c_double * 16),
        ("has_lfo_seed", C.c_uint32),
        ("reserved_lfo", C.c_uint32),
        ("lfo_rng_state", C.c_uint64 * 4),
    ]


class VoiceOptions(C.Structure):
    _fields_ = [
        ("volume", C.c_float),
        ("panning", C.c_float),
        ("speed", C.c_double),
        ("repeat", C.c_uint64),
        ("has_repeat", C.c_uint32),
        ("has_loop_range", C.c_uint32),
        ("loop_start", C.c_uint64),
        ("loop_end", C.c_uint64),
        ("start_time", C.c_uint64),
        ("fade_in_seconds", C.c_float),
        ("fade_out_seconds", C.c_float),
        ("source_rate", C.c_uint32),
        ("non_transient", C.c_uint32),
    ]


class ParamDesc(C.Structure):
    _fields_ = [
        ("fourcc", C.c_uint32),
        ("type", C.c_int32),
        ("min", C.c_float),
        ("max", C.c_float),
        ("default_value", C.c_float),
        ("scaling", C.c_int32),
        ("scaling_arg0", C.c_float),
        ("scaling_arg1", C.c_float),
        ("n_values", C.c_int32),
        ("name", C.c_char_p),
    ]


def make_init(params=None, reverb_seeds=None, lfo_seed=None):
    """Build a pg_effect_init. params: dict {fourcc-str: raw value}; reverb_seeds: (fpd_l, fpd_r, [16 phases]); lfo_seed: the four u64 of the
    Delay LFO's Xoshiro256++ state (Random / Smooth Random shapes)."""
    init = EffectInit()
    params = params or {}
    assert len(params) <= PG_MAX_INIT_PARAMS
    init.n_params = len(params)
    for i, (k, v) in enumerate(params.items()):
        init.fourcc[i] = fourcc(k)
        init.value[i] = float(v)
    if reverb_seeds is not None:
        init.has_reverb_seeds = 1
        init.reverb_fpd_l, init.reverb_fpd_r = int(reverb_seeds[0]), int(reverb_seeds[1])
        for i in range(16):
            init.reverb_vib_phase[i] = float(reverb_seeds[2][i])
    if lfo_seed is not None:
        init.has_lfo_seed = 1
        for i in range(4):
            init.lfo_rng_state[i] = int(lfo_seed[i]) & (2**64 - 1)
    return init


def default_voice_options(**kw):
    """FilePlaybackOptions::default() (reference src/source/file.rs:94-112)."""
    o = VoiceOptions()
    o.volume, o.panning, o.speed = 1.0, 0.0, 1.0
    o.repeat, o.has_repeat, o.has_loop_range = 0, 0, 0
    o.loop_start = o.loop_end = 0
    o.start_time = 0
    o.fade_in_seconds, o.fade_out_seconds = 0.0, 0.05
    o.source_rate, o.non_transient = 0, 0
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def declare(lib, prefix):
    """Declare argtypes/restypes of the API shared by the product (pg_) and, in tests, the oracle (po_)."""
    P = C.POINTER
    f = lambda name: getattr(lib, prefix + name)
    vp = C.c_void_p
    sigs = {
        "effect_initialize": (C.c_int, [vp, C.c_uint32, C.c_size_t, C.c_size_t]),
        "effect_process": (C.c_int, [vp, P(C.c_float), C.c_size_t, C.c_uint64]),
        "effect_tail": (C.c_int64, [vp]),
        "effect_set_parameter": (C.c_int, [vp, C.c_uint32, C.c_float, C.c_int]),
        "effect_message_reset": (C.c_int, [vp]),
        "effect_destroy": (None, [vp]),
        "graph_create": (vp, [C.c_uint32, C.c_uint32, C.c_size_t, C.c_int]),
        "graph_destroy": (None, [vp]),
        "graph_add_mixer": (C.c_int, [vp]),
        "graph_add_mixer_to": (C.c_int, [vp, C.c_int]),
        "graph_add_effect": (C.c_int, [vp, C.c_int, C.c_int, P(EffectInit)]),
        "graph_add_voice": (C.c_int, [vp, C.c_int, P(C.c_float), C.c_size_t, C.c_uint32, C.c_uint32, P(VoiceOptions)]),
        "graph_schedule_param": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_float, C.c_int, C.c_uint64]),
        "graph_schedule_reset": (C.c_int, [vp, C.c_int, C.c_uint64]),
        "graph_remove_effect": (C.c_int, [vp, C.c_int]),
        "graph_remove_mixer": (C.c_int, [vp, C.c_int]),
        "graph_stop_all_voices": (C.c_int, [vp]),
        "graph_move_effect": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "graph_set_voice_volume": (C.c_int, [vp, C.c_int, C.c_float, C.c_uint64]),
        "graph_set_voice_panning": (C.c_int, [vp, C.c_int, C.c_float, C.c_uint64]),
        "graph_stop_voice": (C.c_int, [vp, C.c_int, C.c_uint64]),
        "graph_remove_voice": (C.c_int, [vp, C.c_int]),
        "graph_set_voice_speed": (C.c_int, [vp, C.c_int, C.c_double, C.c_float, C.c_uint64]),
        "graph_seek_voice": (C.c_int, [vp, C.c_int, C.c_double, C.c_uint64]),
        "graph_write": (C.c_size_t, [vp, P(C.c_float), C.c_size_t, C.c_uint64]),
    }
    for name, (res, args) in sigs.items():
        fn = f(name)
        fn.restype = res
        fn.argtypes = args
    return lib


def _preload_hip_runtime():
    """One HIP runtime per process. PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, the same
    soname libphonic_gpu.so needs); if the system runtime from /opt/rocm were loaded first, a later `import torch`
    would bring in a second runtime that sees no GPUs. So when torch is installed its runtime is loaded first
    (RTLD_GLOBAL) and libphonic_gpu.so binds to it by soname. PHONIC_HIP_RUNTIME=system opts out (torch-free runs,
    e.g. profiling the single-GPU bench against /opt/rocm)."""
    import sys

    if os.environ.get("PHONIC_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def preload_rccl():
    """The RCCL that matches the HIP runtime in the process: libphonic_gpu.so looks RCCL up by soname (librccl.so.1) when
    pg_sharded_set_reduce(PG_REDUCE_RCCL) is called. With PyTorch's runtime loaded (see above) that must be PyTorch's RCCL, so it is
    loaded first; without torch (or with PHONIC_HIP_RUNTIME=system) the library finds ROCm's own on its run path."""
    import sys

    if os.environ.get("PHONIC_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def source_hash():
    """16 hex digits over the sources libphonic_gpu.so is built from (csrc/*.hip|.h|.inl, the Makefile, include/phonic_gpu.h). Profiles that
    bench.py quotes (profiles/*_pmc_traffic.json) carry the hash of the build they were measured with."""
    import glob
    import hashlib

    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) + glob.glob(os.path.join(here, "csrc", "*.inl")))
    files += [os.path.join(here, "csrc", "Makefile"), os.path.join(here, "csrc", "phonic_gpu.map"), os.path.join(os.path.dirname(here), "include", "phonic_gpu.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libphonic_gpu.so")


def load():
    """Load libphonic_gpu.so (HIP). Raises if it has not been built — no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PHONIC_LIB: another build of the SAME library (tools/ab_libs/*.so: A/B runs of kernel variants on one box) — still HIP, still no fallback
    path = os.environ.get("PHONIC_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). phonic_amd has no CPU fallback."
        )
    _preload_hip_runtime()
    lib = C.CDLL(path)
    declare(lib, "pg_")
    P = C.POINTER
    vp = C.c_void_p
    lib.pg_last_error_message.restype = C.c_char_p
    lib.pg_last_error_message.argtypes = []
    lib.pg_device_count.restype = C.c_int
    lib.pg_effect_create.restype = vp
    lib.pg_effect_create.argtypes = [C.c_int, P(EffectInit), C.c_int]
    lib.pg_effect_debug_index_log.restype = C.c_int
    lib.pg_effect_debug_index_log.argtypes = [vp, P(C.c_int32), C.c_size_t]
    lib.pg_effect_process_started.restype = C.c_int
    lib.pg_effect_process_started.argtypes = [vp]
    lib.pg_effect_process_stopped.restype = C.c_int
    lib.pg_effect_process_stopped.argtypes = [vp]
    lib.pg_effect_kind_name.restype = C.c_char_p
    lib.pg_effect_kind_name.argtypes = [C.c_int]
    lib.pg_effect_kind_weight.restype = C.c_int
    lib.pg_effect_kind_weight.argtypes = [C.c_int]
    lib.pg_effect_kind_param_count.restype = C.c_int
    lib.pg_effect_kind_param_count.argtypes = [C.c_int]
    lib.pg_effect_kind_param.restype = C.c_int
    lib.pg_effect_kind_param.argtypes = [C.c_int, C.c_int, P(ParamDesc)]
    lib.pg_voice_options_default.restype = None
    lib.pg_voice_options_default.argtypes = [P(VoiceOptions)]
    lib.pg_graph_write_device.restype = C.c_size_t
    lib.pg_graph_write_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, vp]
    lib.pg_graph_set_defer_bus.restype = C.c_int
    lib.pg_graph_set_defer_bus.argtypes = [vp, C.c_int]
    lib.pg_graph_process_bus_device.restype = C.c_int
    lib.pg_graph_process_bus_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, vp]
    lib.pg_graph_synchronize.restype = C.c_int
    lib.pg_graph_synchronize.argtypes = [vp]
    lib.pg_graph_voice_count.restype = C.c_int
    lib.pg_graph_voice_count.argtypes = [vp]
    lib.pg_graph_deferred_units.restype = C.c_int
    lib.pg_graph_deferred_units.argtypes = [vp]
    lib.pg_graph_is_voice_playing.restype = C.c_int
    lib.pg_graph_is_voice_playing.argtypes = [vp, C.c_int]
    lib.pg_graph_kernel_ms.restype = C.c_double
    lib.pg_graph_kernel_ms.argtypes = [vp, C.c_int, P(C.c_uint64)]
    lib.pg_graph_kernel_stats.restype = C.c_int
    lib.pg_graph_kernel_stats.argtypes = [vp, C.c_int, P(C.c_double), P(C.c_uint64), P(C.c_uint64)]
    lib.pg_graph_export_audible.restype = C.c_int
    lib.pg_graph_export_audible.argtypes = [vp, vp, C.c_int, vp]
    lib.pg_graph_audible_words.restype = C.c_int
    lib.pg_graph_audible_words.argtypes = [vp]
    lib.pg_graph_next_main_event.restype = C.c_uint64
    lib.pg_graph_next_main_event.argtypes = [vp, C.c_uint64]
    lib.pg_graph_process_bus_device_flags.restype = C.c_int
    lib.pg_graph_process_bus_device_flags.argtypes = [vp, vp, C.c_size_t, C.c_uint64, vp, vp, C.c_int]
    lib.pg_graph_dynamic_stats.restype = C.c_int
    lib.pg_graph_dynamic_stats.argtypes = [vp, C.c_int, P(C.c_uint64), P(C.c_double), P(C.c_uint64)]
    lib.pg_graph_bus_kernel_stats.restype = C.c_int
    lib.pg_graph_bus_kernel_stats.argtypes = [vp, C.c_int, P(C.c_double), P(C.c_uint64), P(C.c_uint64)]
    lib.pg_graph_bus_kernel.restype = C.c_char_p
    lib.pg_graph_bus_kernel.argtypes = [vp]
    lib.pg_graph_set_max_blocks_per_launch.restype = C.c_int
    lib.pg_graph_set_max_blocks_per_launch.argtypes = [vp, C.c_int]
    lib.pg_sharded_create.restype = vp
    lib.pg_sharded_create.argtypes = [C.c_uint32, C.c_uint32, C.c_size_t, P(C.c_int), C.c_int]
    lib.pg_sharded_destroy.restype = None
    lib.pg_sharded_destroy.argtypes = [vp]
    for name, args in (("shard_count", []), ("set_max_blocks_per_launch", [C.c_int]), ("add_mixer", []), ("add_mixer_to", [C.c_int]),
                       ("add_effect", [C.c_int, C.c_int, P(EffectInit)]), ("add_voice", [C.c_int, P(C.c_float), C.c_size_t, C.c_uint32, C.c_uint32, P(VoiceOptions)]),
                       ("shard_of_mixer", [C.c_int]), ("schedule_param", [C.c_int, C.c_uint32, C.c_float, C.c_int, C.c_uint64]), ("schedule_reset", [C.c_int, C.c_uint64]),
                       ("set_voice_volume", [C.c_int, C.c_float, C.c_uint64]), ("set_voice_panning", [C.c_int, C.c_float, C.c_uint64]),
                       ("set_voice_speed", [C.c_int, C.c_double, C.c_float, C.c_uint64]), ("seek_voice", [C.c_int, C.c_double, C.c_uint64]),
                       ("remove_mixer", [C.c_int]), ("remove_effect", [C.c_int]), ("move_effect", [C.c_int, C.c_int, C.c_int, C.c_int]),
                       ("set_reduce", [C.c_int]), ("reduce_mode", []), ("is_voice_playing", [C.c_int]),
                       ("stop_voice", [C.c_int, C.c_uint64]), ("remove_voice", [C.c_int]), ("stop_all_voices", []), ("synchronize", []), ("device_errors", [])):
        fn = getattr(lib, "pg_sharded_" + name)
        fn.restype = C.c_int
        fn.argtypes = [vp] + args
    lib.pg_sharded_add_stream_voice.restype = C.c_int
    lib.pg_sharded_add_stream_voice.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, P(VoiceOptions)]
    lib.pg_sharded_feed_voice.restype = C.c_int
    lib.pg_sharded_feed_voice.argtypes = [vp, C.c_int, P(C.c_float), C.c_size_t]
    lib.pg_sharded_end_stream_voice.restype = C.c_int
    lib.pg_sharded_end_stream_voice.argtypes = [vp, C.c_int]
    lib.pg_sharded_stream_voice_consumed.restype = C.c_int64
    lib.pg_sharded_stream_voice_consumed.argtypes = [vp, C.c_int]
    lib.pg_sharded_write.restype = C.c_size_t
    lib.pg_sharded_write.argtypes = [vp, P(C.c_float), C.c_size_t, C.c_uint64]
    lib.pg_sharded_write_device.restype = C.c_size_t
    lib.pg_sharded_write_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64]
    lib.pg_graph_add_stream_voice.restype = C.c_int
    lib.pg_graph_add_stream_voice.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, P(VoiceOptions)]
    lib.pg_graph_feed_voice.restype = C.c_int
    lib.pg_graph_feed_voice.argtypes = [vp, C.c_int, P(C.c_float), C.c_size_t]
    lib.pg_graph_end_stream_voice.restype = C.c_int
    lib.pg_graph_end_stream_voice.argtypes = [vp, C.c_int]
    lib.pg_graph_stream_voice_consumed.restype = C.c_int64
    lib.pg_graph_stream_voice_consumed.argtypes = [vp, C.c_int]
    lib.pg_debug_fail_launch_round.restype = None
    lib.pg_debug_fail_launch_round.argtypes = [C.c_int]
    lib.pg_debug_hip_calls.restype = None
    lib.pg_debug_hip_calls.argtypes = [P(C.c_uint64)]
    lib.pg_graph_device_errors.restype = C.c_int
    lib.pg_graph_device_errors.argtypes = [vp]
    lib.pg_graph_set_fast_math.restype = C.c_int
    lib.pg_graph_set_fast_math.argtypes = [vp, C.c_int]
    lib.pg_graph_set_timing_period.restype = C.c_int
    lib.pg_graph_set_timing_period.argtypes = [vp, C.c_int]
    lib.pg_graph_dominant_kernel.restype = C.c_char_p
    lib.pg_graph_dominant_kernel.argtypes = [vp]
    lib.pg_graph_set_staged.restype = C.c_int
    lib.pg_graph_set_staged.argtypes = [vp, C.c_int]
    _LIB = lib
    return lib
