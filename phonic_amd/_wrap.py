"""Thin object wrappers over the C ABI (include/phonic_gpu.h).

`EffectHandle` mirrors the reference `Effect` trait (src/effect.rs:86-215) and `GraphHandle` the
main `MixedSource` as driven by `Player` (src/player.rs, src/source/mixed.rs). Both are written
against a (library, prefix) pair so that the test-suite can drive the CPU oracle (prefix `po_`)
with exactly the same call sequence as the product (prefix `pg_`).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import fourcc


class PhonicError(RuntimeError):
    """Maps the reference `Error` enum (src/error.rs:8-22)."""

    def __init__(self, code, msg=""):
        names = {1: "ParameterError", 2: "NotFoundError", 3: "SendError", 4: "DeviceError", 5: "StateError"}
        super().__init__(f"{names.get(code, code)}: {msg}")
        self.code = code


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class EffectHandle:
    """`dyn Effect` (src/effect.rs:86-215)."""

    def __init__(self, lib, prefix, kind, params=None, reverb_seeds=None, device=0, lfo_seed=None):
        self._lib, self._p = lib, prefix
        self.kind = kind
        init = _capi.make_init(params, reverb_seeds, lfo_seed)
        create = getattr(lib, prefix + "effect_create")
        if prefix == "pg_":
            self._h = create(kind, C.byref(init), device)
        else:
            create.restype = C.c_void_p
            create.argtypes = [C.c_int, C.POINTER(_capi.EffectInit)]
            self._h = create(kind, C.byref(init))
        if not self._h:
            raise PhonicError(_capi.PG_ERR_PARAMETER, self._err() or "effect_create failed")
        self.channels = 2

    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _err(self):
        if self._p == "pg_":
            m = self._lib.pg_last_error_message()
            return m.decode() if m else ""
        return ""

    def _check(self, code):
        if code != 0:
            raise PhonicError(code, self._err())

    def name(self):
        return _capi.FX_NAMES[self.kind]

    def initialize(self, sample_rate, channel_count, max_frames):
        self.channels = channel_count
        self._check(self._fn("effect_initialize")(self._h, sample_rate, channel_count, max_frames))

    def process(self, buf, pos_in_frames=0):
        """In place on a contiguous float32 numpy array of interleaved samples."""
        assert buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"]
        self._check(self._fn("effect_process")(self._h, _f32p(buf), buf.size, pos_in_frames))
        return buf

    def process_tail(self):
        t = self._fn("effect_tail")(self._h)
        if t < 0:
            return None
        return t

    def set_parameter(self, id4, value, normalized=False):
        self._check(self._fn("effect_set_parameter")(self._h, fourcc(id4), float(value), 1 if normalized else 0))

    def reset(self):
        self._check(self._fn("effect_message_reset")(self._h))

    def close(self):
        if self._h:
            self._fn("effect_destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphHandle:
    """The main `MixedSource` plus the `Player` calls that populate it."""

    def __init__(self, lib, prefix, sample_rate=48000, channels=2, max_frames=4096, device=0):
        self._lib, self._p = lib, prefix
        self.sample_rate, self.channels = sample_rate, channels
        self._h = getattr(lib, prefix + "graph_create")(sample_rate, channels, max_frames, device)
        if not self._h:
            raise PhonicError(_capi.PG_ERR_DEVICE, self._err() or "graph_create failed")

    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _err(self):
        if self._p == "pg_":
            m = self._lib.pg_last_error_message()
            return m.decode() if m else ""
        return ""

    def _check(self, code):
        if code != 0:
            raise PhonicError(code, self._err())

    def _id(self, v):
        if v < 0:
            raise PhonicError(-v, self._err())
        return v

    def add_mixer(self, parent=None):
        """Player::add_mixer(parent_mixer_id): None / 0 = child of the main mixer."""
        if not parent:
            return self._id(self._fn("graph_add_mixer")(self._h))
        return self._id(self._fn("graph_add_mixer_to")(self._h, parent))

    def add_effect(self, mixer_id, kind, params=None, reverb_seeds=None, lfo_seed=None):
        init = _capi.make_init(params, reverb_seeds, lfo_seed)
        return self._id(self._fn("graph_add_effect")(self._h, mixer_id, kind, C.byref(init)))

    def add_voice(self, mixer_id, pcm, src_channels, src_rate, **opts):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.size % src_channels == 0
        o = _capi.default_voice_options(**opts)
        return self._id(
            self._fn("graph_add_voice")(self._h, mixer_id, _f32p(pcm), pcm.size // src_channels, src_channels, src_rate, C.byref(o))
        )

    def stop_all_voices(self):
        """Player::stop_all_sources (src/player.rs:1012-1045)."""
        self._check(self._fn("graph_stop_all_voices")(self._h))

    def remove_mixer(self, mixer_id):
        """Player::remove_mixer (src/player.rs:825-867)."""
        self._check(self._fn("graph_remove_mixer")(self._h, mixer_id))

    def remove_effect(self, effect_id):
        """Player::remove_effect (src/player.rs:977-990)."""
        self._check(self._fn("graph_remove_effect")(self._h, effect_id))

    def move_effect(self, effect_id, mixer_id, movement, offset=0):
        """Player::move_effect (src/player.rs:942-972): movement = MOVE_DIRECTION (with offset) / MOVE_START / MOVE_END."""
        self._check(self._fn("graph_move_effect")(self._h, effect_id, mixer_id, movement, offset))

    def schedule_param(self, effect_id, id4, value, sample_time, normalized=False):
        self._check(self._fn("graph_schedule_param")(self._h, effect_id, fourcc(id4), float(value), 1 if normalized else 0, sample_time))

    def schedule_reset(self, effect_id, sample_time):
        self._check(self._fn("graph_schedule_reset")(self._h, effect_id, sample_time))

    def set_voice_volume(self, voice, volume, sample_time):
        self._check(self._fn("graph_set_voice_volume")(self._h, voice, float(volume), sample_time))

    def set_voice_panning(self, voice, panning, sample_time):
        self._check(self._fn("graph_set_voice_panning")(self._h, voice, float(panning), sample_time))

    def stop_voice(self, voice, sample_time):
        self._check(self._fn("graph_stop_voice")(self._h, voice, sample_time))

    def remove_voice(self, voice):
        """MixerMessage::RemoveSource: the source leaves its mixer at the start of the next write, at once (no fade)."""
        self._check(self._fn("graph_remove_voice")(self._h, voice))

    def set_voice_speed(self, voice, speed, sample_time, glide=None):
        """FilePlaybackHandle::set_speed(speed, glide): glide in semitones per second, None = immediate."""
        self._check(self._fn("graph_set_voice_speed")(self._h, voice, float(speed), float(glide) if glide else 0.0, sample_time))

    def seek_voice(self, voice, seconds, sample_time):
        self._check(self._fn("graph_seek_voice")(self._h, voice, float(seconds), sample_time))

    def write(self, out, pos_in_frames):
        """`Source::write`: fills `out` (float32 interleaved), returns samples written."""
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        return self._fn("graph_write")(self._h, _f32p(out), out.size, pos_in_frames)

    def render(self, n_blocks, block_frames=1024, start_pos=0):
        """Offline pull loop of the reference WavOutput (src/output/wav.rs:210-250): fixed size blocks."""
        out = np.zeros((n_blocks, block_frames * self.channels), dtype=np.float32)
        pos = start_pos
        for b in range(n_blocks):
            self.write(out[b], pos)
            pos += block_frames
        return out.reshape(-1)

    def close(self):
        if self._h:
            self._fn("graph_destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
