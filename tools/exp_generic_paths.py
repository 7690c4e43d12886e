"""(box) What the paths that always render on the generic kernel cost: units holding a ResampledSource-backed voice
(`pg_voice_options::source_rate`) and mixers with sub-mixers of their own (`static_defer`). Same voices and effect as the
headline (stereo file -> cubic -> gain/pan -> per-voice Reverb), 1024-frame blocks, one call per block and 16 blocks per call.

  python tools/exp_generic_paths.py [units]      -> one JSON line per variant
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from phonic_amd import _capi, workloads
from phonic_amd.graph import Graph

V = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BLOCK = 1024


def build(kind, g):
    vol = workloads.voice_level(V)
    for i in range(V):
        seeds = workloads.reverb_seeds(i)
        pan = float(np.float32(workloads.voice_pan(i)))
        if kind == "headline":          # the staged kernels
            m = g.add_mixer()
            g.add_effect(m, _capi.FX_REVERB, reverb_seeds=seeds)
            g.add_voice(m, workloads.tone_buffer(i, 44100, 2.0), 2, 44100, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
        elif kind == "resampled":       # file source running at 32 kHz behind a ResampledSource 32 k -> 48 k (a4), plus its own cubic 44.1 -> 32 k
            m = g.add_mixer()
            g.add_effect(m, _capi.FX_REVERB, reverb_seeds=seeds)
            g.add_voice(m, workloads.tone_buffer(i, 44100, 2.0), 2, 44100, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER,
                        source_rate=32000)
        elif kind.startswith("pitch"):  # a sampler's notes: the headline layout at another playback speed (pitchNN = speed N.N; ratio = 0.919 * speed)
            m = g.add_mixer()
            g.add_effect(m, _capi.FX_REVERB, reverb_seeds=seeds)
            g.add_voice(m, workloads.tone_buffer(i, 44100, 2.0), 2, 44100, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER,
                        speed=float(kind[5:]) / 10.0)
        elif kind == "stream":          # the same PCM fed by the host into a device ring (pg_graph_add_stream_voice): 2 s up front, not looped
            m = g.add_mixer()
            g.add_effect(m, _capi.FX_REVERB, reverb_seeds=seeds)
            pcm = workloads.tone_buffer(i, 48000, 2.0)
            sv = g.add_stream_voice(m, 2, 48000, len(pcm) // 2, volume=vol, panning=pan)
            g.feed_voice(sv, pcm)
        elif kind == "nested":          # the reverb sits on a parent mixer, the voice on a sub-mixer of it
            parent = g.add_mixer()
            g.add_effect(parent, _capi.FX_REVERB, reverb_seeds=seeds)
            child = g.add_mixer(parent)
            g.add_voice(child, workloads.tone_buffer(i, 44100, 2.0), 2, 44100, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)


def run(kind, per_call):
    g = Graph(48000, 2, BLOCK, 0)
    g.set_max_blocks_per_launch(32)
    build(kind, g)
    bus = torch.zeros(per_call * BLOCK * 2, device="cuda:0")
    pos = 0
    for _ in range(4):
        g.write_device(bus.data_ptr(), bus.numel(), pos)
        pos += per_call * BLOCK
    g.synchronize()
    calls = max(4, 64 // per_call)
    t0 = time.perf_counter()
    for _ in range(calls):
        g.write_device(bus.data_ptr(), bus.numel(), pos)
        pos += per_call * BLOCK
    g.synchronize()
    dt = time.perf_counter() - t0
    ms = dt * 1e3 / (calls * per_call)
    print(json.dumps({"variant": kind, "units": V, "blocks_per_call": per_call, "ms_per_block": round(ms, 4),
                      "voice_frames_per_s": round(V * BLOCK / (ms * 1e-3))}), flush=True)
    g.close()


KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ("headline", "stream", "resampled", "nested")
for kind in KINDS:
    for per_call in (1, 16):
        run(kind, per_call)
