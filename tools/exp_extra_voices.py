"""(box) What a unit's OTHER voices cost the staged kernel: the headline's 1024 sub-mixers (one looping voice + Reverb each), and in front of /
behind the playing voice an ended one-shot (what a note that restarted leaves in its unit's list: the reference drops an exhausted source from
the mixer, mixed.rs:612-620) or a successor that starts far in the future (Player::play_file_source with a start time, player.rs:519-602).
One call per block and 16-block calls; per variant: ms per step and the time-parallel kernel's time per block.

usage: python tools/exp_extra_voices.py [voices] [blocks]"""
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

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = 1024
stream_t = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream_t)
out = torch.zeros(16 * 2 * N, device="cuda:0")


def case(name, dead_first, dead_last, future, starts_per_block=0, fading=0):
    g = Graph(48000, 2, N, 0)
    g.set_timing_period(1)
    vol = workloads.voice_level(V)
    ms = []
    playing = []
    for i in range(V):
        m = g.add_mixer()
        ms.append(m)
        g.add_effect(m, _capi.FX_REVERB, reverb_seeds=workloads.reverb_seeds(i))
        pan = float(np.float32(workloads.voice_pan(i)))
        for _ in range(dead_first):
            g.add_voice(m, workloads.tone_buffer(i + 3, 44100, 0.05), 2, 44100, volume=vol, panning=pan)
        pv = g.add_voice(m, workloads.tone_buffer(i, 44100, 2.0), 2, 44100, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER,
                         **({"fade_out_seconds": 1.0} if i < fading else {}))
        playing.append(pv)
        for _ in range(dead_last):
            g.add_voice(m, workloads.tone_buffer(i + 5, 44100, 0.05), 2, 44100, volume=vol, panning=pan)
        for _ in range(future):
            g.add_voice(m, workloads.tone_buffer(i + 7, 44100, 2.0), 2, 44100, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER, start_time=10 ** 9)
    # notes that START inside the measured blocks: `starts_per_block` units per block get a second voice whose start time falls into that block
    first_measured = 8 * 16
    rng = np.random.default_rng(3)
    if starts_per_block:
        for b in range(first_measured, first_measured + 2 * NB):
            for j in range(starts_per_block):
                u = (b * 37 + j * 411) % V
                g.add_voice(ms[u], workloads.tone_buffer(u + 11, 44100, 2.0), 2, 44100, volume=vol, panning=0.0, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER,
                            start_time=b * N + int(rng.integers(0, N)))
    # voices in their fade-out through the whole measurement (StopSource with a long fade: the fader arrives ~9 time constants later)
    for i in range(fading):
        g.stop_voice(playing[i * (V // max(1, fading)) % V] if False else playing[i], 4 * N + 17)
    pos = 0
    for _ in range(8):   # (the one-shots end inside the first three blocks)
        g.write_device(out.data_ptr(), 16 * 2 * N, pos, stream_t.cuda_stream)
        pos += 16 * N
    torch.cuda.synchronize()
    res = {"variant": name, "voices_per_unit": 1 + dead_first + dead_last + future}
    for tag, per_call in (("one_call_per_block", 1), ("sixteen_blocks_per_call", 16)):
        g.kernel_stats(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(NB // per_call):
            g.write_device(out.data_ptr(), per_call * 2 * N, pos, stream_t.cuda_stream)
            pos += per_call * N
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, launches, blocks = g.kernel_stats(reset=True)
        res[tag] = {"ms_per_step": round(dt / NB * 1e3, 4), "fast_kernel_ms_per_block": round(ms * launches / max(1, blocks), 4)}
    res["device_errors"] = g.device_errors()
    print(json.dumps(res), flush=True)
    g.close()


case("the playing voice alone", 0, 0, 0)
case("an ended voice in front", 1, 0, 0)
case("an ended voice behind", 0, 1, 0)
case("a successor that has not started", 0, 0, 1)
case("two ended voices in front and a successor", 2, 0, 1)
case("two notes start in every block (a second voice on their units)", 0, 0, 0, starts_per_block=2)
case("46 voices in their fade-out", 0, 0, 0, fading=46)
