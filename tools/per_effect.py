"""(box) What each of the ten stock effects costs ALONE (VERDICT r04 item 8 / missing 5): V sub-mixers, each one stereo 48 kHz looped file
voice (the resampler's bypass branch) and ONE effect at its defaults — steady state, and one ramping case per effect (a parameter command on
every unit at the head of every second call, at a sample time inside the call: the time-parallel ramp paths or the serial lane). 1024-frame
blocks, 16 blocks per call on a caller's stream; ms per block by wall clock around synchronised calls, the dominant kernel by hipEvents.

  python tools/per_effect.py [units=1024] [effects=all]     -> one JSON line per (effect, case) -> profiles/rNN_per_effect.jsonl

B_alg per voice-frame: 8 B source + the effect's delay-line state (Reverb 416, Delay / Chorus / Compressor 32, others 0)
+ 8 / V of output. `bound`: hbm where the algorithmic stream is the larger part of the time at the access stream's rate, else latency.
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

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BLOCK, PER_CALL = 1024, 16
STATE_B = {"Reverb": 416.0, "Delay": 32.0, "Chorus": 32.0, "Compressor": 32.0}
# (kind, name, init params, ramp parameter, two values to alternate between)
CASES = [
    (_capi.FX_GAIN, "Gain", {"gain": 0.8}, "gain", (0.5, 0.9)),
    (_capi.FX_PANNING, "Panning", {"pan ": 0.2}, "pan ", (-0.4, 0.4)),
    (_capi.FX_FILTER, "Filter", {"type": 0, "cuto": 2000.0, "fltq": 0.707}, "cuto", (800.0, 4000.0)),
    (_capi.FX_EQ5, "Eq5", {"gan1": 3.0, "gan3": -4.0, "gan5": 2.0}, "gan2", (-6.0, 6.0)),
    (_capi.FX_DELAY, "Delay", None, "fdbk", (0.3, 0.6)),
    (_capi.FX_REVERB, "Reverb", None, "wet ", (0.2, 0.5)),
    (_capi.FX_CHORUS, "Chorus", None, "dpth", (0.1, 0.4)),
    (_capi.FX_COMPRESSOR, "Compressor", None, "gain", (3.0, 9.0)),
    (_capi.FX_GATE, "Gate", {"thrs": -40.0}, "thrs", (-45.0, -35.0)),
    (_capi.FX_DISTORTION, "Distortion", {"driv": 1.0}, "driv", (0.5, 2.0)),
]
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None


def run(kind, name, params, ramp_param, ramp_vals, ramp):
    g = Graph(48000, 2, BLOCK, 0)
    g.set_max_blocks_per_launch(PER_CALL)
    g.set_timing_period(1)
    vol = workloads.voice_level(V)
    fx = []
    for i in range(V):
        m = g.add_mixer()
        kw = {"reverb_seeds": workloads.reverb_seeds(i)} if kind == _capi.FX_REVERB else {}
        fx.append(g.add_effect(m, kind, params=params, **kw))
        g.add_voice(m, workloads.tone_buffer(i, 48000, 2.0), 2, 48000, volume=vol * 8.0, panning=float(np.float32(workloads.voice_pan(i))), has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    st = torch.cuda.Stream()
    bus = torch.zeros(PER_CALL * BLOCK * 2, device="cuda:0")
    pos = 0
    with torch.cuda.stream(st):
        for _ in range(4):
            g.write_device(bus.data_ptr(), bus.numel(), pos, st.cuda_stream)
            pos += PER_CALL * BLOCK
        torch.cuda.synchronize()
        g.kernel_stats(reset=True)
        g.dynamic_stats(reset=True)
        calls = 12
        t0 = time.perf_counter()
        for c in range(calls):
            if ramp and c % 2 == 0:
                for k, f in enumerate(fx):
                    g.schedule_param(f, ramp_param, ramp_vals[(c // 2 + k) % 2], pos + 100 + (k * 37) % 900)
            g.write_device(bus.data_ptr(), bus.numel(), pos, st.cuda_stream)
            pos += PER_CALL * BLOCK
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ms, launches, blocks = g.kernel_stats(reset=True)
    dyn = g.dynamic_stats(reset=True)
    err = g.device_errors()
    ms_block = dt * 1e3 / (calls * PER_CALL)
    b_alg = 8.0 + STATE_B.get(name, 0.0) + 8.0 / V
    gbs = b_alg * V * BLOCK / (ms_block * 1e-3) / 1e9
    stream_ms = b_alg * V * BLOCK / 5.6e12 * 1e3   # what the bytes alone would take at the access stream's 5.6 TB/s (profiles/r04/r04_ringstream.jsonl)
    print(json.dumps({"effect": name, "case": "ramp" if ramp else "steady", "units": V, "blocks_per_call": PER_CALL, "ms_per_block": round(ms_block, 4),
                      "voice_frames_per_s": round(V * BLOCK / (ms_block * 1e-3)), "b_alg": round(b_alg, 2), "achieved_gbs": round(gbs, 1), "roofline_frac": round(gbs / 8000.0, 4),
                      "bound": "hbm" if stream_ms > 0.5 * ms_block else "latency", "kernel": g.dominant_kernel(),
                      "kernel_ms_per_block": round(ms * launches / blocks, 4) if blocks else None,
                      "deferred_share": round(dyn["deferred_unit_blocks"] / max(1, dyn["unit_blocks"]), 4), "generic_ms_per_block": round(dyn["generic_ms"] / max(1, dyn["generic_timed"]) * dyn["generic_launches"] / (calls * PER_CALL), 4),
                      "ramp": (f"{ramp_param} command on every unit at the head of every second call" if ramp else None), "device_errors": err}), flush=True)
    g.close()


for (kind, name, params, rp, rv) in CASES:
    if only and name not in only:
        continue
    for ramp in (False, True):
        run(kind, name, params, rp, rv, ramp)
