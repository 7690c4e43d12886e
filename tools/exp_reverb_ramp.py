"""usage (GPU box): python tools/exp_reverb_ramp.py [room] — cost of a block while every reverb's `wet` (or room size) smoother moves: 256 sub-mixers
Eq5 -> Reverb -> Gain, a command to each in block 3, wall time of the blocks behind it, time-parallel kernels vs the exact serial ones."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phonic_amd import _capi, workloads  # noqa: E402
from phonic_amd.graph import Graph  # noqa: E402

for exact in (False, True):
    g = Graph(48000, 2, 1024, 0)
    if exact:
        g.set_fast_math(0)
    ids = []
    for i in range(256):
        m = g.add_mixer()
        g.add_effect(m, _capi.FX_EQ5, params={"gan3": 2.0})
        ids.append(g.add_effect(m, _capi.FX_REVERB, reverb_seeds=workloads.reverb_seeds(i)))
        g.add_effect(m, _capi.FX_GAIN, params={"gain": 0.8})
        g.add_voice(m, workloads.tone_buffer(i, 44100, 0.3), 2, 44100, volume=0.05, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    o = np.zeros(2048, np.float32)
    times = []
    for b in range(8):
        if b == 3:
            for k, f in enumerate(ids):
                g.schedule_param(f, "room" if "room" in sys.argv[1:] else "wet ", 0.9, b * 1024 + 10 + k)
        t0 = time.perf_counter()
        assert g.write(o, b * 1024) == 2048
        times.append((time.perf_counter() - t0) * 1e3)
    print("exact serial kernels" if exact else "time-parallel kernels", "ms per block:", [round(t, 2) for t in times], "deferred now", g.deferred_units())
