"""scratch: seed 301229's sub-mixer 2 (Compressor -> Chorus -> Distortion, the Distortion moved to the front before block 3) in variants"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import workloads  # noqa: E402
from phonic_amd import _capi  # noqa: E402
from phonic_amd.graph import Graph  # noqa: E402

seed, sr, mf = 301229, 44100, 4096
plan = F.make_plan(seed)
sizes = [2048, 4096, 4096, 64, 2048, 1365, 1, 4096, 4096]
chain, voices = plan["mixers"][2]
print([( _capi.FX_NAMES[k], p) for (k, p, s) in chain], voices)


def render(g, order, move_at=None, move_idx=None, move_by=-3, exact=False):
    m = g.add_mixer()
    ids = [g.add_effect(m, chain[i][0], params=chain[i][1]) for i in order]
    for (ti, rate, vol, pan) in voices:
        g.add_voice(m, workloads.tone_buffer(ti, rate, 0.12), 2, rate, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    chunks, pos = [], 0
    for b, n in enumerate(sizes):
        if move_at is not None and b == move_at:
            g.move_effect(ids[move_idx], m, _capi.MOVE_DIRECTION, move_by)
        o = np.zeros(2 * n, np.float32)
        assert g.write(o, pos) in (0, 2 * n)
        chunks.append(o)
        pos += n
    return np.concatenate(chunks)


def case(label, *a, **kw):
    g = Graph(sr, 2, mf, 0)
    if kw.pop("exact", False):
        g.set_fast_math(0)
    x = render(g, *a, **kw)
    y = render(oracle.OracleGraph(sr, 2, mf), *a, **kw)
    d = x.astype(np.float64) - y.astype(np.float64)
    bad = np.nonzero(np.abs(d) > 1e-6)[0]
    print(f"{label}: rms {np.sqrt(np.mean(d * d)):.3e} errors {g.device_errors()} first |d| > 1e-6 at sample {int(bad[0]) if bad.size else None} (frame {int(bad[0]) // 2 if bad.size else None})")
    if bad.size:
        i = int(bad[0]) & ~1
        print("   gpu   ", x[i - 4:i + 8])
        print("   oracle", y[i - 4:i + 8])
    return x, y


case("as the seed: [C, Ch, D], D moved to the front before block 3", [0, 1, 2], 3, 2)
case("[D, C, Ch] from the start", [2, 0, 1])
case("[C, D] + move", [0, 2], 3, 1)
case("[Ch, D] + move", [1, 2], 3, 1)
case("[C, Ch, D], D moved by -1 (between C and Ch)", [0, 1, 2], 3, 2, -1)
case("[C, Ch, D], Ch moved by +1 (to the end)", [0, 1, 2], 3, 1, 1)
case("[C, Ch, D], move before block 1", [0, 1, 2], 1, 2)
