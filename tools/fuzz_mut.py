"""usage (GPU box): python tools/fuzz_mut.py SEED SR MAX_FRAMES — a flat fuzz seed of the `rates` family whose difference needs its chain mutations:
which mutation, which sub-mixer, with / without the parameter events, first differing frames."""
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

seed, sr, mf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(41000 + seed)
assert int(rng.choice([22050, 44100, 96000])) == sr and int(rng.choice([256, 512, 2048, 4096])) == mf
plan = F.make_plan(seed)
plan["sizes"] = [int(rng.choice([mf, mf, mf // 2, max(1, mf // 3), 64, 1])) for _ in range(9)]


def render(g, keep_mixers=None, muts=(0, 1), events=True, exact=False):
    seed, descs, sizes, ev_block = plan["seed"], plan["descs"], plan["sizes"], plan["ev_block"]
    fx_ids, voice_ids, fx_mixer = [], [], {}
    for mi, (chain, voices) in enumerate(plan["mixers"]):
        on = keep_mixers is None or mi in keep_mixers
        m = g.add_mixer() if on else None
        for (k, p, s) in chain:
            fx_ids.append((g.add_effect(m, k, params=p, reverb_seeds=workloads.reverb_seeds(s) if k == _capi.FX_REVERB else None) if on else None, k))
            if on:
                fx_mixer[fx_ids[-1][0]] = m
        for (ti, rate, vol, pan) in voices:
            voice_ids.append(g.add_voice(m, workloads.tone_buffer(ti, rate, 0.12), 2, rate, volume=vol, panning=pan, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER) if on else None)
    for (k, p, s) in plan["bus"]:
        fx_ids.append((g.add_effect(0, k, params=p, reverb_seeds=workloads.reverb_seeds(s) if k == _capi.FX_REVERB else None), k))
        fx_mixer[fx_ids[-1][0]] = 0
    chunks, pos = [], 0
    rng2 = np.random.default_rng(9000 + seed)
    for b, n in enumerate(sizes):
        if b == ev_block and events:
            if voice_ids and voice_ids[0] is not None:
                g.set_voice_volume(voice_ids[0], 0.3, pos + 17)
            if fx_ids:
                fid, k = fx_ids[seed % len(fx_ids)]
                d = descs[k][0]
                if d["type"] == 0 and fid is not None:
                    g.schedule_param(fid, F.fourcc_str(d["fourcc"]), 0.35, pos + n // 2, normalized=True)
        if b in (ev_block + 1, ev_block + 2) and fx_ids and seed % 3 != 0:
            which = b - ev_block - 1
            fid, k = fx_ids[int(rng2.integers(0, len(fx_ids)))]
            if fid is None or fid in fx_mixer:
                if rng2.random() < 0.6:
                    arg = int(rng2.integers(-3, 4))
                    if which in muts and fid is not None:
                        g.move_effect(fid, fx_mixer[fid], _capi.MOVE_DIRECTION, arg)
                elif which in muts and fid is not None:
                    g.remove_effect(fid)
                    del fx_mixer[fid]
        o = np.zeros(2 * n, np.float32)
        assert g.write(o, pos) in (0, 2 * n)
        chunks.append(o)
        pos += n
    return np.concatenate(chunks)


def case(label, **kw):
    g = Graph(sr, 2, mf, 0)
    if kw.pop("exact", False):
        g.set_fast_math(0)
    a = render(g, **kw)
    b = render(oracle.OracleGraph(sr, 2, mf), **kw)
    d = a.astype(np.float64) - b.astype(np.float64)
    edges = np.cumsum([0] + [2 * n for n in plan["sizes"]])
    per = [float(np.sqrt(np.mean(d[edges[i]:edges[i + 1]] ** 2))) for i in range(len(plan["sizes"]))]
    bad = np.nonzero(np.abs(d) > 1e-6)[0]
    first = None
    if bad.size:
        blk = int(np.searchsorted(edges, bad[0], side="right") - 1)
        first = (blk, int((bad[0] - edges[blk]) // 2), float(a[bad[0]]), float(b[bad[0]]))
    print(f"{label}: rms {np.sqrt(np.mean(d * d)):.3e} errors {g.device_errors()} per block {[f'{x:.1e}' for x in per]} first |d| > 1e-6 at (block, frame, gpu, oracle) {first}")


print("sizes", plan["sizes"], "event block", plan["ev_block"])
case("both mutations")
case("first only", muts=(0,))
case("second only", muts=(1,))
case("both, no events", events=False)
for mi in range(len(plan["mixers"])):
    case(f"mixer {mi} alone, both", keep_mixers={mi})
    case(f"mixer {mi} alone, both, exact", keep_mixers={mi}, exact=True)
