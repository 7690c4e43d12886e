"""(box) Chunk-grid check: the random graphs of tests/test_gpu_fuzz.py pulled in LONG calls (more frames than max_frames, any length, events
anywhere). A write is walked in the reference's chunks (<= 4096 frames from the call's start and from every event) whatever max_frames is:
  * exact serial kernels: max_frames 1024 / 256 / 1000 must equal max_frames 4096 (one piece per chunk: the old, tested path) BIT FOR BIT;
  * time-parallel kernels: super-block launches must equal single launches bit for bit; every configuration within tolerance of the
    oracle pulled in the SAME calls.
usage: python tools/exp_chunks.py [n_seeds] [base] [families: flat,nested,voices,topology]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from phonic_amd.graph import Graph  # noqa: E402

SR = 48000
LONG = [1024, 2048, 3072, 4096, 5000, 700, 2500, 8192, 6144, 333, 4097, 1, 9000, 1500]


def diff(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())


def first_diff(a, b):
    i = np.flatnonzero(a != b)
    return (int(i[0]) // 2, int(i.size)) if i.size else None


def run_family(name, seeds):
    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(77000 + seed)
        if name == "flat":
            plan = F.make_plan(seed)
            plan["sizes"] = [int(rng.choice(LONG)) for _ in range(9)]
            render = lambda g: F.render_plan(plan_copy(plan), g)  # noqa: E731
        elif name == "nested":
            plan = F.make_nested_plan(seed)
            plan["sizes"] = [int(rng.choice(LONG)) for _ in range(len(plan["sizes"]))]
            render = lambda g: F.render_nested_plan(plan_copy(plan), g)  # noqa: E731
        elif name == "voices":
            plan = F.make_voice_plan(seed)
            old_total = sum(plan["sizes"])
            plan["sizes"] = [int(rng.choice(LONG)) for _ in range(len(plan["sizes"]))]
            scale = sum(plan["sizes"]) / max(1, old_total)
            plan["actions"] = [(ab, kind, vi, x, int(t * scale)) for (ab, kind, vi, x, t) in plan["actions"]]
            render = lambda g: F.render_voice_plan(plan_copy(plan), g)  # noqa: E731
        else:
            plan = F.make_topology_plan(seed)
            plan["steps"] = [(int(rng.choice(LONG)), acts) for (_, acts) in plan["steps"]]
            render = lambda g: F.render_topology_plan(plan_copy(plan), g)  # noqa: E731
        ref = render(oracle.OracleGraph(SR, 2, 1024))
        scale = max(1.0, float(np.abs(ref).max()))
        tol_rms, tol_max = (1e-6, 1e-5) if name == "voices" else (1e-5 * scale, 1e-4 * scale)
        outs = {}

        def make(mf, fast=1, blocks=1):
            g = Graph(SR, 2, mf, 0)
            if not fast:
                g.set_fast_math(0)
            if blocks > 1:
                g.set_max_blocks_per_launch(blocks)
            return g

        msgs = []
        for key, mf, fast, blocks in (("s4096", 4096, 0, 1), ("s1024", 1024, 0, 1), ("s256", 256, 0, 1), ("s1000", 1000, 0, 1),
                                      ("f4096", 4096, 1, 1), ("f1024", 1024, 1, 1), ("f1024x8", 1024, 1, 8), ("f512x16", 512, 1, 16)):
            g = make(mf, fast, blocks)
            outs[key] = render(g)
            err = g.device_errors()
            r, m = diff(outs[key], ref)
            if err or not np.isfinite(outs[key]).all() or r > tol_rms or m > tol_max:
                msgs.append(f"{key}: vs oracle rms {r:.3e} max {m:.3e} (tol {tol_rms:.1e} / {tol_max:.1e}) deverr {err}")
        for key in ("s1024", "s256", "s1000"):
            fd = first_diff(outs[key], outs["s4096"])
            if fd:
                msgs.append(f"{key} != s4096: first differing frame {fd[0]}, {fd[1]} samples, max {float(np.abs(outs[key] - outs['s4096']).max()):.3e}")
        sizes = plan["sizes"] if "sizes" in plan else [n for n, _ in plan["steps"]]
        fd = first_diff(outs["f1024x8"], outs["f1024"])
        if fd:
            edges = np.cumsum([0] + [2 * n for n in sizes])
            per_call = [(i, float(np.abs(outs["f1024x8"][edges[i]:edges[i + 1]] - outs["f1024"][edges[i]:edges[i + 1]]).max())) for i in range(len(sizes))]
            msgs.append(f"f1024x8 != f1024: first differing frame {fd[0]}, {fd[1]} samples, max {float(np.abs(outs['f1024x8'] - outs['f1024']).max()):.3e}; calls {[(i, f'{m:.1e}') for i, m in per_call if m > 0]}")
        if msgs:
            bad += 1
            print(f"[{name} {seed}] sizes {sizes} peak {float(np.abs(ref).max()):.3f}\n   " + "\n   ".join(msgs), flush=True)
        else:
            print(f"[{name} {seed}] ok", flush=True)
    print(f"== {name}: {bad} of {len(seeds)} seeds with findings", flush=True)
    return bad


def plan_copy(plan):
    import copy

    return copy.deepcopy(plan)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fams = (sys.argv[3] if len(sys.argv) > 3 else "flat,nested,voices,topology").split(",")
    total = 0
    for fam in fams:
        total += run_family(fam, range(base, base + n))
    sys.exit(1 if total else 0)
