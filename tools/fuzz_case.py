"""usage (GPU box): python tools/fuzz_case.py SEED [nested|super|voices|rates|topology|long [max_frames]] — where does a fuzz case's GPU-vs-oracle difference come from?

One mode per family of tests/test_gpu_fuzz.py. Flat graphs (default): the seed's plan whole, on the exact serial kernels, one sub-mixer at a time and
with that sub-mixer's chain cut after each effect. nested: whole, exact kernels, without chain mutations, without events, one mixer's chain emptied at
a time. super: super-block pull vs block-by-block pull vs oracle, per block. voices: the file-source plan with its actions, first differing frame.
rates: the flat bisection at the seed's mixer rate and max_frames. topology: shrinks a changing-graph seed to a minimal failing call sequence."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from phonic_amd import _capi  # noqa: E402
from phonic_amd.graph import Graph  # noqa: E402


def diff(a, b, sizes):
    d = a.astype(np.float64) - b.astype(np.float64)
    edges = np.cumsum([0] + [2 * n for n in sizes])
    per = [float(np.sqrt(np.mean(d[edges[i]:edges[i + 1]] ** 2))) for i in range(len(sizes))]
    return float(np.sqrt(np.mean(d * d))), float(np.abs(b).max()), per


def names(chain):
    return [_capi.FX_NAMES[k] for (k, _, _) in chain]


def fmt(per):
    return [f"{x:.1e}" for x in per]


def run(plan, exact=False):
    g = Graph(F.SR, 2, 1024, 0)
    if exact:
        g.set_fast_math(0)
    a = F.render_plan(plan, g)
    b = F.render_plan(plan, oracle.OracleGraph(F.SR, 2, 1024))
    return diff(a, b, plan["sizes"])


def run_nested(plan, exact=False, mutations=True, events=True):
    g = Graph(F.SR, 2, 1024, 0)
    if exact:
        g.set_fast_math(0)
    a = F.render_nested_plan(plan, g, mutations, events)
    b = F.render_nested_plan(plan, oracle.OracleGraph(F.SR, 2, 1024), mutations, events)
    return diff(a, b, plan["sizes"]) + (g.device_errors(),)


seed = int(sys.argv[1])
if len(sys.argv) > 2 and sys.argv[2] == "topology":   # a seed of test_random_topology_changes_while_playing: shrink it to a minimal failing sequence
    plan = F.make_topology_plan(seed)

    def rms_of(p, exact=False):
        g = Graph(F.SR, 2, 1024, 0)
        if exact:
            g.set_fast_math(0)
        try:
            a = F.render_topology_plan(p, g)
            b = F.render_topology_plan(p, oracle.OracleGraph(F.SR, 2, 1024))
        except Exception as e:   # a shrunk sequence may address something that no longer exists
            return -1.0, None
        d = a.astype(np.float64) - b.astype(np.float64)
        edges = np.cumsum([0] + [2 * n for n, _ in p["steps"]])
        return float(np.sqrt(np.mean(d * d))), [float(np.sqrt(np.mean(d[edges[i]:edges[i + 1]] ** 2))) for i in range(len(p["steps"]))]

    base, per = rms_of(plan)
    print("seed", seed, "rms", base, "exact kernels", rms_of(plan, True)[0])
    steps = [(n, list(acts)) for n, acts in plan["steps"]]
    changed = True
    while changed:
        changed = False
        for bi in range(len(steps)):
            for ai in range(len(steps[bi][1]) - 1, -1, -1):
                trial = [(n, [x for j, x in enumerate(acts) if not (i == bi and j == ai)]) for i, (n, acts) in enumerate(steps)]
                r, _ = rms_of({"steps": trial, "descs": plan["descs"]})
                if r > 1e-5:
                    steps = trial
                    changed = True
    while len(steps) > 1 and rms_of({"steps": steps[:-1], "descs": plan["descs"]})[0] > 1e-5:
        steps = steps[:-1]
    r, per = rms_of({"steps": steps, "descs": plan["descs"]})
    print("minimal failing sequence: rms", r, "exact kernels", rms_of({"steps": steps, "descs": plan["descs"]}, True)[0], "per block", fmt(per))
    for bi, (n, acts) in enumerate(steps):
        print(" block", bi, n, "frames")
        for x in acts:
            extra = {"add_effect": (_capi.FX_NAMES[x["kind"]], x["params"]), "add_voice": (x["rate"], "loop" if x["loop"] else "one-shot", round(x["frac"], 3)),
                     "param": (round(x["val"], 3), round(x["frac"], 3), "next block" if x["pick"] & 1 else ""), "move_effect": x["off"]}.get(x["what"], "")
            print("    ", x["what"], "pick", x["pick"], extra)
elif len(sys.argv) > 2 and sys.argv[2] == "rates":   # a seed of test_random_graph_other_rates_and_block_sizes
    rng = np.random.default_rng(41000 + seed)
    sr = int(rng.choice([22050, 44100, 96000]))
    mf = int(rng.choice([256, 512, 2048, 4096]))
    plan = F.make_plan(seed)
    plan["sizes"] = [int(rng.choice([mf, mf, mf // 2, max(1, mf // 3), 64, 1])) for _ in range(9)]
    print("seed", seed, "sample rate", sr, "max_frames", mf, "sizes", plan["sizes"], "event block", plan["ev_block"])
    for i, (chain, voices) in enumerate(plan["mixers"]):
        print(" mixer", i, [(n, p) for n, (_, p, _) in zip(names(chain), chain)], "voices", [(v[1], round(v[2], 2)) for v in voices])
    print(" bus", [(n, p) for n, (_, p, _) in zip(names(plan["bus"]), plan["bus"])])

    def run_r(p, exact=False, mutations=True):
        g = Graph(sr, 2, mf, 0)
        if exact:
            g.set_fast_math(0)
        a = F.render_plan(p, g, mutations=mutations)
        b = F.render_plan(p, oracle.OracleGraph(sr, 2, mf), mutations=mutations)
        return diff(a, b, p["sizes"])
    for label, kw in (("time-parallel kernels", {}), ("exact serial kernels", {"exact": True}), ("no mutations", {"mutations": False})):
        rms, peak, per = run_r(plan, **kw)
        print(f"whole graph, {label}: rms {rms:.3e} peak {peak:.3f} per block {fmt(per)}")
    for i, (chain, voices) in enumerate(plan["mixers"]):
        for cut in range(len(chain) + 1):
            p = copy.copy(plan)
            p["mixers"] = [(chain[:cut], voices)]
            p["bus"] = []
            p["ev_block"] = 99
            rms, peak, per = run_r(p)
            rms_x, _, _ = run_r(p, True)
            print(f"mixer {i} alone, chain {names(chain[:cut])}: rms {rms:.3e} (exact kernels {rms_x:.3e}) peak {peak:.3f} per block {fmt(per)}")
    if plan["bus"]:
        for cut in range(1, len(plan["bus"]) + 1):
            p = copy.copy(plan)
            p["bus"] = plan["bus"][:cut]
            p["mixers"] = [([], v) for _, v in plan["mixers"]]
            p["ev_block"] = 99
            rms, peak, per = run_r(p)
            rms_x, _, _ = run_r(p, True)
            print(f"sources + bus {names(p['bus'])}: rms {rms:.3e} (exact kernels {rms_x:.3e}) per block {fmt(per)}")
elif len(sys.argv) > 2 and sys.argv[2] == "long":   # a flat seed of test_random_graphs_in_long_calls: the plan pulled in calls of 1 .. 9000 frames
    rng = np.random.default_rng(77000 + seed)
    plan = F.make_plan(seed)
    plan["sizes"] = [int(rng.choice(F.LONG_CALLS)) for _ in range(9)]
    mf = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    print("seed", seed, "max_frames", mf, "sizes", plan["sizes"], "event block", plan["ev_block"])
    for i, (chain, voices) in enumerate(plan["mixers"]):
        print(" mixer", i, [(n, p) for n, (_, p, _) in zip(names(chain), chain)], "voices", [(v[1], round(v[2], 2)) for v in voices])
    print(" bus", [(n, p) for n, (_, p, _) in zip(names(plan["bus"]), plan["bus"])])

    def run_l(p, exact=True, mutations=True):
        g = Graph(F.SR, 2, mf, 0)
        if exact:
            g.set_fast_math(0)
        a = F.render_plan(copy.deepcopy(p), g, mutations=mutations)
        b = F.render_plan(copy.deepcopy(p), oracle.OracleGraph(F.SR, 2, 1024), mutations=mutations)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        worst = int(np.argmax(d))
        return diff(a, b, p["sizes"]) + (worst // 2, float(d[worst]), float(b[worst]))
    for label, kw in (("exact serial kernels", {}), ("time-parallel kernels", {"exact": False}), ("exact, no mutations", {"mutations": False})):
        rms, peak, per, wf, wd, wv = run_l(plan, **kw)
        print(f"whole graph, {label}: rms {rms:.3e} peak {peak:.3f} per call {fmt(per)}; worst frame {wf}: |diff| {wd:.3e} at value {wv:.4f}")
    for i, (chain, voices) in enumerate(plan["mixers"]):
        for cut in range(len(chain) + 1):
            p = copy.copy(plan)
            p["mixers"] = [(chain[:cut], voices)]
            p["bus"] = []
            p["ev_block"] = 99
            rms, peak, per, wf, wd, wv = run_l(p)
            print(f"mixer {i} alone, chain {names(chain[:cut])}: rms {rms:.3e} peak {peak:.3f} per call {fmt(per)}; worst frame {wf}: {wd:.3e}")
    if plan["bus"]:
        for cut in range(1, len(plan["bus"]) + 1):
            p = copy.copy(plan)
            p["bus"] = plan["bus"][:cut]
            p["mixers"] = [([], v) for _, v in plan["mixers"]]
            p["ev_block"] = 99
            rms, peak, per, wf, wd, wv = run_l(p)
            print(f"sources + bus {names(p['bus'])}: rms {rms:.3e} per call {fmt(per)}; worst frame {wf}: {wd:.3e}")
elif len(sys.argv) > 2 and sys.argv[2] == "voices":   # a seed of test_random_voice_features_match_oracle
    plan = F.make_voice_plan(seed)
    print("seed", seed, "sizes", plan["sizes"])
    for v in plan["voices"]:
        print(" voice on mixer", v["mixer"], v["tone"], v["opt"])
    pos = np.cumsum([0] + plan["sizes"])
    for a_ in sorted(plan["actions"]):
        print(" action in front of block", a_[0], "(frame", int(pos[a_[0]]), ")", a_[1:])
    a = F.render_voice_plan(plan, Graph(F.SR, 2, 1024, 0))
    b = F.render_voice_plan(plan, oracle.OracleGraph(F.SR, 2, 1024))
    print("rms per block", fmt(diff(a, b, plan["sizes"])[2]))
    bad = np.flatnonzero(a != b)
    print("samples differing:", bad.size, "first frame", bad[0] // 2 if bad.size else None, "last frame", bad[-1] // 2 if bad.size else None)
    big = np.flatnonzero(np.abs(a.astype(np.float64) - b) > 1e-6)
    print("samples with |d| > 1e-6:", big.size, "first frame", big[0] // 2 if big.size else None, "last", big[-1] // 2 if big.size else None)
    if big.size:
        f = big[0] // 2
        for k in range(max(f - 2, 0), f + 6):
            print("  ", k, "gpu", a[2 * k], a[2 * k + 1], "oracle", b[2 * k], b[2 * k + 1])
elif len(sys.argv) > 2 and sys.argv[2] == "super":   # a seed of test_random_graph_superblock_writes
    plan = F.make_plan(seed)
    rng = np.random.default_rng(11000 + seed)
    plan["sizes"] = [1024 * int(rng.integers(1, 5)) for _ in range(7)]
    print("seed", seed, "call sizes in blocks", [n // 1024 for n in plan["sizes"]], "event block", plan["ev_block"])
    for i, (chain, voices) in enumerate(plan["mixers"]):
        print(" mixer", i, [(n, p) for n, (_, p, _) in zip(names(chain), chain)], "voices", [(v[1], round(v[2], 2)) for v in voices])
    print(" bus", [(n, p) for n, (_, p, _) in zip(names(plan["bus"]), plan["bus"])])
    g = Graph(F.SR, 2, 1024, 0)
    g.set_max_blocks_per_launch(4)
    a = F.render_plan(plan, g, events_at_call_start=True)
    a1 = F.render_plan(plan, Graph(F.SR, 2, 1024, 0), split=1024, events_at_call_start=True)
    gx = Graph(F.SR, 2, 1024, 0)
    gx.set_fast_math(0)
    ax = F.render_plan(plan, gx, split=1024, events_at_call_start=True)
    b = F.render_plan(plan, oracle.OracleGraph(F.SR, 2, 1024), split=1024, events_at_call_start=True)
    blocks = [1024] * (len(a) // 2048)
    print("super vs block-by-block, samples differing per block:", [int(np.count_nonzero(a[i * 2048:(i + 1) * 2048] != a1[i * 2048:(i + 1) * 2048])) for i in range(len(blocks))])
    print("super vs block-by-block rms per block:", fmt(diff(a, a1, blocks)[2]))
    print("block-by-block vs oracle rms per block:", fmt(diff(a1, b, blocks)[2]))
    print("exact kernels  vs oracle rms per block:", fmt(diff(ax, b, blocks)[2]))
    print("device errors", g.device_errors())
elif len(sys.argv) > 2 and sys.argv[2] == "nested":
    plan = F.make_nested_plan(seed)
    print("seed", seed, "sizes", plan["sizes"], "events (block, frac, pick, value)", [(e[0], round(e[1], 3), e[2], round(e[3], 3)) for e in plan["ev_plan"]])
    for i, (parent, chain, voices) in enumerate(plan["mixers"]):
        print(" mixer", i, "parent", parent, [(n, p) for n, (_, p, _) in zip(names(chain), chain)], "voices", [(v[1], round(v[2], 2)) for v in voices])
    for label, kw in (("time-parallel kernels", {}), ("exact serial kernels", {"exact": True}), ("no chain mutations", {"mutations": False}),
                      ("no events, no mutations", {"mutations": False, "events": False})):
        rms, peak, per, err = run_nested(plan, **kw)
        print(f"whole graph, {label}: rms {rms:.3e} peak {peak:.3f} device errors {err} per block {fmt(per)}")
    for i, (parent, chain, voices) in enumerate(plan["mixers"]):
        if not chain:
            continue
        p = copy.copy(plan)
        p["mixers"] = [(pa, [] if j == i else ch, vo) for j, (pa, ch, vo) in enumerate(plan["mixers"])]
        rms, peak, per, err = run_nested(p, mutations=False, events=False)
        print(f"without the chain of mixer {i} {names(chain)} (no events, no mutations): rms {rms:.3e} per block {fmt(per)}")
else:
    plan = F.make_plan(seed)
    print("seed", seed, "sizes", plan["sizes"], "event block", plan["ev_block"])
    for i, (chain, voices) in enumerate(plan["mixers"]):
        print(" mixer", i, names(chain), "voices", [(v[1], round(v[2], 2)) for v in voices])
    print(" bus", names(plan["bus"]))
    for label, exact in (("time-parallel kernels", False), ("exact serial kernels", True)):
        rms, peak, per = run(plan, exact)
        print(f"whole graph, {label}: rms {rms:.3e} peak {peak:.3f} per block {fmt(per)}")
    for i, (chain, voices) in enumerate(plan["mixers"]):
        for cut in range(len(chain) + 1):
            p = copy.copy(plan)
            p["mixers"] = [(chain[:cut], voices)]
            p["bus"] = []
            p["ev_block"] = 99  # no events, no chain mutations: the chain as built
            rms, peak, per = run(p)
            rms_x, _, _ = run(p, True)
            print(f"mixer {i} alone, chain {names(chain[:cut])}: rms {rms:.3e} (exact kernels {rms_x:.3e}) peak {peak:.3f} worst block {max(per):.1e}")
