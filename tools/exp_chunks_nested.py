"""(box) Narrow a finding of tools/exp_chunks.py in the nested family: python tools/exp_chunks_nested.py SEED MF — the plan's tree and events, then the exact
serial kernels at max_frames MF against max_frames 4096 with the events removed one at a time."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F  # noqa: E402
from exp_chunks import LONG  # noqa: E402
from phonic_amd import _capi  # noqa: E402
from phonic_amd.graph import Graph  # noqa: E402

seed, mf = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(77000 + seed)
plan = F.make_nested_plan(seed)
plan["sizes"] = [int(rng.choice(LONG)) for _ in range(len(plan["sizes"]))]
edges = np.cumsum([0] + plan["sizes"])
print("sizes", plan["sizes"], "edges", list(map(int, edges)))
for i, (parent, chain, voices) in enumerate(plan["mixers"]):
    print(f"mixer {i + 1}: parent {parent + 1 if parent >= 0 else 0} chain {[_capi.FX_NAMES[k] for (k, _, _) in chain]} voices {[(t, r) for (t, r, _, _) in voices]}")
for (eb, frac, pick, val) in plan["ev_plan"]:
    print(f"event in call {eb} at frame {int(edges[eb]) + int(frac * plan['sizes'][eb])}: pick {pick} ({'fx' if pick % 2 == 0 else 'voice'} {(pick >> 1)}) value {val:.3f}")


def run(p, m, mutations=True):
    g = Graph(48000, 2, m, 0)
    g.set_fast_math(0)
    return F.render_nested_plan(copy.deepcopy(p), g, mutations=mutations)


def report(tag, p, mutations=True):
    a, b = run(p, mf, mutations), run(p, 4096, mutations)
    i = np.flatnonzero(a != b)
    print(f"{tag}: " + (f"{i.size} samples differ, first frame {int(i[0]) // 2}, max {float(np.abs(a - b).max()):.3e}" if i.size else "equal"))


report("all events", plan)
report("no chain mutations", plan, mutations=False)
for k in range(len(plan["ev_plan"])):
    p = copy.deepcopy(plan)
    del p["ev_plan"][k]
    report(f"without event {k}", p)
