"""(box) python tools/exp_super_vs_single.py SEED — the changing-topology plan of a seed in long calls: super-block launches against single launches, per call."""
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

seed = int(sys.argv[1])
rng = np.random.default_rng(77000 + seed)
plan = F.make_topology_plan(seed)
plan["steps"] = [(int(rng.choice(LONG)), acts) for (_, acts) in plan["steps"]]
pos = 0
for n, acts in plan["steps"]:
    print(pos, n, [(a["what"], _capi.FX_NAMES[a["kind"]], round(a["frac"], 3)) for a in acts])
    pos += n
outs = {}
for key, blocks, staged, fast in (("single", 1, 1, 1), ("super", 8, 1, 1), ("single_nostage", 1, 0, 1), ("super_nostage", 8, 0, 1)):
    g = Graph(48000, 2, 1024, 0)
    g.set_staged(staged)
    if blocks > 1:
        g.set_max_blocks_per_launch(blocks)
    outs[key] = F.render_topology_plan(copy.deepcopy(plan), g)
edges = np.cumsum([0] + [2 * n for n, _ in plan["steps"]])
for a, b in (("single", "super"), ("single_nostage", "super_nostage"), ("single", "single_nostage")):
    d = np.abs(outs[a] - outs[b])
    print(a, "vs", b, [(i, f"{float(d[edges[i]:edges[i + 1]].max()):.1e}") for i in range(len(edges) - 1) if d[edges[i]:edges[i + 1]].max() > 0])
    i = np.flatnonzero(d)
    if i.size:
        print("   first frame", int(i[0]) // 2, "last", int(i[-1]) // 2)
