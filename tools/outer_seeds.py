"""(box) a seed of the file-feature fuzz family with one action removed at a time: which action a GPU-vs-oracle difference needs.
usage: python tools/outer_seeds.py SEED"""
import copy, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle
import test_gpu_fuzz as F
from phonic_amd.graph import Graph
seed = int(sys.argv[1])
plan0 = F.make_voice_plan(seed)
def run(plan):
    b = F.render_voice_plan(plan, oracle.OracleGraph(F.SR, 2, 1024))
    a = F.render_voice_plan(plan, Graph(F.SR, 2, 1024, 0))
    big = np.flatnonzero(np.abs(a.astype(np.float64) - b) > 1e-6)
    return big.size, (int(big[0] // 2) if big.size else None), float(np.abs(a.astype(np.float64) - b).max())
print("whole", run(plan0))
acts = sorted(plan0["actions"])
for i, a_ in enumerate(acts):
    p = copy.deepcopy(plan0); p["actions"] = [x for j, x in enumerate(acts) if j != i]
    print("without", a_, "->", run(p))
for i, v in enumerate(plan0["voices"]):
    p = copy.deepcopy(plan0); p["voices"] = [x for j, x in enumerate(plan0["voices"]) if j != i]
    p["actions"] = [(b, k, (vi if vi < i else vi - 1), x, t) for (b, k, vi, x, t) in plan0["actions"] if vi != i]
    if p["voices"]: print("without voice", i, v["tone"], v["opt"], "->", run(p))
p = copy.deepcopy(plan0); p["sizes"] = [1024] * ((sum(plan0["sizes"]) + 1023) // 1024)
print("1024-frame blocks only (actions keep their block index)", run(p))
