"""(box) Narrow a finding of tools/exp_chunks.py in the file-source family: python tools/exp_chunks_voice.py SEED MF — every voice of the plan on its
own (with its actions), exact serial kernels at max_frames MF against max_frames 4096."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from exp_chunks import LONG  # noqa: E402
from phonic_amd.graph import Graph  # noqa: E402

seed, mf = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(77000 + seed)
plan = F.make_voice_plan(seed)
old_total = sum(plan["sizes"])
plan["sizes"] = [int(rng.choice(LONG)) for _ in range(len(plan["sizes"]))]
scale = sum(plan["sizes"]) / max(1, old_total)
plan["actions"] = [(ab, kind, vi, x, int(t * scale)) for (ab, kind, vi, x, t) in plan["actions"]]
print("sizes", plan["sizes"], "edges", list(np.cumsum(plan["sizes"])))
for vi, v in enumerate(plan["voices"]):
    p = copy.deepcopy(plan)
    p["voices"] = [v]
    p["actions"] = [(ab, kind, 0, x, t) for (ab, kind, w, x, t) in plan["actions"] if w == vi]
    outs = []
    for m in (mf, 4096):
        g = Graph(48000, 2, m, 0)
        g.set_fast_math(0)
        outs.append(F.render_voice_plan(copy.deepcopy(p), g))
    ref = F.render_voice_plan(copy.deepcopy(p), oracle.OracleGraph(48000, 2, 1024))
    i = np.flatnonzero(outs[0] != outs[1])
    print(f"voice {vi}: mixer {v['mixer']} tone {v['tone']} opt {v['opt']}\n   actions {p['actions']}")
    print(f"   mf{mf} vs mf4096: {i.size} samples differ" + (f", first frame {int(i[0]) // 2}, max {float(np.abs(outs[0] - outs[1]).max()):.3e}" if i.size else ""),
          f"| mf4096 vs oracle max {float(np.abs(outs[1] - ref).max()):.2e} | mf{mf} vs oracle max {float(np.abs(outs[0] - ref).max()):.2e}")
    if i.size:
        f0 = int(i[0]) // 2
        for name, o in (("mf", outs[0]), ("4096", outs[1]), ("oracle", ref)):
            print("   ", name, np.array2string(o[2 * (f0 - 2):2 * (f0 + 3)], precision=6))
