"""usage (GPU box): python tools/fuzz_case.py SEED — where does a fuzz graph's GPU-vs-oracle difference come from? Renders the seed's plan
(tests/test_gpu_fuzz.py: make_plan / render_plan) whole, on the exact serial kernels, one sub-mixer at a time and with that sub-mixer's chain cut
after each effect, and prints the RMS difference against the oracle per block."""
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


def run(plan, exact=False):
    g = Graph(F.SR, 2, 1024, 0)
    if exact:
        g.set_fast_math(0)
    a = F.render_plan(plan, g)
    b = F.render_plan(plan, oracle.OracleGraph(F.SR, 2, 1024))
    d = a.astype(np.float64) - b.astype(np.float64)
    per = [float(np.sqrt(np.mean(x * x))) for x in np.array_split(d, len(plan["sizes"]))]
    return float(np.sqrt(np.mean(d * d))), float(np.abs(b).max()), per


def names(chain):
    return [_capi.FX_NAMES[k] for (k, _, _) in chain]


seed = int(sys.argv[1])
plan = F.make_plan(seed)
print("seed", seed, "sizes", plan["sizes"], "event block", plan["ev_block"])
for i, (chain, voices) in enumerate(plan["mixers"]):
    print(" mixer", i, names(chain), "voices", [(v[1], round(v[2], 2)) for v in voices])
print(" bus", names(plan["bus"]))
for label, exact in (("time-parallel kernels", False), ("exact serial kernels", True)):
    rms, peak, per = run(plan, exact)
    print(f"whole graph, {label}: rms {rms:.3e} peak {peak:.3f} per block {[f'{x:.1e}' for x in per]}")
for i, (chain, voices) in enumerate(plan["mixers"]):
    for cut in range(len(chain) + 1):
        p = copy.copy(plan)
        p["mixers"] = [(chain[:cut], voices)]
        p["bus"] = []
        p["ev_block"] = 99  # no events, no chain mutations: the chain as built
        rms, peak, per = run(p)
        rms_x, _, _ = run(p, True)
        print(f"mixer {i} alone, chain {names(chain[:cut])}: rms {rms:.3e} (exact kernels {rms_x:.3e}) peak {peak:.3f} worst block {max(per):.1e}")
