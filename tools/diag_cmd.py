"""(box) What ONE commanded unit costs: the headline's 1024 voices, one call per block, and in every block exactly one command on one unit —
the reverb's `wet`, the source's volume, its panning — at frame 0 of the block (applied at its head: one segment) or in its middle (the
sub-mixer splits its block there: two segments, mixed.rs:679-712). Per case: ms per step, the generic kernel's time per launch (hipEvents
riding on its dispatch) and the time-parallel kernel's, next to the steady graph — where the 2.5-3.5x of `bench.py --workload dyn` come from.
With PHONIC_LIB pointing at a -DPG_DIAG build: also the shader-clock stamps of the generic kernel's first deferred unit for the last block.

usage: python tools/diag_cmd.py [voices] [blocks]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from phonic_amd import _capi, workloads
from phonic_amd.graph import Graph

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 96
N = 1024
lib = _capi.load()
lib.pg_graph_diag.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
stream_t = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream_t)
out = torch.zeros(2 * N, device="cuda:0")
NAMES = {56: "kernel entry", 57: "unit taken", 58: "before segments", 59: "unit left", 16: "voice staged", 17: "schedule done", 18: "window filled", 19: "interp done", 1: "after source", 8: "fx staged", 9: "processor logic", 11: "block params",
         3: "predelay done", 12: "rec setup", 2: "chunk setup", 4: "anchors", 5: "phase 3 done", 6: "epilogue", 7: "tail done", 14: "effects done", 15: "end"}


def case(kind, off, per_block=1):
    g = Graph(48000, 2, N, 0)
    g.set_timing_period(1)
    plan = workloads.build_dyn(g, V, 4.0)
    buf = (C.c_uint64 * 64)()
    lib.pg_graph_diag(g._h, buf, 64)
    pos = 0
    for _ in range(32):
        g.write_device(out.data_ptr(), 2 * N, pos, stream_t.cuda_stream)
        pos += N
    torch.cuda.synchronize()
    g.kernel_stats(reset=True)
    g.dynamic_stats(reset=True)
    rng = np.random.default_rng(5)
    t0 = time.perf_counter()
    for b in range(NB):
        if kind is not None:
            for j in range(per_block):
                u = int((b * 37 + j * 101) % V)
                t = pos + off
                if kind == "wet":
                    g.schedule_param(plan["reverbs"][u], "wet ", float(np.float32(rng.uniform(0.2, 0.5))), t)
                elif kind == "volume":
                    g.set_voice_volume(plan["voices"][u], float(np.float32(rng.uniform(0.5, 1.0) / 32.0)), t)
                else:
                    g.set_voice_panning(plan["voices"][u], float(np.float32(rng.uniform(-1.0, 1.0))), t)
        g.write_device(out.data_ptr(), 2 * N, pos, stream_t.cuda_stream)
        pos += N
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, launches, blocks = g.kernel_stats(reset=True)
    st = g.dynamic_stats(reset=True)
    lib.pg_graph_diag(g._h, buf, 64)
    t = [buf[i] for i in range(64)]
    line = {"kind": kind or "steady", "frame": off, "commands_per_block": per_block if kind else 0, "ms_per_step": round(dt / NB * 1e3, 4), "fast_kernel_ms": round(ms, 4),
            "generic_ms_per_launch": round(st["generic_ms"] / max(1, st["generic_timed"]), 4), "generic_launches_with_work": st["generic_launches_with_work"],
            "deferred_unit_blocks": st["deferred_unit_blocks"], "device_errors": g.device_errors()}
    print(json.dumps(line))
    if any(t) and kind is not None:
        order = [56, 57, 58, 16, 17, 18, 19, 1, 8, 9, 11, 3, 12, 2, 4, 5, 6, 7, 14, 15, 59]
        have = [(k, t[k]) for k in order if t[k]]
        base = min(v for _, v in have)
        print("   stamps of the generic kernel's first unit, last block (cycles from its first stamp; a split block's second segment overwrites the first's):")
        print("   " + ", ".join(f"{NAMES[k]} {v - base}" for k, v in sorted(have, key=lambda kv: kv[1])))
    g.close()


case(None, 0)
for kind in ("wet", "volume", "pan"):
    for off in (0, 512):
        case(kind, off)
case("wet", 512, per_block=10)
case("volume", 512, per_block=10)
