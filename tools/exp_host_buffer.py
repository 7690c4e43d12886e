# experiment (GPU box): the boundary's host-buffer form — pg_graph_write renders into the caller's HOST buffer (one D2H copy of the rendered frames and one
# stream wait per call) — against the device-resident form bench.py times (pg_graph_write_device, inputs and output in HBM): the PCIe-inclusive rate DESIGN §5
# quotes beside `value` (never as `value`). One JSON line per call size.
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from phonic_amd.graph import Graph
from phonic_amd import workloads, _capi
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for blocks in (1, 4, 16):
    res = {}
    for mode in ("device_async", "device_sync", "host"):
        g = Graph(48000, 2, 1024, 0)
        workloads.build_headline(g, V, 0, V, 2.0)
        g.set_timing_period(0)
        if blocks > 1: g.set_max_blocks_per_launch(blocks)
        n = blocks * 2048
        bus = torch.zeros(n, device="cuda:0"); out = np.zeros(n, dtype=np.float32)
        st = torch.cuda.Stream(); s = st.cuda_stream
        def call(pos):
            if mode == "host": g.write(out, pos)
            else:
                g.write_device(bus.data_ptr(), n, pos, s)
                if mode == "device_sync": st.synchronize()
        pos = 0
        for _ in range(64 // blocks + 8): call(pos); pos += blocks * 1024
        torch.cuda.synchronize()
        calls = max(256 // blocks, 32)
        t0 = time.perf_counter()
        for _ in range(calls): call(pos); pos += blocks * 1024
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[mode] = {"ms_per_block": dt / (calls * blocks) * 1e3, "ms_per_call": dt / calls * 1e3, "voice_frames_per_s": V * 1024 * calls * blocks / dt}
    print(json.dumps({"experiment": "host_buffer_write", "voices": V, "blocks_per_call": blocks, "frames_per_call": blocks * 1024, "library": _capi.source_hash(), **res,
                      "host_over_device_sync_us_per_call": (res["host"]["ms_per_call"] - res["device_sync"]["ms_per_call"]) * 1e3}), flush=True)
