# usage (GPU box): PHONIC_LIB=<variant .so> python tools/lib_hash.py [blocks] — sha256 of the master bus the selected library renders for the headline
# (1024 voices), C5 (1024 voices) and C3 (1024 voices) over `blocks` 1024-frame blocks in 16-block calls and again one call per block: two libraries
# that print the same lines render the same bits (the check in front of an interleaved A/B of a kernel variant that must not change the output).
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from phonic_amd.graph import Graph
from phonic_amd import workloads, _capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
for name, build in (("headline", workloads.build_headline), ("c5", workloads.build_c5), ("c3", workloads.build_c3)):
    for per_call in (16, 1):
        g = Graph(48000, 2, 1024, 0)
        build(g, 1024, 0, 1024, 2.0)
        if per_call > 1: g.set_max_blocks_per_launch(per_call)
        bus = torch.zeros(per_call * 2048, device="cuda:0")
        h = hashlib.sha256(); pos = 0
        for _ in range(N // per_call):
            g.write_device(bus.data_ptr(), per_call * 2048, pos, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            h.update(bus.cpu().numpy().tobytes()); pos += per_call * 1024
        print(f"{name:9s} {per_call:2d} blocks per call: {h.hexdigest()[:24]}  (library {_capi.source_hash()})", flush=True)
