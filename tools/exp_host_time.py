# experiment: is the bench loop host-bound? enqueue time per step vs total time per step (GPU box)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from phonic_amd.graph import Graph
import workloads
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = Graph(48000, 2, 1024, 0)
workloads.build_headline(g, V, 0, V, 2.0)
g.set_timing_period(0)
bus = torch.zeros(2048, device="cuda:0")
rs = torch.cuda.Stream(); torch.cuda.synchronize(); torch.cuda.set_stream(rs); stream = rs.cuda_stream if "--default-stream" not in sys.argv else 0
pos = 0
for i in range(30):
    g.write_device(bus.data_ptr(), 2048, pos, stream); pos += 1024
torch.cuda.synchronize()
for rep in range(3):
    K = 200
    t0 = time.perf_counter()
    for i in range(K):
        g.write_device(bus.data_ptr(), 2048, pos, stream); pos += 1024
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
