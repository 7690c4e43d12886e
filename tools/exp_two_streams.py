# experiment: K graphs of 1024/K headline voices each, driven on K streams, stream k delayed by k * offset at the start — does running
# the stages of the two halves out of phase raise throughput?
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
from phonic_amd.graph import Graph
import workloads
V=1024
def run(K, offset_us):
    gs=[]; streams=[torch.cuda.Stream() for _ in range(K)]
    for k in range(K):
        g=Graph(48000,2,1024,0); workloads.build_headline(g, V//K, k*(V//K), V, 2.0); g.set_timing_period(0); gs.append(g)
    buses=[torch.zeros(2048,device='cuda:0') for _ in range(K)]
    pos=0
    for i in range(20):
        for k in range(K): gs[k].write_device(buses[k].data_ptr(), 2048, pos, streams[k].cuda_stream)
        pos+=1024
    torch.cuda.synchronize()
    for k in range(1, K):
        with torch.cuda.stream(streams[k]): torch.cuda._sleep(int(offset_us * k * 2100))   # ~2.1 GHz cycles
    t0=time.perf_counter()
    N=150
    for i in range(N):
        for k in range(K): gs[k].write_device(buses[k].data_ptr(), 2048, pos, streams[k].cuda_stream)
        pos+=1024
    torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    print(f"K={K} offset {offset_us:3d} us: {dt/N*1e3:.4f} ms per 1024-voice block  -> {V*1024*N/dt/1e6:.0f} Mvf/s", flush=True)
for K, off in ((1,0),(2,0),(2,35),(2,70),(4,0),(4,35),(1,0),(2,70),(2,35)):
    run(K, off)
