# experiment: K graphs of 1024/K headline voices each, driven on K streams — does phase mixing across streams raise throughput?
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
from phonic_amd.graph import Graph
import workloads
V=1024
for K in (1, 2, 4, 1, 2, 4):
    gs=[]; streams=[torch.cuda.Stream() for _ in range(K)]
    for k in range(K):
        g=Graph(48000,2,1024,0); workloads.build_headline(g, V//K, k*(V//K), V, 2.0); gs.append(g)
    buses=[torch.zeros(2048,device='cuda:0') for _ in range(K)]
    pos=0
    def step():
        global pos
        for k in range(K):
            gs[k].write_device(buses[k].data_ptr(), 2048, pos, streams[k].cuda_stream)
        pos+=1024
    for i in range(20):
        step()
        if i == 2 and K > 1:   # de-phase the streams once
            for k in range(1, K):
                torch.cuda.synchronize(); 
    torch.cuda.synchronize()
    # offset: let stream 0 run half a block ahead
    t0=time.perf_counter()
    N=100
    for i in range(N): step()
    torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    print(f"K={K}: {dt/N*1e3:.4f} ms per 1024-voice block  -> {V*1024*N/dt/1e6:.0f} Mvf/s")
    del gs
