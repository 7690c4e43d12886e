# per-workgroup start / stage ends of the staged kernel's last launch (diagnostic build, -DPG_DIAG); s_memrealtime ticks = 10 ns
import sys, ctypes as C, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
from phonic_amd.graph import Graph
from phonic_amd import _capi
import workloads
V=1024
F=int(sys.argv[sys.argv.index('--frames')+1]) if '--frames' in sys.argv else 1024   # frames per call (= max_frames): a host's callback size
g=Graph(48000,2,F,0)
workloads.build_headline(g,V,0,V,2.0)
lib=_capi.load()
lib.pg_graph_diag.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.c_int]
N=64+4*4096
buf=(C.c_uint64*N)()
lib.pg_graph_diag(g._h,buf,N)
bus=torch.zeros(2*F,device='cuda:0')
pos=0
for i in range(12):
    g.write_device(bus.data_ptr(),2*F,pos); pos+=F
g.synchronize()
lib.pg_graph_diag(g._h,buf,N)
a=np.array(buf[64:64+4*V],dtype=np.int64).reshape(V,4)
t0=a[:,0].min()
us=(a-t0)/100.0
print("starts  us: min %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(us[:,0],[0,50,90,100])))
for i,n in enumerate(["stage1","stage2","stage3"]):
    d=us[:,i+1]-us[:,i]
    print("%s dur us: min %.1f p50 %.1f p90 %.1f max %.1f" % ((n,)+tuple(np.percentile(d,[0,50,90,100]))))
print("ends    us: min %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(us[:,3],[0,50,90,100])))
tot=us[:,3]-us[:,0]
print("per-WG total us: min %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(tot,[0,50,90,100])))
late=np.argsort(us[:,0])[-8:]
print("latest starters (slot, start, end):", [(int(s), round(float(us[s,0]),1), round(float(us[s,3]),1)) for s in late])
# where do the slow workgroups sit? (dispatch order: workgroup i -> XCD i % 8)
d2=us[:,2]-us[:,1]
slots=np.arange(V)
print("stage2 p50 by XCD (slot % 8):", [round(float(np.median(d2[slots%8==x])),1) for x in range(8)])
print("end    p50 by XCD (slot % 8):", [round(float(np.median(us[slots%8==x,3])),1) for x in range(8)])
print("stage2 p50 by dispatch quarter (slot // 256):", [round(float(np.median(d2[slots//256==q])),1) for q in range(4)])
print("end    p50 by dispatch quarter:", [round(float(np.median(us[slots//256==q,3])),1) for q in range(4)])
print("end    max by dispatch quarter:", [round(float(np.max(us[slots//256==q,3])),1) for q in range(4)])
hi=float(us[:,3].max()); h,e=np.histogram(us[:,3],bins=12,range=(0.5*hi,hi)); print("ends histogram from %.0f us in steps of %.1f us:" % (0.5*hi,(hi-0.5*hi)/12), h.tolist())
print("stage-1 end p50/p90/max: %.1f %.1f %.1f ; stage-2 end p50/p90/max: %.1f %.1f %.1f" % (tuple(np.percentile(us[:,1],[50,90,100]))+tuple(np.percentile(us[:,2],[50,90,100]))))
