"""(box, -DPG_DIAG build) shader-clock stamps of workgroup 0 of the C3 graph (1024 mono voices, Filter -> Chorus): where a unit's block goes."""
import sys, ctypes as C
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
from phonic_amd.graph import Graph
from phonic_amd import _capi
import workloads
V=int(sys.argv[1]) if len(sys.argv)>1 else 1024
K=int(sys.argv[2]) if len(sys.argv)>2 else 1    # blocks per call: > 1 = super-block launches (the stamps are those of the launch's LAST block)
g=Graph(48000,2,1024,0)
g.set_max_blocks_per_launch(max(K,1))
workloads.build_c3(g,V,0,V)
lib=_capi.load()
lib.pg_graph_diag.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.c_int]
buf=(C.c_uint64*64)()
lib.pg_graph_diag(g._h,buf,64)
bus=torch.zeros(2048*K,device='cuda:0')
pos=0
for i in range(20):
    g.write_device(bus.data_ptr(),2048*K,pos); pos+=1024*K
g.synchronize()
lib.pg_graph_diag(g._h,buf,64)
t=[int(buf[i]) for i in range(64)]
names={0:'start',16:'voice staged',17:'schedule done',18:'window filled',19:'interp done',1:'after source',8:'fx staged (last effect)',9:'processor logic',24:'chorus: start',25:'chorus: phases',26:'chorus: svf scan',
       27:'chorus: chunk 0 taps',28:'chorus: chunk 0 writes',29:'chorus: chunk 1 taps',30:'chorus: chunk 1 writes',14:'effects done',15:'end'}
prev=t[0]
for k in [0,16,17,18,19,1,8,9,24,25,26,27,28,29,30,14,15]:
    print(f"{names[k]:28s} +{(t[k]-prev):8d} cyc  (t={t[k]-t[0]})")
    prev=t[k]
