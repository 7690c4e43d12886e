"""(box, -DPG_DIAG build) shader-clock stamps of the BUS workgroup of a C2 / C4 graph: where a lone workgroup spends a block of the main mixer's chain."""
import sys, ctypes as C
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
from phonic_amd.graph import Graph
from phonic_amd import _capi
import workloads
wl=sys.argv[1] if len(sys.argv)>1 else 'c4'
g=Graph(48000,2,1024,0)
(workloads.build_c4 if wl=='c4' else workloads.build_c2)(g)
lib=_capi.load()
lib.pg_graph_diag.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.c_int]
buf=(C.c_uint64*64)()
lib.pg_graph_diag(g._h,buf,64)
bus=torch.zeros(2048,device='cuda:0')
pos=0
for i in range(20):
    g.write_device(bus.data_ptr(),2048,pos); pos+=1024
g.synchronize()
lib.pg_graph_diag(g._h,buf,64)
t=[int(buf[i]) for i in range(64)]
names={0:'start',1:'after source / bus load',8:'fx staged',9:'processor logic',24:'comp: start',25:'comp: peaks',26:'comp: tracked peak',27:'comp: window max + dB',28:'comp: serial envelope',29:'comp: gain + output',30:'comp: line written',
       10:'reverb_params',11:'t_max',12:'rec setup',2:'rev: chunk setup done',3:'rev: predelay done',4:'rev: biquadA done',5:'rev: phase3 done',6:'rev: epilogue done',7:'rev: B/asin/C/mix done',14:'effects done',15:'end'}
order=[0,1,8,9,24,25,26,27,28,29,30,14,15] if wl=='c4' else [0,1,8,9,10,11,12,2,3,4,5,6,7,14,15]
prev=t[0]
for k in order:
    print(f"{names[k]:28s} +{(t[k]-prev):8d} cyc  (t={t[k]-t[0]})")
    prev=t[k]
