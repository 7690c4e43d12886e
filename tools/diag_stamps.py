import sys, ctypes as C
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
from phonic_amd.graph import Graph
from phonic_amd import _capi
import workloads
V=int(sys.argv[1]) if len(sys.argv)>1 and sys.argv[1].isdigit() else 1024
N=int(sys.argv[sys.argv.index('--frames')+1]) if '--frames' in sys.argv else 1024   # frames per call (= max_frames)
g=Graph(48000,2,N,0)
(workloads.build_c5 if 'c5' in sys.argv else workloads.build_headline)(g,V,0,V,2.0)
lib=_capi.load()
lib.pg_graph_diag.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.c_int]
buf=(C.c_uint64*64)()
lib.pg_graph_diag(g._h,buf,64)
bus=torch.zeros(2*N,device='cuda:0')
pos=0
for i in range(20):
    g.write_device(bus.data_ptr(),2*N,pos); pos+=N
g.synchronize()
lib.pg_graph_diag(g._h,buf,64)
t=[buf[i] for i in range(64)]
print('sched dbg', [hex(buf[i]) for i in range(40,49)])
print('done@last chunk', buf[20], buf[21], 't_max', buf[22], 'mvalid', [buf[24+i] for i in range(16)])
names={0:'start',16:'voice staged',17:'schedule done',18:'window filled',19:'interp done',1:'after source',8:'fx staged',9:'processor logic',10:'reverb_params',11:'t_max',12:'rec setup',2:'rev: chunk setup done',3:'rev: predelay done',4:'rev: biquadA done',5:'rev: phase3 done',6:'rev: epilogue done',7:'rev: B/asin/C/mix done',14:'effects done',15:'end'}
prev=t[0]
names[13]='sched published (stage 1 end)'
if 'c5' in sys.argv:
    names.update({40:'leading fx 0 (Filter)',41:'leading fx 1 (Eq5)',42:'leading fx 2 (Delay)',46:'  Filter: state in LDS',43:'  Filter: chain entered',44:'  Filter: staged as f64',45:'  Filter: scan done',47:'  Filter: processor done'})
for k in ([0,16,17,18,19,1,46,43,44,45,47,40,41,42,8,11,3,14,13,12,2,4,5,6,7,15] if 'c5' in sys.argv else [0,16,17,18,19,1,8,11,3,14,13,12,2,4,5,6,7,15] if '--staged' in sys.argv else [0,16,17,18,19,1,8,9,10,11,12,2,3,4,5,6,7,14,15]):
    print(f"{names[k]:28s} +{(t[k]-prev):8d} cyc  (t={t[k]-t[0]})")
    prev=t[k]

laps=[buf[50+i] for i in range(5)]
t30=[buf[i] for i in (16,36,37,30,31,32,33,34,35,17)]
n30=['voice staged','file_source_write entered','src_write_buffer entered','sched_parallel entered','first barrier passed','closed forms + 3 composes','wave scan (6 shuffle + compose steps)','cross-wave prefix + walk + stores','last barrier passed','schedule done (lane-0 bookkeeping + barrier)']
if t30[1]:
    for i in range(1,len(t30)): print(f"schedule: {n30[i]:44s} +{t30[i]-t30[i-1]:8d} cyc")
print('phase-3 laps of wave 0, summed over 20 blocks (cycles/block):', {n: laps[i]//20 for i,n in enumerate(['line taps issued','ap loads+sin+chain','interp+householder','barrier wait','stores'])})

if '--staged' in sys.argv:
    t3=[buf[i] for i in (6,56,57,58,59,7,60,61,62,15)]
    n3=['stage 2 done','sig reloaded / scan B starts','scan B done','asin done / scan C starts','scan C done','dry mix done','(post logic starts)','post logic done','state written back','unit output stored']
    for i in range(1,len(t3)): print(f"stage 3: {n3[i]:34s} +{t3[i]-t3[i-1]:8d} cyc")
