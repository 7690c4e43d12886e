"""Stamps of a bus chain that is ONE Reverb (the bottleneck stage of C2's chain, alone: its workgroup is block 0 of the bus launch) through the last block of
a 16-block call — diagnostic build (-DPG_DIAG); shader-clock cycles.   usage (GPU box): bash tools/run_diag_bus_chain.sh"""
import sys, ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from phonic_amd.graph import Graph
from phonic_amd import _capi, workloads
g = Graph(48000, 2, 1024, 0)
g.set_max_blocks_per_launch(16)
vol = workloads.voice_level(64)
for i in range(64):
    g.add_voice(0, workloads.tone_buffer(i, 48000, 2.0), 2, 48000, volume=vol, panning=float(np.float32(workloads.voice_pan(i))), has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
g.add_effect(0, _capi.FX_REVERB, reverb_seeds=workloads.reverb_seeds(0))
lib = _capi.load()
lib.pg_graph_diag.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 64)()
lib.pg_graph_diag(g._h, buf, 64)
bus = torch.zeros(2048 * 16, device='cuda:0')
pos = 0
for i in range(6):
    g.write_device(bus.data_ptr(), 2048 * 16, pos); pos += 1024 * 16
g.synchronize()
lib.pg_graph_diag(g._h, buf, 64)
t = [int(buf[i]) for i in range(64)]
names = {60: 'block begins (poll done)', 61: 'input in LDS, next block requested', 11: 'processor pre + block params', 3: 'front: predelay done', 4: 'front: biquad A done / mid: anchors done',
         12: 'mid: records set up', 2: 'mid: chunk length', 5: 'mid: sub-chunks done', 6: 'mid: epilogue done', 56: 'tail begins', 57: 'tail: scan B', 58: 'tail: asin', 59: 'tail: scan C', 7: 'tail: dry mix',
         62: 'processor post done', 63: 'block stored'}
order = sorted((k for k in names if t[k] >= t[60]), key=lambda k: t[k])
prev = t[60]
for k in order:
    print(f"{names[k]:44s} +{t[k] - prev:8d} cyc  (t={t[k] - t[60]})")
    prev = t[k]
laps = [int(buf[50 + i]) for i in range(5)]
print('mid laps of wave 0 (cycles, the last block):', dict(zip(['taps issued', 'ap loads+sin+chain', 'interp+feedback', 'barrier wait', 'stores'], laps)))
