"""One process per GPU with a deferred bus (bench.py's multi-rank form for graphs with main-mixer effects), two ranks sharing GPU 0 over gloo
(RCCL refuses two ranks on one device; on a node the same code runs over RCCL): voices sharded over the ranks, partial master buses AND the
ranks' `audible` words in ONE sum-reduce, the bus chain on the root with the summed words."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SR, N, PER_CALL, CALLS = 48000, 1024, 4, 15


def _build(g, rank, world):
    """Four main-mixer one-shots (~1.4 blocks long, starting in call 2), spread over the ranks; Gain -> Delay (known tail) on the main mixer."""
    from phonic_amd import _capi, workloads

    for i in range(4):
        if world == 1 or i % world == rank:
            g.add_voice(0, workloads.tone_buffer(i, 48000, 0.03), 2, 48000, volume=0.5, start_time=2 * PER_CALL * N)
    g.add_effect(0, _capi.FX_GAIN, params={"gain": 0.8})
    g.add_effect(0, _capi.FX_DELAY, params={"dlay": 30.0, "fdbk": 0.3, "wet_": 0.6})


def _rank(rank, world, port, with_flags, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from phonic_amd.graph import Graph
    from phonic_amd.parallel import reduce_master_bus

    g = Graph(SR, 2, N, 0)
    g.set_max_blocks_per_launch(PER_CALL)
    g.set_defer_bus(True)
    _build(g, rank, world)
    n = PER_CALL * 2 * N
    outs = []
    for c in range(CALLS):
        buf = torch.zeros(n + PER_CALL, dtype=torch.float32, device="cuda:0")
        pos = c * PER_CALL * N
        w = g.write_device(buf.data_ptr(), n, pos)
        assert w in (0, n)
        g.export_audible(buf.data_ptr() + 4 * n, PER_CALL)       # (all zero when the call had nothing to render)
        g.synchronize()
        reduce_master_bus(buf, root=0)
        if rank == 0:
            if with_flags:
                g.process_bus_device(buf.data_ptr(), n, pos, flags_ptr=buf.data_ptr() + 4 * n, n_words=PER_CALL)
            else:
                g.process_bus_device(buf.data_ptr(), n, pos)
            g.synchronize()
            outs.append(buf[:n].cpu().numpy().copy())
    assert g.device_errors() == 0
    if rank == 0:
        q.put(np.concatenate(outs))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(with_flags):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, with_flags, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = q.get(timeout=240)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return got


def test_two_ranks_reduce_their_audible_words_with_the_bus():
    """The root's bus chain must take EffectProcessor's per-chunk decisions (bypass, tails: src/source/mixed/effect.rs:56-145) from the ranks'
    summed `audible` words as the one main mixer takes them from its own sources: the voices end in call 2, the Delay counts its known tail down
    over the silent chunks and bypasses itself — the sharded render equals the unsharded graph's and the oracle's to the end of the run (exact
    zeros included). Without the words the chain takes every chunk for audible and never bypasses: still the same audio here (it processes
    silence), which is what the ranks did before the words travelled — kept as the second case."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from phonic_amd.graph import Graph

    refs = []
    for g in (Graph(SR, 2, N, 0), oracle.OracleGraph(SR, 2, N)):
        if isinstance(g, Graph):
            g.set_max_blocks_per_launch(PER_CALL)
        _build(g, 0, 1)
        o = np.zeros((CALLS, PER_CALL * 2 * N), np.float32)
        for c in range(CALLS):
            assert g.write(o[c], c * PER_CALL * N) == o[c].size
        refs.append(o.reshape(-1))
    single, ref = refs
    got = _two_ranks(True)
    for want in (single, ref):
        d = got.astype(np.float64) - want.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-6 and float(np.abs(d).max()) <= 1e-5, (float(np.sqrt(np.mean(d * d))), float(np.abs(d).max()))
    blocks = got.reshape(CALLS * PER_CALL, -1)
    assert np.abs(blocks[2 * PER_CALL + 1]).max() > 1e-2 and np.abs(blocks[2 * PER_CALL + 3]).max() > 1e-4
    assert np.array_equal(blocks[-1], single.reshape(CALLS * PER_CALL, -1)[-1]) and np.abs(blocks[-1]).max() == 0.0     # bypassed, like the one mixer's chain
    plain = _two_ranks(False)
    d = plain.astype(np.float64) - ref.astype(np.float64)
    assert float(np.abs(d).max()) <= 1e-5
