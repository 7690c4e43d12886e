"""N > 1 path on CPU: world_size 2 over gloo. Each rank renders its shard of the voices (the CPU oracle stands in for the
GPU graph: no GPU here), the partial master buses are sum-reduced to rank 0 with phonic_amd.parallel, and rank 0 checks
the result against the unsharded graph. Covers shard_range + reduce_master_bus, the code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_voices, blocks, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import workloads
    from phonic_amd.parallel import reduce_master_bus, shard_range

    start, count = shard_range(n_voices, rank, world)
    g = oracle.OracleGraph(48000, 2, 512)
    workloads.build_headline(g, count, first_voice=start, total_voices=n_voices, seconds=0.1)
    out = []
    pos = 0
    for _ in range(blocks):
        blk = np.zeros(1024, np.float32)
        g.write(blk, pos)
        bus = torch.from_numpy(blk)
        reduce_master_bus(bus, root=0)
        out.append(bus.numpy().copy())
        pos += 512
    if rank == 0:
        q.put(np.concatenate(out))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from phonic_amd.parallel import shard_range

    for n, w in ((1024, 8), (10, 4), (3, 8), (8192, 8), (7, 2)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for s, c in spans:
            assert s == pos
            pos += c


def test_two_rank_voice_sharding_and_bus_reduce():
    n_voices, blocks, world = 5, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_voices, blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import workloads

    g = oracle.OracleGraph(48000, 2, 512)
    workloads.build_headline(g, n_voices, seconds=0.1)
    ref = g.render(blocks, 512)
    # f32 sum order differs (shard sums are reduced instead of one serial sum): reassociation error only
    np.testing.assert_allclose(got, ref, atol=2e-7)
    assert np.abs(ref).max() > 1e-3


def _ring_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phonic_amd.parallel import MasterBusRing

    n, m = 8, 3
    ring = MasterBusRing(n, m, "cpu", n_buffers=2, root=0)
    val = lambda step, r: float((step + 1) * (1 if r == 0 else 100))   # the rank's partial bus of block `step`: a constant

    def run(first, count):
        for step in range(first, first + count):
            ring.slot().fill_(val(step, rank))
            ring.submit()

    snaps = []
    run(0, 5)            # super-block 0 complete (one reduce), super-block 1 holds two blocks
    ring.drain()         # ... reduced as a partial super-block; the next block opens a fresh super-block
    snaps.append([b.clone().numpy() for b in ring.buffers])
    run(5, 7)            # three more super-blocks through the ring of two: buffers are written again behind their reduces
    ring.drain()
    snaps.append([b.clone().numpy() for b in ring.buffers])
    ring.drain()         # nothing in flight: a no-op
    if rank == 0:
        q.put(snaps)
    dist.barrier()
    dist.destroy_process_group()


def test_master_bus_ring_super_block_reduce_two_ranks():
    """phonic_amd.parallel.MasterBusRing — the buffer ring bench.py renders into at N > 1 — over gloo with two ranks: one reduce per
    complete super-block, the partly filled one at drain(), buffer reuse behind the reduce issued a ring round earlier. The root
    ends up with the sum over ranks of every block, in the slot the block was rendered into."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    snaps = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, m = 8, 3
    total = lambda step: float((step + 1) * 101)
    first, second = snaps
    # after 5 blocks + drain: buffer 0 = blocks 0, 1, 2; buffer 1 = blocks 3, 4 (third slot never written)
    for j, step in enumerate((0, 1, 2)):
        assert np.all(first[0][j * n : (j + 1) * n] == total(step))
    for j, step in enumerate((3, 4)):
        assert np.all(first[1][j * n : (j + 1) * n] == total(step))
    assert np.all(first[1][2 * n :] == 0.0)
    # blocks 5 .. 11 start at a super-block boundary (ring position 2 -> buffer 0): 5-7 in buffer 0, 8-10 in buffer 1, 11 in buffer 0 again
    assert np.all(second[1][: n] == total(8)) and np.all(second[1][n : 2 * n] == total(9)) and np.all(second[1][2 * n :] == total(10))
    assert np.all(second[0][: n] == total(11))                       # reduced alone by drain()
    assert np.all(second[0][n : 2 * n] == total(6)) and np.all(second[0][2 * n :] == total(7))   # left from the reduce of blocks 5-7


def _ring_multi_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phonic_amd.parallel import MasterBusRing

    n, m = 4, 6
    ring = MasterBusRing(n, m, "cpu", n_buffers=2, root=0)
    step = 0

    def render(k):  # k consecutive blocks in ONE call, as bench.py renders a super-block
        nonlocal step
        assert 1 <= k <= ring.room()
        view = ring.slots(k)
        assert view.numel() == k * n
        for j in range(k):
            view[j * n:(j + 1) * n] = float((step + j + 1) * (1 if rank == 0 else 1000))
        ring.submit(k)
        step += k

    for k in (4, 2, 6, 3):      # 4 + 2 fill super-block 0 (reduce), 6 = a whole one, 3 open the third
        render(k)
    ring.close()                # ... which ends early: its 3 blocks are reduced, the next block opens the next buffer
    render(3)
    assert ring.room() == 3
    ring.close()                # (what bench.py does when its next call is larger than the room left)
    render(5)
    last_before = ring.last_block().clone()
    ring.drain()
    if rank == 0:
        q.put(([b.clone().numpy() for b in ring.buffers], last_before.numpy(), ring.last_block().clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_master_bus_ring_multi_block_slots_two_ranks():
    """The super-block form of the ring (bench.py renders several blocks per pg_graph_write_device call): slots(k) hands out k consecutive
    block slots of the buffer being filled, submit(k) advances by k and issues the reduce when the super-block is complete, last_block()
    is the block submitted last (summed over ranks on the root once drained)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_multi_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    bufs, last_before, last_after = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, m = 4, 6
    total = lambda step: float((step + 1) * 1001)
    # buffer 0: super-block 0 (blocks 0-5), then the closed one (12-14 over its first three slots), then the last call (18-22);
    # buffer 1: super-block 1 (blocks 6-11), then blocks 15-17 over its first three slots. Every slot holds a sum over both ranks.
    expect0 = [18, 19, 20, 21, 22, 5]
    expect1 = [15, 16, 17, 9, 10, 11]
    for j in range(m):
        assert np.all(bufs[0][j * n:(j + 1) * n] == total(expect0[j])), (j, bufs[0])
        assert np.all(bufs[1][j * n:(j + 1) * n] == total(expect1[j])), (j, bufs[1])
    assert np.all(last_before == 23.0) and np.all(last_after == total(22))


def test_master_bus_ring_schedule_at_eight_ranks():
    """The same ring schedule — whole super-blocks, a super-block that ends early, a partly filled one reduced by drain(), buffer reuse behind
    the reduce issued a ring round earlier — at the node's rank count: EIGHT ranks over gloo (rank counts above two had never run; one
    GPU box cannot hold eight ranks on its card, the control flow does not need one)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_multi_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        bufs, last_before, last_after = q.get(timeout=300)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    n, m = 4, 6
    total = lambda step: float((step + 1) * 7001)      # rank 0 writes (step + 1), the seven others (step + 1) * 1000
    expect0 = [18, 19, 20, 21, 22, 5]
    expect1 = [15, 16, 17, 9, 10, 11]
    for j in range(m):
        assert np.all(bufs[0][j * n:(j + 1) * n] == total(expect0[j])), (j, bufs[0])
        assert np.all(bufs[1][j * n:(j + 1) * n] == total(expect1[j])), (j, bufs[1])
    assert np.all(last_before == 23.0) and np.all(last_after == total(22))
