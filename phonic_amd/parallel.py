"""Multi-GPU sharding of the mixer graph: one process per GPU, voices (sub-mixers / sources) are partitioned over
ranks — the reference's own parallel axis is independent sub-mixers (src/source/mixed/submixer/thread_pool.rs:92-121) —
and the partial master buses meet in ONE sum-reduce per block (the caller-side sum of worker outputs,
src/source/mixed.rs:522-536). On ROCm torch.distributed's "nccl" backend is RCCL over xGMI; the same code runs on
gloo for the CPU tests. The per-block message is 8 bytes x frames (8 KiB at 1024 frames): latency bound."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Static block partition: rank r owns voices [start, start + count). Remainders go to the lowest ranks."""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def reduce_master_bus(bus, root=0, group=None):
    """Sum the ranks' partial master-bus blocks into `bus` on `root` (in place). No-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(bus, dst=root, op=dist.ReduceOp.SUM, group=group)
    return bus


def next_call_frames(g, pos, want_frames, group=None):
    """Deferred-bus graphs, one process per GPU: a call must end where a main-mixer event of ANY rank comes due (the one main mixer cuts its
    chunk there for every sub-mixer and for its effect chain, src/source/mixed.rs:679-712). Returns min(want_frames, frames up to the first such
    event over all ranks) — one MIN all-reduce of a single integer (host side; skip it for graphs that never schedule main-mixer events)."""
    t = g.next_main_event(pos)
    n = want_frames if t is None else min(want_frames, t - pos)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        v = torch.tensor([n], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            v = v.cuda()
        dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
        n = int(v.item())
    return n


class MasterBusRing:
    """Master-bus buffers of the sharded render: a ring of `n_buffers` super-blocks of `blocks_per_reduce` blocks each. A rank renders
    block `step` into `slot(step)`; `submit(step)` issues ONE asynchronous sum-reduce to `root` when that block completes its
    super-block (SURVEY §8e "per super-block": offline rendering has no deadline per block; 1 = per block, the real-time setting);
    `slot` waits for the reduce issued `n_buffers` super-blocks ago before its buffer is written again; `drain()` reduces a partly
    filled super-block, waits for everything in flight and lets the next block open a fresh super-block. With RCCL the reduce
    runs on RCCL's own stream, ordered behind the render stream by an event, under the renders of the following super-blocks;
    `Work.wait()` only makes the current stream wait. Without a process group the ring is plain double buffering (world size 1)."""

    def __init__(self, n_samples, blocks_per_reduce, device, n_buffers=4, root=0, group=None, force_distributed=False, extra=0):
        self.n_samples, self.m, self.n_buffers, self.root, self.group = int(n_samples), max(1, int(blocks_per_reduce)), int(n_buffers), root, group
        # `extra` floats of room behind the last block of a buffer: a call's `audible` words travel right behind its samples (slots(k, extra))
        self.extra = int(extra)
        self.buffers = [torch.zeros(self.m * self.n_samples + self.extra, dtype=torch.float32, device=device) for _ in range(self.n_buffers)]
        self.pending = [None] * self.n_buffers
        self.step = 0  # next block (super-block aligned after drain())
        self._last = None
        # (force_distributed: a one-rank group still issues its reduces — exercises the backend where only one GPU can be leased)
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_distributed)

    def _where(self):
        return (self.step // self.m) % self.n_buffers, self.step % self.m

    def room(self):
        """Blocks left in the super-block being filled."""
        return self.m - self.step % self.m

    def slots(self, n_blocks=1, extra=0):
        """The [n_blocks * n_samples (+ extra)] view the next `n_blocks` consecutive blocks are rendered into (n_blocks <= room()); on the root
        it later holds the sum over ranks. extra: floats behind the samples for what rides along in the same reduce (the ranks' `audible`
        words; the room is the next block's until that block is rendered)."""
        k, j = self._where()
        assert 1 <= n_blocks <= self.m - j, (n_blocks, j, self.m)
        assert 0 <= extra <= min(self.extra, self.n_samples), (extra, self.extra)
        if j == 0 and self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        return self.buffers[k][j * self.n_samples : (j + n_blocks) * self.n_samples + extra]

    def slot(self):
        return self.slots(1)

    def submit(self, n_blocks=1):
        """The blocks rendered into `slots(n_blocks)` are complete (enqueued on the current stream): advance, reduce a finished super-block."""
        k, j = self._where()
        self.step += n_blocks
        self._last = (k, j + n_blocks - 1)
        if self.distributed and j + n_blocks == self.m:
            self.pending[k] = dist.reduce(self.buffers[k][: self.m * self.n_samples], dst=self.root, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return k, j

    def close(self):
        """End the super-block being filled early (the caller's next call does not fit into it): its blocks are reduced now, asynchronously,
        and the next block opens the next buffer. Every rank must close at the same step (the bench's call sizes depend on the step only)."""
        k, j = self._where()
        if j != 0:
            if self.distributed:
                self.pending[k] = dist.reduce(self.buffers[k][: j * self.n_samples], dst=self.root, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.step += self.m - j

    def last_block(self):
        """View of the block submitted last (call after drain(): on the root it then holds the reduced sum)."""
        if getattr(self, "_last", None) is None:
            return None
        k, j = self._last
        return self.buffers[k][j * self.n_samples : (j + 1) * self.n_samples]

    def drain(self):
        self.close()  # every rank has rendered the same number of blocks: a partly filled super-block still owes its reduce
        for i in range(self.n_buffers):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
