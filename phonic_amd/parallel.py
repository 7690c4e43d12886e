"""Multi-GPU sharding of the mixer graph: one process per GPU, voices (sub-mixers / sources) are partitioned over
ranks — the reference's own parallel axis is independent sub-mixers (src/source/mixed/submixer/thread_pool.rs:92-121) —
and the partial master buses meet in ONE sum-reduce per block (the caller-side sum of worker outputs,
src/source/mixed.rs:522-536). On ROCm torch.distributed's "nccl" backend is RCCL over xGMI; the same code runs on
gloo for the CPU tests. The per-block message is 8 bytes x frames (8 KiB at 1024 frames): latency bound."""
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
