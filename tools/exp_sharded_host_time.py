"""(box) Host issue time of the in-library sharded handle: python tools/exp_sharded_host_time.py [shards] [voices] [blocks per call]
Every shard's launch sequence is issued by a thread of its own (shard 0 by the caller's; PHONIC_SHARD_THREADS=0: one thread issues all). Here: N
shards on ONE device (a 1-GPU lease) — the launch sequence per shard is what it would be across devices, but the threads share one device's
runtime locks and queues, and the peer copies are local. Prints the host's enqueue time per call next to
the call's total time and the plain graph's figures with the same voices."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from phonic_amd import workloads  # noqa: E402
from phonic_amd.graph import Graph, ShardedGraph  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = 1024


def run(g, name, sharded):
    g.set_max_blocks_per_launch(max(K, 1))
    workloads.build_headline(g, V, 0, V, 2.0)
    bus = torch.zeros(K * 2 * N, device="cuda:0")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    pos = 0

    def write():
        nonlocal pos
        if sharded:
            w = g.write_device(bus.data_ptr(), K * 2 * N, pos)
        else:
            w = g.write_device(bus.data_ptr(), K * 2 * N, pos, stream.cuda_stream)
        assert w == K * 2 * N
        pos += K * N

    for _ in range(30):
        write()
    g.synchronize()
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        calls = 200
        t0 = time.perf_counter()
        for _ in range(calls):
            write()
        t1 = time.perf_counter()
        g.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = {"what": name, "shards": S if sharded else 1, "voices": V, "blocks_per_call": K, "host_enqueue_us_per_call": 1e6 * (t1 - t0) / calls, "total_us_per_call": 1e6 * (t2 - t0) / calls}
        r["host_share"] = r["host_enqueue_us_per_call"] / r["total_us_per_call"]
        if best is None or r["total_us_per_call"] < best["total_us_per_call"]:
            best = r
    print(json.dumps(best), flush=True)


run(Graph(48000, 2, N, 0), "plain graph, caller's stream", False)
threads = os.environ.get("PHONIC_SHARD_THREADS", "1") != "0"
run(ShardedGraph([0] * S, 48000, 2, N), f"pg_sharded, {S} shards on one device, " + ("one issuing thread per shard" if threads else "one host thread issues every shard (PHONIC_SHARD_THREADS=0)"), True)
