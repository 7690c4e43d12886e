#!/usr/bin/env python3
"""bench.py — throughput of the phonic DSP hot path on MI355X.

A "step" is one 1024-frame block (the reference's offline block size, src/output/wav.rs:25) of the whole
graph: every voice is resampled, run through its effect chain and summed into the master bus. Default
workload = the configuration BASELINE.json's target is quoted on: "headline" = 1024 stereo 44.1 kHz voices
-> cubic resampler -> gain/pan -> per-voice Reverb -> mixer sum (SURVEY.md §8d "H"), per GPU.

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU: voices are sharded over ranks (weak scaling: --voices per GPU), each rank renders its partial master
bus on its GPU, and the partial buses meet in one RCCL sum-reduce to rank 0 per block (the reference's caller-side
sum of worker outputs, src/source/mixed.rs:522-536).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel launch(es) of the graph, named in `roofline.kernel`, hipEvent-timed
on the launching stream) and `cpu_baseline` (the CPU oracle — a C++ port of the reference path, NOT the Rust binary — on this
box's host cores, bounded sample of about 12 s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

# algorithmic bytes per voice-frame (derivation in DESIGN.md)
B_ALG = {"headline": 423.4, "c2": 14.6, "c3": 36.0, "c4": 7.5, "c5": 456.0}
DEFAULT_VOICES = {"headline": 1024, "c2": 64, "c3": 1024, "c4": 256, "c5": 1024}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_workload(g, name, n_voices, first_voice, total_voices, seconds):
    import workloads

    if name == "headline":
        workloads.build_headline(g, n_voices, first_voice, total_voices, seconds)
    elif name == "c2":
        workloads.build_c2(g, n_voices, seconds)
    elif name == "c3":
        workloads.build_c3(g, n_voices, first_voice, total_voices, seconds)
    elif name == "c4":
        workloads.build_c4(g, n_voices, seconds)
    elif name == "c5":
        workloads.build_c5(g, n_voices, first_voice, total_voices, seconds)
    else:
        raise ValueError(name)


def pmc_traffic(name, v_per_gpu, block):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 FETCH correction applied), if one matches this configuration.
    bench.py cannot run the profiler on itself; the file names the command that produced it."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == name and d.get("voices_per_gpu") == v_per_gpu and d.get("block_frames") == block:
            best = d
    return best["traffic_bytes_per_launch"] if best else None


def cpu_baseline(name, block, seconds_budget=12.0):
    """Times the CPU oracle on a bounded sample of the same workload with all host cores (one graph per core). A short parallel
    run calibrates the block count so that the timed sample takes about `seconds_budget` seconds on this host."""
    import ctypes as C

    import oracle

    cores = os.cpu_count() or 1
    threads = cores
    lib = oracle.lib()
    per_graph = {"headline": 4, "c2": 16, "c3": 8, "c4": 32, "c5": 2}[name]
    n_graphs = threads

    def run(n_blocks):
        graphs = [oracle.OracleGraph(48000, 2, block) for _ in range(n_graphs)]
        for i, g in enumerate(graphs):
            build_workload(g, name, per_graph, i * per_graph, per_graph * n_graphs, 0.5)
        handles = (C.c_void_p * n_graphs)(*[g._h for g in graphs])
        outs = np.zeros(n_graphs * n_blocks * block * 2, np.float32)
        t0 = time.perf_counter()
        lib.po_graphs_render_parallel(handles, n_graphs, threads, outs.ctypes.data_as(C.POINTER(C.c_float)), block * 2, n_blocks, 0)
        return time.perf_counter() - t0

    cal_blocks = 16
    t_cal = run(cal_blocks)
    n_blocks = int(max(cal_blocks, min(4000, seconds_budget * cal_blocks / max(t_cal, 1e-6))))
    dt = run(n_blocks)
    vf = n_graphs * per_graph * n_blocks * block
    return {
        "value": vf / dt,
        "unit": "voice-frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_graphs} oracle graphs x {per_graph} voices x {n_blocks} blocks of {block} frames, {threads} threads, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=sorted(B_ALG))
    ap.add_argument("--voices", type=int, default=0, help="voices PER GPU (default: the config's count)")
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact", action="store_true", help="disable the time-parallel paths (exact serial evaluation)")
    ap.add_argument("--time-every", type=int, default=4, help="hipEvent-time the dominant kernel every n-th step (the event pair costs ~8 us per step)")
    ap.add_argument("--reduce-every", type=int, default=16, help="multi-GPU: blocks per RCCL master-bus reduce (offline super-block; 1 = per block, the real-time setting)")
    ap.add_argument("--staged", type=int, default=1, help="reverb sub-mixers: 1 = staged kernel (default), 2 = one launch per stage, 0 = fused fast kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # PHONIC_BENCH_SHARED_GPU=1 (test hook, 1-GPU boxes): every rank renders on GPU 0 and the ranks meet over gloo — exercises the
    # multi-rank control flow (sharding, buffer ring, async reduce, timing) where RCCL cannot run (it refuses two ranks on one GPU)
    shared_gpu = os.environ.get("PHONIC_BENCH_SHARED_GPU", "") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from phonic_amd.graph import Graph
    from phonic_amd.parallel import MasterBusRing, reduce_master_bus

    name = args.workload
    v_per_gpu = args.voices or DEFAULT_VOICES[name]
    total_voices = v_per_gpu * world
    block = args.block
    g = Graph(48000, 2, block, local_rank)
    if args.exact:
        g.set_fast_math(0)
    g.set_staged(args.staged)
    g.set_timing_period(args.time_every)
    bus_on_root = name in ("c2", "c4")  # bus effects run once on the root after the reduce
    if world > 1 and bus_on_root:
        g.set_defer_bus(True)
    build_workload(g, name, v_per_gpu, rank * v_per_gpu, total_voices, 2.0)

    n_samples = block * 2
    # Master-bus buffers: a ring of N_BUS super-blocks of M blocks each. Offline rendering (the reference's WavOutput pull loop,
    # src/output/wav.rs:210-250) has no deadline per block, so the partial buses of M consecutive blocks travel in ONE RCCL reduce
    # (M x 8 KiB; SURVEY §8e "per super-block"): the reduce of super-block s (RCCL's own stream, ordered after the renders by an
    # event) overlaps the renders of the following ones, and the render stream only waits when a buffer comes round again.
    # --reduce-every 1 is the real-time setting (one reduce per block). Bus effects on the root (c2 / c4) run per block.
    # The headline launch fills the chip exactly (1024 workgroups = 256 CUs x 4 resident): a block whose dispatch finds an RCCL
    # workgroup resident on some CU leaves one render workgroup waiting for a second round, so fewer, larger reduces also mean fewer
    # disturbed blocks.
    M = 1 if (world == 1 or bus_on_root) else max(1, args.reduce_every)
    # a real (non-default) stream: pg_graph_write_device is asynchronous only on a caller's stream — the default stream's handle is
    # NULL, which the ABI reads as "the graph's own stream, synchronous" (include/phonic_gpu.h). torch and RCCL ops order after it.
    render_stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.synchronize()
    torch.cuda.set_stream(render_stream)
    stream = render_stream.cuda_stream
    ring = MasterBusRing(n_samples, M, f"cuda:{local_rank}", n_buffers=4, root=0)
    if bus_on_root:
        ring.distributed = False  # c2 / c4: the reduce is issued per block below, in front of the root's bus effects
    pos = 0

    def step():
        nonlocal pos
        bus = ring.slot()
        w = g.write_device(bus.data_ptr(), n_samples, pos, stream)
        if w != n_samples:
            raise RuntimeError("graph write failed: " + str(w))
        if world > 1 and bus_on_root:
            reduce_master_bus(bus, root=0)
            if rank == 0:
                g.process_bus_device(bus.data_ptr(), n_samples, pos, stream)
        ring.submit()
        pos += block

    drain = ring.drain
    buses = ring.buffers

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    g.kernel_ms(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms, launches = g.kernel_ms(reset=True)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    peak = float(max(b.abs().max().item() for b in buses))

    if rank == 0:
        vf_total = total_voices * block * args.steps
        value = vf_total / dt
        vf_per_launch = v_per_gpu * block
        achieved = B_ALG[name] * vf_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "sample-frames/sec (48 kHz stereo) through mixer+FX+resample",
            "value": value,
            "unit": "voice-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": {"headline": "H: 1024 stereo 44.1k voices -> cubic resampler -> gain/pan -> per-voice Reverb -> mixer sum",
                             "c2": "C2: 64 stereo 48k voices, Eq5+Reverb on the bus", "c3": "C3: 1024 mono voices, per-voice Filter+Chorus",
                             "c4": "C4: 256 stereo 44.1k voices -> cubic -> bus limiter", "c5": "C5: per-voice Filter->Eq5->Delay->Reverb"}[name],
                "voices_per_gpu": v_per_gpu,
                "total_voices": total_voices,
                "block_frames": block,
                "sample_rate": 48000,
                "master_frames_per_s": value / total_voices,
                "x_realtime": value / total_voices / 48000.0,
                "sharding": f"voices/{world}" + (f" + RCCL reduce(sum) of the master bus per {M} block(s)" if world > 1 else ""),
                "exact_mode": bool(args.exact),
                "bus_peak": peak,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(name, v_per_gpu, block),
                "kernel": g.dominant_kernel(),
                "kernel_ms": kernel_ms,
                "launches": launches,
                "timed_every": args.time_every,
                "bytes_per_voice_frame": B_ALG[name],
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, block)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
