"""Full-size checks of the headline configuration (1024 voices, BASELINE.json) through size-independent properties:
the oracle cannot render 1024 reverb voices in seconds, so the full graph is checked by
  * linearity of the mixer: the sum of independently rendered voice shards equals the full render (f32 reassociation only);
  * determinism: two runs are bit-identical (deterministic tree sum, no atomics);
  * call-size invariance of the steady state: 4 x 1024 frames == 1 x 4096 frames (MixedSource chunking, mixed.rs:679-712);
  * oracle spot checks: voices are independent until the master sum, so single voices of the 1024 (first / middle / last)
    rendered alone on the GPU must match the oracle.
"""
import numpy as np
import pytest

import oracle
import workloads

pytestmark = pytest.mark.gpu
SR = 48000
V = 1024


def gpu_graph(max_frames=1024):
    from phonic_amd.graph import Graph

    return Graph(SR, 2, max_frames, 0)


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def test_full_size_linearity_determinism_and_spot_checks():
    blocks = 6
    g = gpu_graph()
    workloads.build_headline(g, V, 0, V, seconds=0.25)
    full = g.render(blocks, 1024)
    assert np.isfinite(full).all() and np.abs(full).max() > 1e-2
    # determinism
    g2 = gpu_graph()
    workloads.build_headline(g2, V, 0, V, seconds=0.25)
    assert np.array_equal(g2.render(blocks, 1024), full)
    # linearity over shards (the multi-GPU decomposition): 4 shards of 256 voices
    acc = np.zeros_like(full, dtype=np.float64)
    for s in range(4):
        gs = gpu_graph()
        workloads.build_headline(gs, 256, s * 256, V, seconds=0.25)
        acc += gs.render(blocks, 1024)
    assert rms(acc - full) <= 1e-6 and np.abs(acc - full).max() <= 1e-5
    # oracle spot checks on single voices of the full configuration
    for i in (0, 511, 1023):
        gg, gc = gpu_graph(), oracle.OracleGraph(SR, 2, 1024)
        for h in (gg, gc):
            workloads.build_headline(h, 1, i, V, seconds=0.25)
        a, b = gg.render(blocks, 1024), gc.render(blocks, 1024)
        assert rms(a - b) <= 1e-5 / np.sqrt(V) * 4, (i, rms(a - b))  # per-voice level is 1/sqrt(V) of the bus


def test_call_size_invariance_steady_state():
    a_g = gpu_graph(4096)
    b_g = gpu_graph(4096)
    for h in (a_g, b_g):
        workloads.build_headline(h, 64, 0, 64, seconds=0.25)
    a = a_g.render(2, 4096)
    b = b_g.render(8, 1024)
    assert rms(a - b) <= 1e-6


def test_long_run_stays_on_the_oracle():
    """32 s of audio (1500 blocks of 1024 frames) through 8 headline voices: the exact closed forms (vibrato phases, resampler
    schedule) and the blocked scans must not drift — the last 100 blocks are held to the same 1e-5 RMS as the first."""
    import oracle

    blocks = 1500
    g = gpu_graph()
    gc = oracle.OracleGraph(48000, 2, 1024)
    for x in (g, gc):
        workloads.build_headline(x, 8, 0, 8, seconds=0.5)
    a = g.render(blocks, 1024)
    b = gc.render(blocks, 1024)
    for sl in (slice(0, 100 * 2048), slice((blocks - 100) * 2048, blocks * 2048)):
        d = a[sl].astype(np.float64) - b[sl].astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-5
        assert float(np.abs(d).max()) <= 1e-4
    assert np.abs(a[-2048:]).max() > 1e-3


@pytest.mark.parametrize("steps,warmup,reduce_every", [(6, 2, 8), (13, 3, 4), (6, 2, 1)])
def test_bench_two_ranks_on_one_gpu_control_flow(steps, warmup, reduce_every):
    """bench.py's multi-rank path (voice sharding, ring of super-block bus buffers, asynchronous reduce per `reduce_every` blocks incl.
    the partly filled super-block at the end, barrier + max-over-ranks timing)
    with two ranks sharing GPU 0 over gloo (PHONIC_BENCH_SHARED_GPU=1; RCCL refuses two ranks on one GPU): one JSON line from rank 0,
    whole-job voice count, audible master bus."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PHONIC_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--voices", "64", "--reduce-every", str(reduce_every)]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["config"]["total_voices"] == 128 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["bus_peak"] > 0.01 and "cpu_baseline" not in d
