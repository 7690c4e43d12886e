"""Full-size checks of the headline configuration (1024 voices, BASELINE.json): against the oracle on all host cores (the whole bus,
`test_headline_and_c5_full_size_bus_against_the_oracle`) and through size-independent properties:
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


def _f32_sum_in_mixer_order(rows):
    """The mixer sum's fixed f32 order (pg_mix_kernel / pg_mix_kernel_1+2): partial sums over groups of 16 units in unit order,
    then the groups in order, every add rounded to f32."""
    n = rows.shape[0]
    total = np.zeros(rows.shape[1], np.float32)
    for g0 in range(0, n, 16):
        acc = np.zeros(rows.shape[1], np.float32)
        for u in range(g0, min(g0 + 16, n)):
            acc = (acc + rows[u]).astype(np.float32)
        total = (total + acc).astype(np.float32)
    return total


@pytest.mark.parametrize("n_units", [4096, 4097, 8192])
def test_mixer_sum_large_unit_counts_bit_exact(n_units):
    """The mixer-graph sum over more than 4096 units takes the two-launch path (pg_mix_kernel_1 / pg_mix_kernel_2, partials through
    HBM), up to 4096 the one-launch kernel; both must produce the documented f32 sum order bit for bit. Units = main-mixer sources
    that play a constant (48 kHz stereo, unit gain, centre pan: resampler bypass, gain and pan skipped — the row IS the PCM value),
    so the expected bus is a pure function of the sum order. Equal start times are inserted BEFORE existing sources
    (mixed.rs:324-329): unit order = reverse order of addition."""
    rng = np.random.default_rng(n_units)
    vals = (rng.standard_normal((n_units, 2)) * 1e-3).astype(np.float32)
    frames = 600
    g = gpu_graph(512)
    for i in range(n_units):
        pcm = np.tile(vals[i], frames).astype(np.float32)
        g.add_voice(0, np.concatenate([pcm, np.zeros(2, np.float32)]), 2, SR, volume=1.0, panning=0.0, fade_out_seconds=-1.0)
    out = np.zeros(1024, np.float32)
    assert g.write(out, 0) == 1024
    expect = _f32_sum_in_mixer_order(vals[::-1])
    assert np.array_equal(out.reshape(-1, 2), np.tile(expect, (512, 1)))
    ref = vals.astype(np.float64).sum(axis=0)
    assert np.abs(out.reshape(-1, 2)[0] - ref).max() < 1e-6


def test_c5_8192_voices_full_size():
    """BASELINE config 5 at its stated size on ONE GPU (8192 voices x Filter -> Eq5 -> Delay -> Reverb, ~43 GB of delay-line state;
    the N = 1 anchor of the strong-scaling claim): determinism (two builds, bit-identical), linearity of the mixer over
    the 8 shards of 1024 voices the 8-GPU run uses, and oracle spot checks of single voices. 24 blocks = 24 576 frames: the Delay's
    first echo (18 000 frames) is inside the run. Also the first exercise of the > 4096-unit mixer path on sub-mixer units."""
    V5, blocks = 8192, 24
    g = gpu_graph()
    workloads.build_c5(g, V5, 0, V5, seconds=0.25)
    full = g.render(blocks, 1024)
    assert np.isfinite(full).all() and np.abs(full).max() > 1e-2
    g.close()
    g2 = gpu_graph()
    workloads.build_c5(g2, V5, 0, V5, seconds=0.25)
    again = g2.render(blocks, 1024)
    g2.close()
    assert np.array_equal(again, full)
    acc = np.zeros_like(full, dtype=np.float64)
    for s in range(8):
        gs = gpu_graph()
        workloads.build_c5(gs, 1024, s * 1024, V5, seconds=0.25)
        acc += gs.render(blocks, 1024)
        gs.close()
    assert rms(acc - full) <= 2e-6 and np.abs(acc - full).max() <= 2e-5
    level = workloads.voice_level(V5)
    for i in (0, 4095, 8191):
        gg, gc = gpu_graph(), oracle.OracleGraph(SR, 2, 1024)
        for h in (gg, gc):
            workloads.build_c5(h, 1, i, V5, seconds=0.25)
        a, b = gg.render(blocks, 1024), gc.render(blocks, 1024)
        assert rms(a - b) <= 1e-5 * level * 4, (i, rms(a - b))  # a single voice sits at `level` of the bus scale
        assert np.abs(a[2 * 18000:] - a[:2 * 6576]).max() > 0  # (sanity: not a constant)


def oracle_bus_parallel(build, total_voices, blocks, slice_voices=1024):
    """The master bus of `total_voices` independent voices from the oracle on all host cores: voices are independent until the sum (the
    reference's own parallel axis, thread_pool.rs), so every core renders a small oracle graph (po_graphs_render_parallel) and the buses are
    added in f64. Built and rendered in slices of `slice_voices` so that the oracle never holds more than ~6 GB of delay lines."""
    import ctypes as C
    import os

    lib = oracle.lib()
    threads = max(1, os.cpu_count() or 1)
    total = np.zeros(blocks * 2048, np.float64)
    for s0 in range(0, total_voices, slice_voices):
        n_slice = min(slice_voices, total_voices - s0)
        n_graphs = min(threads, n_slice)
        per = (n_slice + n_graphs - 1) // n_graphs
        graphs = []
        for gi in range(n_graphs):
            first, cnt = s0 + gi * per, max(0, min(per, n_slice - gi * per))
            if cnt == 0:
                break
            h = oracle.OracleGraph(SR, 2, 1024)
            build(h, cnt, first)
            graphs.append(h)
        handles = (C.c_void_p * len(graphs))(*[h._h for h in graphs])
        outs = np.zeros(len(graphs) * blocks * 2048, np.float32)
        assert lib.po_graphs_render_parallel(handles, len(graphs), threads, outs.ctypes.data_as(C.POINTER(C.c_float)), 2048, blocks, 0) == 0
        total += outs.reshape(len(graphs), -1).astype(np.float64).sum(axis=0)
        for h in graphs:
            h.close()
    return total


@pytest.mark.parametrize("name,voices", [("headline", 1024), ("c5", 8192)])
def test_headline_and_c5_full_size_bus_against_the_oracle(name, voices):
    """The WHOLE master bus at the stated sizes — H: 1024 voices -> cubic -> gain / pan -> per-voice Reverb; C5: 8192 voices -> Filter -> Eq5 ->
    Delay -> Reverb — 24 blocks (the Delay's first echo at 18 000 frames is inside), against the sum of the oracle's per-voice renders
    (f64 sum of per-graph buses, all host cores): <= 1e-5 RMS, <= 1e-4 max-abs on a bus that peaks well above 1e-2. The GPU renders it as the
    bench does (super-block writes)."""
    blocks = 24
    build = (lambda h, n, first: workloads.build_headline(h, n, first, voices, seconds=0.25)) if name == "headline" else \
            (lambda h, n, first: workloads.build_c5(h, n, first, voices, seconds=0.25))
    g = gpu_graph()
    g.set_max_blocks_per_launch(8)
    build(g, voices, 0)
    a = np.zeros(blocks * 2048, np.float32)
    assert g.write(a[:2 * 2048], 0) == 2 * 2048                       # two single blocks' worth first: the units reach the steady state
    for c in range(2, blocks, 8):
        n = min(8, blocks - c)
        assert g.write(a[c * 2048:(c + n) * 2048], c * 1024) == n * 2048
    assert g.device_errors() == 0
    g.close()
    b = oracle_bus_parallel(build, voices, blocks)
    d = a.astype(np.float64) - b
    assert np.abs(b).max() > 1e-2
    assert rms(d) <= 1e-5 and np.abs(d).max() <= 1e-4, (rms(d), np.abs(d).max())


@pytest.mark.parametrize("name,voices,blocks", [("c2", 64, 45), ("c3", 1024, 24), ("c4", 256, 24)])
def test_baseline_configs_full_size_against_oracle(name, voices, blocks):
    """BASELINE configs 2, 3 and 4 at their stated sizes. Their oracle renders in seconds (no per-voice reverb), so the full graph is
    compared directly — <= 1e-5 RMS, <= 1e-4 max — plus determinism and, for C3 (independent per-voice chains), shard linearity."""
    def build(h, n=voices, first=0):
        if name == "c2":
            workloads.build_c2(h, n, seconds=0.25)
        elif name == "c3":
            workloads.build_c3(h, n, first, voices, seconds=0.25)
        else:
            workloads.build_c4(h, n, seconds=0.25)
    g, gc = gpu_graph(), oracle.OracleGraph(SR, 2, 1024)
    build(g)
    build(gc)
    a, b = g.render(blocks, 1024), gc.render(blocks, 1024)
    d = a.astype(np.float64) - b.astype(np.float64)
    assert rms(d) <= 1e-5 and np.abs(d).max() <= 1e-4, (rms(d), np.abs(d).max())
    assert np.abs(a).max() > 1e-2
    g2 = gpu_graph()
    build(g2)
    assert np.array_equal(g2.render(blocks, 1024), a)
    if name == "c3":
        acc = np.zeros_like(a, dtype=np.float64)
        for s in range(4):
            gs = gpu_graph()
            build(gs, voices // 4, s * (voices // 4))
            acc += gs.render(blocks, 1024)
        assert rms(acc - a) <= 1e-6 and np.abs(acc - a).max() <= 1e-5


def _run_bench(args, env_extra, launcher=None, timeout=900):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if "--strong-c5-voices" not in args:
        args = args + ["--strong-c5-voices", "0"]     # (the BASELINE config 5 leg of the default line: its own test below)
    cmd = (launcher or [sys.executable]) + [os.path.join(root, "bench.py")] + args
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_plain_invocation_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts one child per rank before it touches the GPU and forwards rank
    0's JSON line (PHONIC_BENCH_SHARED_GPU=1: both ranks on GPU 0 over gloo, RCCL refuses two ranks on one device). Weak scaling by
    default; `--scaling strong --total-voices` splits a fixed voice count over the ranks."""
    d = _run_bench(["--gpus", "2", "--steps", "8", "--warmup", "2", "--repeats", "2", "--voices", "48"], {"PHONIC_BENCH_SHARED_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["total_voices"] == 96 and d["config"]["voices_per_gpu"] == 48
    assert d["config"]["backend"] == "gloo" and d["value"] > 0 and d["config"]["bus_peak"] > 0.01 and d["repeats"]["n"] == 2
    d = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "2", "--workload", "c5", "--scaling", "strong", "--total-voices", "25"],
                   {"PHONIC_BENCH_SHARED_GPU": "1"})
    assert d["scaling"] == "strong" and d["config"]["total_voices"] == 25 and d["config"]["voices_per_gpu"] == 13 and d["value"] > 0
    assert abs(d["value"] - 25 * 1024 * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-6


def test_bench_rccl_branch_with_a_one_rank_group():
    """With 1-GPU leases the `nccl` (= RCCL) branch of bench.py cannot meet a second rank; PHONIC_BENCH_FORCE_DIST=1 makes the single rank
    build its RCCL process group anyway and push every super-block through an asynchronous dist.reduce on RCCL's stream (ordered behind the
    render stream, waited for before the buffer is reused): group creation, the collective's launch and the stream ordering run on the real
    backend. The bus must come back unchanged (sum over one rank)."""
    d = _run_bench(["--steps", "24", "--warmup", "8", "--repeats", "2", "--voices", "64", "--superblock", "8", "--no-cpu-baseline"], {"PHONIC_BENCH_FORCE_DIST": "1"})
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1 and d["config"]["rccl_group_forced"] is True
    assert d["value"] > 0 and d["config"]["bus_peak"] > 0.01
    ref = _run_bench(["--steps", "24", "--warmup", "8", "--repeats", "2", "--voices", "64", "--superblock", "8", "--no-cpu-baseline"], {})
    assert abs(ref["config"]["bus_peak"] - d["config"]["bus_peak"]) < 1e-7


def test_bench_single_gpu_line_shape():
    """The default invocation's JSON contract at a reduced size: roofline (kernel-timed), cpu_baseline with its flags, repeats."""
    d = _run_bench(["--steps", "20", "--warmup", "4", "--repeats", "3", "--voices", "64"], {})
    assert d["n_gpus"] == 1 and d["unit"] == "voice-frames/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["launches"] > 0 and r["kernel_ms"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and "-O" in c["flags"] and c["value"] > 0
    assert d["repeats"]["ms_per_step_min"] <= d["ms_per_step"] <= d["repeats"]["ms_per_step_max"]
    # the real-time call pattern (one write call per block) is timed in the same run on the same graph
    rt = d["config"]["realtime"]
    assert rt["blocks_per_call"] == 1 and rt["ms_per_step"] > 0 and 0 < rt["roofline_frac"] < 1 and rt["repeats"] == 3
    assert d["config"]["blocks_per_call_requested"] == 32 and d["config"]["blocks_per_call"] == 20      # (as issued: a 20-step leg is one call of 20 blocks)
    assert c["single_thread"]["cores"] == 1 and 0 < c["single_thread"]["value"] <= c["value"] * 1.5
    assert "strong_c5" not in d["config"]


def test_bench_line_carries_the_strong_scaling_leg_of_config_5():
    """config.strong_c5 of the default line: BASELINE config 5 (per-voice Filter -> Eq5 -> Delay -> Reverb) with a FIXED voice count split over the
    ranks, in the same run as the headline — what the driver's 1 / 2 / 4 / 8-GPU runs divide to get the >= 6x claim. Reduced voice
    count here; one rank, then two ranks sharing GPU 0 over gloo (the leg's graph is built after the headline's has been released)."""
    d = _run_bench(["--steps", "12", "--warmup", "4", "--repeats", "2", "--voices", "32", "--no-cpu-baseline", "--no-realtime", "--strong-c5-voices", "25"], {})
    s5 = d["config"]["strong_c5"]
    assert s5["total_voices"] == 25 and s5["voices_per_gpu"] == 25 and s5["scaling"] == "strong" and s5["value"] > 0 and s5["bus_peak"] > 1e-3
    assert abs(s5["value"] - 25 * 1024 * 12 / (s5["ms_per_step"] * 12e-3)) / s5["value"] < 1e-6
    assert d["config"]["total_voices"] == 32 and d["scaling"] == "weak"
    d2 = _run_bench(["--gpus", "2", "--steps", "12", "--warmup", "4", "--repeats", "2", "--voices", "32", "--strong-c5-voices", "25"], {"PHONIC_BENCH_SHARED_GPU": "1"})
    s5 = d2["config"]["strong_c5"]
    assert d2["n_gpus"] == 2 and s5["total_voices"] == 25 and s5["voices_per_gpu"] == 13 and s5["n_gpus"] == 2 and s5["bus_peak"] > 1e-3


def test_bench_legs_cover_the_requested_time_and_bus_workloads_use_superblocks():
    """Without --repeats the legs are repeated until --min-seconds of wall time are covered (a 2 ms timed region is thin);
    the workloads with a bus chain (C2, C4) now render super-blocks like the others (one mix launch over all blocks + one bus launch that walks
    them), also with the bus deferred behind a two-rank reduce."""
    d = _run_bench(["--steps", "20", "--warmup", "4", "--voices", "64", "--min-seconds", "0.2", "--no-cpu-baseline"], {})
    assert d["repeats"]["timed_seconds"] >= 0.19 and d["repeats"]["n"] >= 5, d["repeats"]
    assert d["config"]["realtime"]["timed_seconds"] >= 0.09, d["config"]["realtime"]
    for wl in ("c2", "c4"):
        d = _run_bench(["--steps", "32", "--warmup", "8", "--repeats", "3", "--workload", wl, "--superblock", "16", "--no-cpu-baseline", "--no-realtime"], {})
        assert d["config"]["blocks_per_call"] == 16 and d["roofline"]["blocks_per_launch"] > 4 and d["config"]["bus_peak"] > 0.005, (wl, d["config"], d["roofline"])
        one = _run_bench(["--steps", "32", "--warmup", "8", "--repeats", "3", "--workload", wl, "--superblock", "1", "--no-cpu-baseline"], {})
        assert abs(one["config"]["bus_peak"] - d["config"]["bus_peak"]) < 1e-6, (wl, one["config"]["bus_peak"], d["config"]["bus_peak"])          # the same audio either way (same number of blocks rendered: no real-time legs)
    d = _run_bench(["--gpus", "2", "--steps", "16", "--warmup", "4", "--repeats", "2", "--workload", "c4", "--superblock", "4", "--voices", "32"], {"PHONIC_BENCH_SHARED_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["blocks_per_call"] == 4 and d["config"]["bus_peak"] > 0.005, d["config"]


def test_bench_fails_loudly_when_the_rccl_group_cannot_be_created():
    """No silent fallback to another backend: two ranks told to use the SAME device cannot form an RCCL group (RCCL refuses a duplicate GPU);
    every rank must end with a non-zero exit code and ONE line naming the rank and RCCL's error."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("PHONIC_BENCH_SHARED_GPU", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--voices", "8", "--strong-c5-voices", "0"], cwd=root, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate())
    assert all(p.returncode not in (0, None) for p in procs), [p.returncode for p in procs]
    assert any("RCCL group creation failed" in e and "rank" in e for (_, e) in outs), [e[-600:] for (_, e) in outs]
    assert not any(o.strip().startswith("{") for (o, _) in outs)                       # and no result line


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
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--voices", "64", "--reduce-every", str(reduce_every),
           "--superblock", str(min(reduce_every, 4)), "--repeats", "2", "--strong-c5-voices", "0"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["config"]["total_voices"] == 128 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["bus_peak"] > 0.01 and "cpu_baseline" not in d
