#!/usr/bin/env python3
"""bench.py — throughput of the phonic DSP hot path on MI355X.

A "step" is one 1024-frame block (the reference's offline block size, src/output/wav.rs:25) of the whole
graph: every voice is resampled, run through its effect chain and summed into the master bus. Default
workload = the configuration BASELINE.json's target is quoted on: "headline" = 1024 stereo 44.1 kHz voices
-> cubic resampler -> gain/pan -> per-voice Reverb -> mixer sum (SURVEY.md §8d "H"), per GPU.

  python bench.py --gpus 1 --steps 50 --warmup 10
  python bench.py --gpus N ...                     (spawns one child process per GPU itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N --workload c5 --scaling strong --total-voices 8192     (BASELINE config 5)

Multi-GPU: voices are sharded over ranks (weak scaling: --voices per GPU; strong scaling: --total-voices split over the
ranks), each rank renders its partial master bus on its GPU, and the partial buses meet in one RCCL sum-reduce to rank 0 per
super-block (the reference's caller-side sum of worker outputs, src/source/mixed.rs:522-536).

The timed region is a series of legs of exactly `--steps` blocks, each bracketed by barrier + torch.cuda.synchronize() on both sides and
reduced with MAX over ranks, repeated until `--min-seconds` (0.5 s) of wall time are covered (`--repeats` fixes the count);
`ms_per_step` / `value` come from the MEDIAN leg and the spread is printed under `repeats`. At one GPU a second set of legs pulls the
same graph with ONE write call per block — the reference's real-time call pattern — and lands in `config.realtime`.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel launch(es) of the graph, named in `roofline.kernel`, hipEvent-timed
on the launching stream) and `cpu_baseline` (the CPU oracle — a C++ port of the reference path, NOT the Rust binary — on this
box's host cores, bounded sample of about 12 s).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per voice-frame (derivation in DESIGN.md)
B_ALG = {"headline": 423.4, "c2": 14.6, "c3": 36.0, "c4": 7.5, "c5": 456.0}
DEFAULT_VOICES = {"headline": 1024, "c2": 64, "c3": 1024, "c4": 256, "c5": 1024}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WORKLOAD_TEXT = {
    "headline": "H: 1024 stereo 44.1k voices -> cubic resampler -> gain/pan -> per-voice Reverb -> mixer sum",
    "c2": "C2: 64 stereo 48k voices, Eq5+Reverb on the bus",
    "c3": "C3: 1024 mono voices, per-voice Filter+Chorus",
    "c4": "C4: 256 stereo 44.1k voices -> cubic -> bus limiter",
    "c5": "C5: per-voice Filter->Eq5->Delay->Reverb",
}


def build_workload(g, name, n_voices, first_voice, total_voices, seconds):
    from phonic_amd import workloads

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
    """HBM bytes per 1024-frame block of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 FETCH correction applied). Only a file recorded with THIS build of the
    kernels counts: the file carries the hash of the library sources it was measured with (phonic_amd._capi.source_hash);
    bench.py cannot run the profiler on itself. Returns (bytes or None, note)."""
    import glob

    from phonic_amd import _capi

    have = _capi.source_hash()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == name and d.get("voices_per_gpu") == v_per_gpu and d.get("block_frames") == block:
            if d.get("source_hash") == have:
                return d["traffic_bytes_per_block"] if "traffic_bytes_per_block" in d else d["traffic_bytes_per_launch"], os.path.basename(f)
            stale = stale or os.path.basename(f)
    return None, (f"no PMC pass recorded for source hash {have}" + (f" (latest: {stale}, other build)" if stale else ""))


def effective_cores():
    """Host cores this process may actually use: the scheduler affinity, capped by the cgroup's CPU quota (a GPU box of the pool shows 256 CPUs
    and grants 16 of them: cpu.max = "1600000 100000") — the thread count of the all-core CPU baseline and the `cores` it reports."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(name, block, seconds_budget=12.0):
    """Times the CPU oracle on a bounded sample of the same workload: (a) with all host cores (one graph per core), about `seconds_budget`
    seconds, and (b) on ONE thread (one graph), about a third of that. A short run calibrates each block count on this host.
    The library timed here is oracle/_build/libphonic_oracle_native.so (-O3 -march=native, built on this host when g++ is there),
    else the portable -O2 build the parity tests use; the flags are reported."""
    import ctypes as C

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle

    cores = effective_cores()
    lib, flags = oracle.lib_native()
    per_graph = {"headline": 4, "c2": 16, "c3": 8, "c4": 32, "c5": 2}[name]

    def run(n_blocks, n_graphs, threads):
        graphs = [oracle.OracleGraph(48000, 2, block, library=lib) for _ in range(n_graphs)]
        for i, g in enumerate(graphs):
            build_workload(g, name, per_graph, i * per_graph, per_graph * n_graphs, 0.5)
        handles = (C.c_void_p * n_graphs)(*[g._h for g in graphs])
        outs = np.zeros(n_graphs * n_blocks * block * 2, np.float32)
        t0 = time.perf_counter()
        lib.po_graphs_render_parallel(handles, n_graphs, threads, outs.ctypes.data_as(C.POINTER(C.c_float)), block * 2, n_blocks, 0)
        return time.perf_counter() - t0

    def sample(n_graphs, threads, budget):
        cal_blocks = 16
        t_cal = run(cal_blocks, n_graphs, threads)
        n_blocks = int(max(cal_blocks, min(4000, budget * cal_blocks / max(t_cal, 1e-6))))
        dt = run(n_blocks, n_graphs, threads)
        return n_graphs * per_graph * n_blocks * block / dt, n_blocks, dt

    v_all, nb_all, dt_all = sample(cores, cores, seconds_budget)
    v_one, nb_one, dt_one = sample(1, 1, seconds_budget / 3.0)
    return {
        "value": v_all,
        "unit": "voice-frames/s",
        "cores": cores,
        "kind": "port",
        "flags": flags,
        "sample": f"{cores} oracle graphs x {per_graph} voices x {nb_all} blocks of {block} frames, {cores} threads, {dt_all:.1f} s",
        "single_thread": {"value": v_one, "cores": 1, "sample": f"1 oracle graph x {per_graph} voices x {nb_one} blocks of {block} frames, 1 thread, {dt_one:.1f} s"},
    }


class ClockSampler:
    """Clock / power / temperature of THIS rank's GPU while the timed legs run: a thread polls the card's sysfs hwmon files (sclk = freq1_input,
    mclk = freq2_input, socket power = power1_input, junction / memory temperature = temp2_input / temp3_input; readable by an ordinary user on
    the pool's boxes, tools/probe_clocks.sh) every `period` seconds — the files are plain reads of the driver's cached metrics table, no child
    process, no re-exec. Sampled while the GPU is busy: an idle card reads its deep-sleep clock (~100 MHz) and says nothing about the legs.
    The card is found by the device's PCI address (torch's device properties), so the other seven GPUs of the host — other tenants — are not
    what is read. Why it is here: boxes of the pool differ by up to 12 % on the dominant kernel and a single run spans 0.47-0.63 of the
    roofline; the per-leg fractions are reported next to the clocks they ran at (`roofline.by_sclk`)."""

    def __init__(self, local_rank, period=0.004):
        import glob
        import threading

        self.period, self.samples, self.note = period, [], ""
        self._stop = threading.Event()
        self._thread = None
        self.card = self.hwmon = None
        try:
            import torch

            pr = torch.cuda.get_device_properties(local_rank)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception as e:  # noqa: BLE001
            want = None
            self.note = f"no PCI address from torch ({type(e).__name__})"
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            cards.append(d)
            if want and os.path.basename(os.path.realpath(d)).lower().startswith(want):
                self.card = d
        if self.card is None and len(cards) == 1:
            self.card = cards[0]
        if self.card is None:
            self.note = self.note or f"no drm card at PCI address {want} ({len(cards)} amdgpu cards visible)"
            return
        hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*")))
        self.hwmon = hw[0] if hw else None
        if self.hwmon is None:
            self.note = "card has no hwmon directory"
        self.pci = os.path.basename(os.path.realpath(self.card))

    @staticmethod
    def _read(path, scale):
        try:
            return float(open(path).read().strip()) / scale
        except (OSError, ValueError):
            return None

    def read_metrics(self):
        """The driver's gpu_metrics table (sysfs binary, version 1.8 as the pool's MI355X boxes expose it: 3872 bytes, header {size, format 1,
        content 8}; offsets checked against a dump taken under load, tools/probe_metrics.py): accumulation_counter (1 ms ticks) and
        ppt_residency_acc — ticks during which the PACKAGE POWER TRACKING limit throttled the chip — the eight XCDs' current shader clocks,
        the memory clock, socket power, memory-controller activity. None for any other table version."""
        import struct

        try:
            b = open(os.path.join(self.card, "gpu_metrics"), "rb").read()
        except OSError:
            return None
        if len(b) < 340 or b[2] != 1 or b[3] != 8 or int.from_bytes(b[0:2], "little") != len(b):
            return None
        hot, mem, vr, power, gfx_act, umc_act = struct.unpack_from("<6H", b, 4)
        acc, prochot, ppt, sock_thm, vr_thm, hbm_thm = struct.unpack_from("<6I", b, 40)
        xcd = struct.unpack_from("<8H", b, 296)
        uclk = struct.unpack_from("<H", b, 336)[0]
        return {"acc": acc, "ppt": ppt, "socket_thm": sock_thm, "hbm_thm": hbm_thm, "prochot": prochot, "xcd_mhz": xcd, "uclk_mhz": uclk, "power_w": power, "umc_activity": umc_act,
                "temp_hotspot_c": hot, "temp_mem_c": mem}

    def read_once(self):
        h = self.hwmon
        return (time.perf_counter(), self._read(h + "/freq1_input", 1e6), self._read(h + "/freq2_input", 1e6), self._read(h + "/power1_input", 1e6),
                self._read(h + "/temp2_input", 1e3), self._read(h + "/temp3_input", 1e3), self.read_metrics())

    def start(self):
        import threading

        if self.hwmon is None or self._thread is not None:
            return

        def run():
            while not self._stop.is_set():
                self.samples.append(self.read_once())
                self._stop.wait(self.period)

        self._stop.clear()
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def dpm_state(self):
        """The DPM tables' current entries (pp_dpm_*: the line marked `*`) and the performance level — one read, for the record."""
        out = {}
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level"):
            try:
                txt = open(os.path.join(self.card, name)).read().strip().splitlines()
            except OSError:
                continue
            cur = [ln for ln in txt if ln.rstrip().endswith("*")]
            out[name] = (cur[0].rstrip(" *") if cur else txt[0]) if txt else None
        return out

    def ppt_share_at(self, t, half_window=0.05):
        """Share of the 1 ms ticks around perf_counter time t (+- half_window s) during which the power limit throttled the chip."""
        gm = [(smp[0], smp[6]) for smp in self.samples if smp[6] and abs(smp[0] - t) <= half_window]
        if len(gm) < 2:
            return None
        d_acc = (gm[-1][1]["acc"] - gm[0][1]["acc"]) & 0xFFFFFFFF
        return (((gm[-1][1]["ppt"] - gm[0][1]["ppt"]) & 0xFFFFFFFF) / d_acc) if d_acc else None

    def sclk_at(self, t):
        """sclk (MHz) of the sample nearest to perf_counter time t."""
        best = None
        for smp in self.samples:
            if smp[1] is not None and (best is None or abs(smp[0] - t) < abs(best[0] - t)):
                best = smp
        return best[1] if best else None

    @staticmethod
    def _dist(vals):
        v = sorted(x for x in vals if x is not None)
        if not v:
            return None
        q = lambda f: v[min(len(v) - 1, int(f * len(v)))]
        return {"min": v[0], "p10": q(0.10), "p50": q(0.50), "p90": q(0.90), "max": v[-1]}

    def summary(self, windows):
        """Distribution of the samples that fall inside the timed windows [(t0, t1), ...] (all samples when none does)."""
        if self.hwmon is None:
            return {"source": None, "note": self.note}
        inside = [smp for smp in self.samples if any(a <= smp[0] <= b for a, b in windows)] or self.samples
        gm = [smp[6] for smp in inside if smp[6]]
        metrics = None
        if len(gm) >= 2:
            d_acc = (gm[-1]["acc"] - gm[0]["acc"]) & 0xFFFFFFFF
            res = lambda k: (((gm[-1][k] - gm[0][k]) & 0xFFFFFFFF) / d_acc) if d_acc else None
            metrics = {"source": "gpu_metrics v1.8 (sysfs)", "ticks": d_acc,
                       "ppt_throttled_share": res("ppt"), "socket_thermal_throttled_share": res("socket_thm"), "hbm_thermal_throttled_share": res("hbm_thm"), "prochot_share": res("prochot"),
                       "xcd_sclk_mhz": {"min": min(min(m["xcd_mhz"]) for m in gm), "p50": self._dist(sum(m["xcd_mhz"]) / 8.0 for m in gm)["p50"], "max": max(max(m["xcd_mhz"]) for m in gm)},
                       "uclk_mhz": self._dist(m["uclk_mhz"] for m in gm), "umc_activity_pct": self._dist(m["umc_activity"] for m in gm), "temp_hotspot_c": self._dist(m["temp_hotspot_c"] for m in gm)}
        return {"gpu_metrics": metrics, "source": "sysfs hwmon, polled every %g ms by a thread of bench.py while the legs ran" % (self.period * 1e3), "pci": self.pci, "samples": len(inside),
                "sclk_mhz": self._dist(s[1] for s in inside), "mclk_mhz": self._dist(s[2] for s in inside), "socket_power_w": self._dist(s[3] for s in inside),
                "temp_junction_c": self._dist(s[4] for s in inside), "temp_memory_c": self._dist(s[5] for s in inside), "dpm": self.dpm_state()}


def dist_of(vals, scale=1.0):
    v = sorted(vals)
    if not v:
        return {}
    q = lambda f: v[min(len(v) - 1, int(f * len(v)))] * scale
    return {"p10": q(0.10), "p50": q(0.50), "p90": q(0.90)}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, exactly what torch.distributed.run sets) BEFORE this process imports torch or touches HIP — a process that has
    initialised the GPU must never be replaced or forked — forward rank 0's JSON line and fail if any rank fails."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile

    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")  # rank 0's stdout (a file, not a pipe: nobody has to drain it while the ranks are polled)
    deadline = time.time() + float(os.environ.get("PHONIC_BENCH_RANK_TIMEOUT", "900"))
    failed = None
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out0 if r == 0 else sys.stderr))
        # one deadline for all ranks; the first rank that fails ends the others at once (a rank that died during start-up would otherwise
        # leave rank 0 waiting in init_process_group / barrier until the backend's own timeout)
        while any(p.poll() is None for p in procs):
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if time.time() > deadline:
                failed = "ranks still running at the deadline"
                break
            time.sleep(0.05)
    finally:
        for p in procs:   # also on KeyboardInterrupt / SIGTERM of the parent: no GPU process is left behind
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                pass
    rcs = [p.returncode for p in procs]
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    if failed or any(rc != 0 for rc in rcs):
        raise SystemExit(f"bench.py: {failed or 'a rank failed'}; rank exit codes {rcs}")


def measure_dyn(args, local_rank, block, mf, sb):
    """bench.py --workload dyn: the headline's voices OFF the steady state — what the reference's real life looks like (events at sample times,
    src/source/mixed.rs:679-712,761-924; voices that end and start, :558-624; auto-bypassed effects and silent sub-mixers,
    src/source/mixed/effect.rs:56-145). One GPU. The same process first times the steady headline graph (all voices audible, no events) with the
    same call pattern; then `--dyn-seconds` of audio of the dynamic graph: per second of audio `--churn` % of the voices end and restart at
    random sample times, `--events` commands (reverb `wet`, source volume, source panning — equal shares) land at random sample times on random
    audible voices, `--silent` % of the voices have run past their tails. Reported: ms per step (offline calls of --superblock blocks, and one
    call per block), the ratio to the steady graph, the share of unit-blocks that left the time-parallel kernels and the generic kernel's time."""
    import numpy as np
    import torch

    from phonic_amd import workloads
    from phonic_amd.graph import Graph

    V = args.voices or 1024
    sr = 48000
    audio_s = args.dyn_seconds
    warm_s = 12.0 if args.silent > 0 else 1.0   # (silent voices: reverb tail 5.6 s + the sub-mixer's 2 s gate must have passed)
    warm_blocks = int(warm_s * sr / block) // sb * sb + sb
    steps = max(sb, int(audio_s * sr / block) // sb * sb)
    stream_t = torch.cuda.Stream(device=local_rank)
    torch.cuda.synchronize()
    torch.cuda.set_stream(stream_t)
    stream = stream_t.cuda_stream
    out_buf = torch.zeros(sb * block * 2, dtype=torch.float32, device=f"cuda:{local_rank}")

    def make(dynamic):
        g = Graph(sr, 2, mf, local_rank)
        g.set_timing_period(1)
        g.set_max_blocks_per_launch(max(1, min(64, sb * block // mf)))
        if dynamic:
            # the dynamic span is rendered twice (offline calls, then one call per block): restarts are planned over both spans
            plan = workloads.build_dyn(g, V, 2 * audio_s + 1.0, args.churn, args.silent, first_frame=warm_blocks * block, seed=args.dyn_seed)
        else:
            workloads.build_headline(g, V)
            plan = None
        return g, plan

    def run(g, drv, n_blocks, per_call, pos):
        """n_blocks blocks in calls of per_call; before every call the commands that fall into it are scheduled (workloads.DynDriver)."""
        n_cmds = 0
        done = 0
        while done < n_blocks:
            k = min(per_call, n_blocks - done)
            if drv:
                n_cmds += drv.schedule(g, pos, pos + k * block)
            w = g.write_device(out_buf.data_ptr(), k * block * 2, pos, stream)
            if w != k * block * 2:
                raise RuntimeError("graph write failed: " + str(w))
            pos += k * block
            done += k
        return pos, n_cmds

    def timed(g, drv, n_blocks, per_call, pos):
        torch.cuda.synchronize()
        g.kernel_stats(reset=True)
        g.dynamic_stats(reset=True)
        t0 = time.perf_counter()
        pos, n_cmds = run(g, drv, n_blocks, per_call, pos)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, launches, blocks = g.kernel_stats(reset=True)
        st = g.dynamic_stats(reset=True)
        return pos, dt, n_cmds, st, (ms, launches, blocks)

    res = {}
    for dynamic in (False, True):
        g, plan = make(dynamic)
        drv = workloads.DynDriver(plan, args.events, args.dyn_seed + 1, kinds=[int(c) for c in args.dyn_kinds]) if dynamic else None
        pos, _ = run(g, None, warm_blocks if dynamic else 2 * sb, sb, 0)   # warm-up: no commands
        legs = {}
        for tag, per_call in (("offline", sb), ("realtime", 1)):
            pos, dt, n_cmds, st, ks = timed(g, drv, steps, per_call, pos)
            legs[tag] = {"ms_per_step": dt / steps * 1e3, "commands": n_cmds, "commands_per_block": n_cmds / steps,
                         "unit_blocks": st["unit_blocks"], "deferred_unit_blocks": st["deferred_unit_blocks"],
                         "deferred_share": st["deferred_unit_blocks"] / max(1, st["unit_blocks"]),
                         "generic_launches": st["generic_launches"], "generic_launches_with_work": st["generic_launches_with_work"],
                         "generic_ms_per_step": (st["generic_ms"] / max(1, st["generic_timed"])) * st["generic_launches"] / steps,
                         "generic_ms_per_launch": st["generic_ms"] / max(1, st["generic_timed"]),
                         "fast_kernel_ms_per_block": (ks[0] * ks[1] / ks[2]) if ks[2] else 0.0, "fast_blocks_per_launch": (ks[2] / ks[1]) if ks[1] else 0.0}
        err = g.device_errors()
        if err:
            raise RuntimeError(f"kernel consistency flags raised: {err}")
        peak = float(out_buf[: 2 * block].abs().max().item())
        res["dynamic" if dynamic else "steady"] = {"legs": legs, "bus_peak": peak}
        g.close()
    torch.cuda.set_stream(torch.cuda.default_stream(local_rank))
    d, s0 = res["dynamic"]["legs"], res["steady"]["legs"]
    line = {
        "metric": "sample-frames/sec (48 kHz stereo) through mixer+FX+resample", "value": V * block / (d["offline"]["ms_per_step"] * 1e-3), "unit": "voice-frames/s",
        "n_gpus": 1, "steps": steps, "warmup": warm_blocks, "ms_per_step": d["offline"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "dyn: the headline's voices off the steady state (events, voices that end and restart, auto-bypassed voices)", "voices_per_gpu": V,
                   "block_frames": block, "max_frames": mf, "blocks_per_call": sb, "churn_pct_per_s": args.churn, "events_per_s": args.events, "kinds": args.dyn_kinds, "silent_pct": args.silent,
                   "audio_seconds": steps * block / sr, "seed": args.dyn_seed},
        "dyn": {"offline": d["offline"], "realtime": d["realtime"], "steady_offline_ms_per_step": s0["offline"]["ms_per_step"], "steady_realtime_ms_per_step": s0["realtime"]["ms_per_step"],
                "ratio_offline": d["offline"]["ms_per_step"] / s0["offline"]["ms_per_step"], "ratio_realtime": d["realtime"]["ms_per_step"] / s0["realtime"]["ms_per_step"],
                "bus_peak": res["dynamic"]["bus_peak"], "steady_bus_peak": res["steady"]["bus_peak"]},
    }
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0, help="timed legs of exactly --steps blocks each, the median leg is reported (default: as many as --min-seconds needs, "
                    "at least 5 — a 20-step leg lasts 2 ms and single legs scatter by +-8 % with the clock state of the box)")
    ap.add_argument("--workload", default="headline", choices=sorted(B_ALG) + ["dyn"])
    ap.add_argument("--churn", type=float, default=0.0, help="--workload dyn: %% of the voices that end and restart per second of audio, at random sample times")
    ap.add_argument("--events", type=float, default=0.0, help="--workload dyn: parameter / volume / panning commands per second of audio, at random sample times")
    ap.add_argument("--silent", type=float, default=0.0, help="--workload dyn: %% of the voices that have run past their tails (auto-bypassed)")
    ap.add_argument("--dyn-seconds", type=float, default=10.0, help="--workload dyn: seconds of audio per timed span")
    ap.add_argument("--dyn-seed", type=int, default=1234)
    ap.add_argument("--dyn-kinds", default="012", help="--workload dyn: which commands --events draws from: 0 = the reverb's `wet`, 1 = source volume, 2 = source panning")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --voices per GPU; strong: --total-voices split over the GPUs")
    ap.add_argument("--voices", type=int, default=0, help="weak scaling: voices PER GPU (default: the config's count)")
    ap.add_argument("--total-voices", type=int, default=0, help="strong scaling: voices of the whole job (default: c5 8192, else the config's count)")
    ap.add_argument("--block", type=int, default=1024, help="frames per step (= per write call when --superblock 1: a host's callback size)")
    ap.add_argument("--max-frames", type=int, default=0, help="the graph's max_frames: the kernels' piece size (default: min(--block, 1024) — a write is walked in the "
                    "reference's <= 4096-frame chunks whatever this is; the staged kernels take pieces of <= 1024 frames)")
    ap.add_argument("--superblock", type=int, default=32, help="blocks rendered per pg_graph_write_device call (offline pull loop; 1 = one call per block, the real-time setting)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clocks", action="store_true", help="do not poll the card's sysfs clock / power / temperature files during the legs (config.clocks)")
    ap.add_argument("--no-realtime", action="store_true", help="skip the second set of legs (one write call per block) behind config.realtime")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="timed legs are repeated until this much wall time is covered (unless --repeats is given)")
    ap.add_argument("--exact", action="store_true", help="disable the time-parallel paths (exact serial evaluation)")
    ap.add_argument("--time-every", type=int, default=1, help="hipEvent-time the dominant kernel every n-th launch round (the event pair costs ~8 us of stream time: once per super-block by default)")
    ap.add_argument("--reduce-every", type=int, default=0, help="multi-GPU: blocks per RCCL master-bus reduce (default: the super-block; 1 = per block, the real-time setting)")
    ap.add_argument("--strong-c5-voices", type=int, default=8192, help="voices of the BASELINE config 5 leg reported under config.strong_c5 (split over the GPUs; 0 = skip it)")
    ap.add_argument("--staged", type=int, default=1, help="reverb sub-mixers: 1 = staged kernel (default), 2 = one launch per stage, 0 = fused fast kernel")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # PHONIC_BENCH_SHARED_GPU=1 (test hook, 1-GPU boxes): every rank renders on GPU 0 and the ranks meet over gloo — exercises the
    # multi-rank control flow (sharding, buffer ring, async reduce, timing) where RCCL cannot run (it refuses two ranks on one GPU)
    shared_gpu = os.environ.get("PHONIC_BENCH_SHARED_GPU", "") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    # PHONIC_BENCH_FORCE_DIST=1 (test hook, 1-GPU boxes): a single rank still builds its RCCL group and sends every super-block through
    # dist.reduce on RCCL's stream — the `nccl` branch (group creation, asynchronous reduce ordered behind the render stream, Work.wait)
    # runs on the real backend even where only one GPU can be leased.
    force_dist = world == 1 and os.environ.get("PHONIC_BENCH_FORCE_DIST", "") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if shared_gpu else "nccl"
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            # RCCL or nothing: a group that cannot be created (or cannot carry a first collective) ends the run with one line naming the
            # rank and RCCL's error and a non-zero exit code — there is no fallback to another backend
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device=f"cuda:{local_rank}")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"first all_reduce over {world} rank(s) returned {probe.item()}")
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"bench.py: rank {rank}/{world} (cuda:{local_rank}): RCCL group creation failed: {type(e).__name__}: {' '.join(str(e).split())[:600]}\n")
                sys.stderr.flush()
                os._exit(3)

    from phonic_amd.graph import Graph
    from phonic_amd.parallel import MasterBusRing, reduce_master_bus, shard_range

    block = args.block
    mf = args.max_frames or min(block, 1024)   # the kernels' piece size; a step (and a real-time call) is `block` frames
    sb = max(1, args.superblock)
    # a launch sequence renders at most 64 pieces of max_frames, and a deferred-bus call (several ranks, bus chain on the root) leaves at most
    # max(64, 4096 / max_frames) `audible` words, one per piece (include/phonic_gpu.h): calls are sized to stay within both
    pieces_per_block = (block + mf - 1) // mf
    sb = max(1, min(sb, 64 // pieces_per_block))
    dist_on = world > 1 or force_dist  # (a forced one-rank group goes through the same barriers and reductions)

    def measure(name, scaling, voices_arg, total_voices_arg, min_seconds, with_realtime, sampler=None):
        """Builds the workload's graph on this rank, renders warm-up and timed legs; returns what the result line is made of."""
        if scaling == "strong":
            total_voices = total_voices_arg or (8192 if name == "c5" else DEFAULT_VOICES[name])
            first_voice, v_per_gpu = shard_range(total_voices, rank, world)
        else:
            v_per_gpu = voices_arg or DEFAULT_VOICES[name]
            total_voices = v_per_gpu * world
            first_voice = rank * v_per_gpu
        bus_on_root = name in ("c2", "c4")  # bus effects: once, behind the sum (on the root after the reduce when there are several ranks)
        g = Graph(48000, 2, mf, local_rank)
        if args.exact:
            g.set_fast_math(0)
        g.set_staged(args.staged)
        g.set_timing_period(args.time_every)
        g.set_max_blocks_per_launch(max(1, min(64, sb * block // mf)))
        if world > 1 and bus_on_root:
            g.set_defer_bus(True)
        build_workload(g, name, v_per_gpu, first_voice, total_voices, 2.0)

        n_samples = block * 2
        # Master-bus buffers: a ring of N_BUS super-blocks of M blocks each. Offline rendering (the reference's WavOutput pull loop,
        # src/output/wav.rs:210-250) has no deadline per block, so M consecutive blocks are rendered by ONE pg_graph_write_device call
        # (the reference's MixedSource::write loops over its <= 4096-frame chunks the same way, src/source/mixed.rs:679-712) and their
        # partial buses travel in ONE RCCL reduce (M x 8 KiB; SURVEY §8e "per super-block"): the reduce of super-block s (RCCL's own
        # stream, ordered after the renders by an event) overlaps the renders of the following ones, and the render stream only waits
        # when a buffer comes round again. --superblock 1 is the real-time setting (one call and one reduce per block).
        M = sb if bus_on_root else max(sb, args.reduce_every or sb)  # (c2 / c4: reduce and bus chain once per call, see render)
        M = (M + sb - 1) // sb * sb  # a reduce covers whole super-blocks
        # a real (non-default) stream: pg_graph_write_device is asynchronous only on a caller's stream — the default stream's handle is
        # NULL, which the ABI reads as "the graph's own stream, synchronous" (include/phonic_gpu.h). torch and RCCL ops order after it.
        render_stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.synchronize()
        torch.cuda.set_stream(render_stream)
        stream = render_stream.cuda_stream
        words_max = max(1, sb * pieces_per_block)   # `audible` words of the largest call: one per piece of max_frames
        ring = MasterBusRing(n_samples, M, f"cuda:{local_rank}", n_buffers=4, root=0, force_distributed=force_dist, extra=words_max if bus_on_root else 0)
        if bus_on_root:
            ring.distributed = False  # c2 / c4: the reduce is issued per block below, in front of the root's bus effects
        pos = 0
        calls = []   # blocks per write call, as issued (reported under config.blocks_per_call)

        def render(n_blocks, per_call=None):
            """n_blocks consecutive blocks: super-blocks of `per_call` (default: --superblock; one ABI call each), the remainder in one smaller call."""
            nonlocal pos
            done = 0
            per_call = per_call or sb
            while done < n_blocks:
                left = n_blocks - done
                parts = (left + per_call - 1) // per_call          # equal super-blocks (100 blocks at 32 per call: 4 x 25, not 32 + 32 + 32 + 4):
                k = min((left + parts - 1) // parts, ring.m)       # every launch pays about one block time of ramp-up and drain
                if k > ring.room():
                    ring.close()                                    # (the ring's super-block ends where the call does)
                nw = k * pieces_per_block if (world > 1 and bus_on_root) else 0
                bus = ring.slots(k, extra=nw)
                calls.append(k)
                w = g.write_device(bus.data_ptr(), k * n_samples, pos, stream)
                if w != k * n_samples:
                    raise RuntimeError("graph write failed: " + str(w))
                if world > 1 and bus_on_root:
                    if g.audible_words() != nw:   # (no events in the bench: an event-free call leaves one word per piece)
                        raise RuntimeError(f"deferred-bus write left {g.audible_words()} words, expected {nw}")
                    # the ranks' `audible` words ride behind the samples: ONE sum-reduce carries both, and the root's chain bypasses itself over
                    # silence as the one main mixer does (EffectProcessor's decisions per chunk, src/source/mixed/effect.rs:56-145)
                    g.export_audible(bus.data_ptr() + 4 * k * n_samples, nw, stream)
                    reduce_master_bus(bus, root=0)
                    if rank == 0:
                        g.process_bus_device(bus.data_ptr(), k * n_samples, pos, stream, flags_ptr=bus.data_ptr() + 4 * k * n_samples, n_words=nw)
                ring.submit(k)
                pos += k * block
                done += k

        def leg(n_blocks, per_call=None):
            torch.cuda.synchronize()
            g.kernel_ms(reset=True)
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            render(n_blocks, per_call)
            ring.drain()
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dt = t1 - t0
            ms, launches, blocks = g.kernel_stats(reset=True)
            bus = g.bus_kernel_stats(reset=True)
            return dt, ms, launches, blocks, bus, t0, t1

        def legs_for(seconds, per_call=None):
            """Timed legs of exactly --steps blocks until `seconds` of wall time are covered (at least 5, at most 2001; --repeats overrides):
            a 20-step leg lasts 2 ms, single legs scatter with the clock state of the box. Returns (legs, per-leg seconds, MAX over ranks)."""
            first = leg(args.steps, per_call)
            d0 = first[0]
            if dist_on:
                t = torch.tensor([d0], dtype=torch.float64, device=f"cuda:{local_rank}")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)   # every rank runs the same number of legs
                d0 = float(t.item())
            n = args.repeats if args.repeats > 0 else max(5, min(2001, int(1.2 * seconds / max(d0, 1e-6)) | 1))   # (+20 %: the first leg runs slower than the rest)
            legs = [first] + [leg(args.steps, per_call) for _ in range(n - 1)]
            dts = [l[0] for l in legs]
            if dist_on:
                t = torch.tensor(dts, dtype=torch.float64, device=f"cuda:{local_rank}")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dts = [float(x) for x in t.tolist()]
            torch.cuda.synchronize()
            return legs, dts

        def roofline_of(legs):
            """Dominant kernel: algorithmic bytes of one launch / its average duration, per leg; a launch renders `blocks_per_launch` blocks of this
            rank's voices (super-block launches loop over the blocks inside the kernel). Returns (sorted GB/s, median leg's (ms, blocks per launch, GB/s, launches), [(leg mid time, GB/s)])."""
            per_leg = []
            for (_, ms, launches, blocks, _bus, lt0, lt1) in legs:
                if launches and ms > 0:
                    bpl = blocks / launches
                    per_leg.append((ms, bpl, B_ALG[name] * v_per_gpu * mf * bpl / (ms * 1e-3) / 1e9, launches, 0.5 * (lt0 + lt1)))   # (a launch renders bpl pieces of max_frames frames)
            if not per_leg:
                return [0.0], (0.0, 0.0, 0.0, 0), []
            by_time = [(p[4], p[2]) for p in per_leg]   # (leg mid time, GB/s): what the clock samples are matched against
            return sorted(p[2] for p in per_leg), sorted(per_leg, key=lambda p: p[2])[len(per_leg) // 2][:4], by_time

        if sampler:
            sampler.start()
        render(args.warmup)
        ring.drain()
        del calls[:]
        legs, dts = legs_for(min_seconds)
        offline_calls = sorted(calls)
        # the real-time call pattern — ONE write call per block, as the reference's WavOutput and cpal callbacks pull (src/output/wav.rs:210-250,
        # src/output/cpal.rs:700-723) — timed in the same run on the same graph: no super-block launches, every block its own launch sequence
        rt_legs, rt_dts = (legs_for(min_seconds / 2, 1) if (sb > 1 and world == 1 and with_realtime) else (None, None))
        # ... and what a host that WAITS for every callback's samples sees (cpal's data callback, src/output/cpal.rs:700-723: the samples are handed
        # over when the callback returns): one call, one stream synchronisation, per call — the call's latency, launch and wait included
        rt_sync = None
        if rt_legs is not None:
            lat = []
            for _ in range(max(50, min(400, args.steps * 4))):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                render(1, 1)
                render_stream.synchronize()
                lat.append(time.perf_counter() - t0)
            ring.drain()
            lat.sort()
            rt_sync = {"calls": len(lat), "ms_per_call_p50": lat[len(lat) // 2] * 1e3, "ms_per_call_p10": lat[len(lat) // 10] * 1e3, "ms_per_call_p90": lat[(9 * len(lat)) // 10] * 1e3}
        if sampler:
            sampler.stop()
        last = ring.last_block()
        peak = float(last.abs().max().item()) if last is not None else 0.0  # the last rendered block only (on the root: the sum over ranks)

        dev_err = g.device_errors()
        if dev_err:
            raise RuntimeError(f"kernel consistency flags raised: {dev_err}")
        res = dict(name=name, scaling=scaling, v_per_gpu=v_per_gpu, total_voices=total_voices, M=M, dts=dts, legs=legs, rt_dts=rt_dts, rt_legs=rt_legs, rt_sync=rt_sync, peak=peak,
                   offline_calls=offline_calls, kernel=g.dominant_kernel(), bus_kernel=g.bus_kernel(), roofline_of=roofline_of)
        torch.cuda.set_stream(torch.cuda.default_stream(local_rank))
        del ring
        g.close()
        return res

    if args.workload == "dyn":
        return measure_dyn(args, local_rank, block, mf, sb)
    sampler = ClockSampler(local_rank) if rank == 0 and not args.no_clocks else None
    R = measure(args.workload, args.scaling, args.voices, args.total_voices, args.min_seconds, not args.no_realtime, sampler)
    # BASELINE config 5 as north_star states it — 8192 voices with the full chain, voice-sharded over the GPUs of the node — rides along in
    # the same line (config.strong_c5): the driver's 1 / 2 / 4 / 8-GPU runs then measure the >= 6x claim itself, next to the
    # weak-scaling `value` of the headline. --strong-c5-voices 0 switches it off.
    S5 = None
    if args.strong_c5_voices > 0 and not (args.workload == "c5" and args.scaling == "strong"):
        S5 = measure("c5", "strong", 0, args.strong_c5_voices, min(args.min_seconds, 0.25), False)
    name, v_per_gpu, total_voices, M = R["name"], R["v_per_gpu"], R["total_voices"], R["M"]
    dts, legs, rt_dts, rt_legs, peak, offline_calls, roofline_of = R["dts"], R["legs"], R["rt_dts"], R["rt_legs"], R["peak"], R["offline_calls"], R["roofline_of"]
    if rank == 0:
        med = int(np.argsort(dts)[len(dts) // 2])
        dt = dts[med]
        value = total_voices * block * args.steps / dt
        ach, (ms_l, bpl_l, achieved, launches_l), legs_by_time = roofline_of(legs)
        traffic, traffic_note = pmc_traffic(name, v_per_gpu, mf)
        out = {
            "metric": "sample-frames/sec (48 kHz stereo) through mixer+FX+resample",
            "value": value,
            "unit": "voice-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "repeats": {"n": len(dts), "reported": "median", "ms_per_step_min": min(dts) / args.steps * 1e3, "ms_per_step_median": dt / args.steps * 1e3,
                        "ms_per_step_max": max(dts) / args.steps * 1e3, "timed_seconds": sum(dts)},
            "config": {
                "workload": WORKLOAD_TEXT[name],
                "voices_per_gpu": v_per_gpu,
                "total_voices": total_voices,
                "block_frames": block,
                "max_frames": mf,
                "blocks_per_call": offline_calls[len(offline_calls) // 2] if offline_calls else sb,   # as issued (the median call of the timed legs)
                "blocks_per_call_requested": sb,
                "sample_rate": 48000,
                "master_frames_per_s": value / total_voices,
                "x_realtime": value / total_voices / 48000.0,
                "sharding": f"voices/{world}" + (f" + {'RCCL' if backend == 'nccl' else backend} reduce(sum) of the master bus per {M} block(s)" if world > 1 else ""),
                "rccl_ranks": dist.get_world_size() if ((world > 1 or force_dist) and backend == "nccl") else (0 if world > 1 else 1),
                "rccl_group_forced": bool(force_dist),
                "backend": backend,
                "exact_mode": bool(args.exact),
                "bus_peak": peak,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_min": ach[0] / HBM_PEAK_GBS,
                "frac_max": ach[-1] / HBM_PEAK_GBS,
                **{"frac_" + k: v for k, v in dist_of(ach, 1.0 / HBM_PEAK_GBS).items()},   # the per-leg distribution: frac_p10 / p50 / p90
                "traffic": traffic * bpl_l if traffic else None,
                "traffic_note": traffic_note,
                "kernel": R["kernel"],
                "kernel_ms": ms_l,
                "blocks_per_launch": bpl_l,
                "kernel_ms_per_block": ms_l / bpl_l if bpl_l else 0.0,   # per piece of max_frames frames
                "kernel_ms_per_step": ms_l / bpl_l * block / mf if bpl_l else 0.0,
                "launches": launches_l,
                "timed_every": args.time_every,
                "bytes_per_voice_frame": B_ALG[name],
                "algorithmic_bytes_per_launch": B_ALG[name] * v_per_gpu * mf * bpl_l,
            },
        }
        if sampler:
            # the clocks the legs ran at, and the legs' fractions by the shader clock of the nearest sample (100 MHz bins)
            out["config"]["clocks"] = sampler.summary([(l[5], l[6]) for l in legs])
            bins = {}
            pairs = []
            for (t, gbs) in legs_by_time:
                f = sampler.sclk_at(t)
                if f is not None:
                    bins.setdefault(int(f // 100) * 100, []).append(gbs / HBM_PEAK_GBS)
                    pairs.append((f, gbs / HBM_PEAK_GBS))
            ppt_pairs = [(sampler.ppt_share_at(t), gbs / HBM_PEAK_GBS) for (t, gbs) in legs_by_time]
            ppt_pairs = [p for p in ppt_pairs if p[0] is not None]
            if len(ppt_pairs) > 2:
                lo = [f for (sh, f) in ppt_pairs if sh < 0.25]
                hi = [f for (sh, f) in ppt_pairs if sh >= 0.25]
                out["roofline"]["by_power_throttle"] = {"legs_below_25pct_throttled": {"legs": len(lo), **dist_of(lo)}, "legs_above": {"legs": len(hi), **dist_of(hi)}}
            if bins:
                out["roofline"]["by_sclk"] = {str(k): {"legs": len(v), **dist_of(v)} for k, v in sorted(bins.items())}
                if len(pairs) > 2:
                    xs, ys = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
                    out["roofline"]["corr_frac_sclk"] = float(np.corrcoef(xs, ys)[0, 1]) if xs.std() > 0 and ys.std() > 0 else 0.0
            if rt_legs:
                out["config"]["clocks"]["realtime_legs"] = {k: v for k, v in sampler.summary([(l[5], l[6]) for l in rt_legs]).items() if k in ("samples", "sclk_mhz", "socket_power_w", "temp_junction_c", "gpu_metrics")}
        # Which launch dominates by GPU time? Graphs whose work sits behind the sum (BASELINE configs 2 and 4) spend it in the main mixer's chain:
        # one workgroup per effect, a latency chain — the line then names that launch, bound "latency", with its time per block
        unit_ms = sum(l[1] * l[2] for l in legs)
        bus_ms = sum(l[4][0] * l[4][1] for l in legs)
        bus_blocks = sum(l[4][2] for l in legs)
        if bus_ms > unit_ms and bus_blocks:
            per_block_ms = bus_ms / bus_blocks
            achieved_bus = B_ALG[name] * v_per_gpu * mf / (per_block_ms * 1e-3) / 1e9
            out["roofline"].update({
                "bound": "latency", "kernel": R["bus_kernel"], "achieved": achieved_bus, "frac": achieved_bus / HBM_PEAK_GBS,
                "kernel_ms": bus_ms / max(1, sum(l[4][1] for l in legs)), "blocks_per_launch": bus_blocks / max(1, sum(l[4][1] for l in legs)),
                "kernel_ms_per_block": per_block_ms, "kernel_ms_per_step": per_block_ms * block / mf, "launches": sum(l[4][1] for l in legs),
                "latency_chain_us_per_block": per_block_ms * 1e3, "shader_cycles_per_block_at_2.4GHz": per_block_ms * 1e-3 * 2.4e9,
                "unit_kernels": {"kernel": R["kernel"], "ms_per_block": (unit_ms / max(1, sum(l[3] for l in legs))), "share_of_timed_gpu_ms": unit_ms / (unit_ms + bus_ms)},
                "note": "the launch that dominates by GPU time is the main mixer's effect chain behind the sum: a chain of per-frame recurrences on one workgroup per effect — "
                        "bounded by that workgroup's latency chain, not by HBM (frac = the workload's algorithmic bytes over THIS launch's time, for the record)"})
            for k in ("frac_min", "frac_max", "traffic", "algorithmic_bytes_per_launch"):
                out["roofline"].pop(k, None)
        if name == "c3" and out["roofline"]["bound"] == "hbm":
            # C3 (Filter -> Chorus per voice: 36 B per voice-frame) is ONE workgroup's latency chain per unit and block — ~64 K shader cycles of
            # dependent trips, barriers and two scans with every workgroup resident (stamps: profiles/r05_c3_stamps.txt) — not a byte stream:
            # the line says so, like the bus chains of C2 / C4
            out["roofline"].update({"bound": "latency", "latency_chain_us_per_block": out["roofline"]["kernel_ms_per_block"] * 1e3,
                                    "shader_cycles_per_block_at_2.4GHz": out["roofline"]["kernel_ms_per_block"] * 1e-3 * 2.4e9,
                                    "note": "every unit's block is one workgroup's chain of dependent steps (source, Filter scan, chorus phases, SVF scan, two tap / write rounds); all 1024 "
                                            "workgroups are resident at once, so the block takes one chain; frac = algorithmic bytes over that time, for the record"})
        if rt_legs:
            rt_dt = rt_dts[int(np.argsort(rt_dts)[len(rt_dts) // 2])]
            rt_ach, (rt_ms, rt_bpl, rt_achieved, _), _rt_bt = roofline_of(rt_legs)
            out["config"]["realtime"] = {
                "what": f"one pg_graph_write_device call per {block}-frame block (src/output/wav.rs:210-250, cpal), same graph, same run",
                "blocks_per_call": 1,
                "ms_per_step": rt_dt / args.steps * 1e3,
                "value": total_voices * block * args.steps / rt_dt,
                "roofline_frac": rt_achieved / HBM_PEAK_GBS,
                "roofline_frac_min": rt_ach[0] / HBM_PEAK_GBS,
                "roofline_frac_max": rt_ach[-1] / HBM_PEAK_GBS,
                **{"roofline_frac_" + k: v for k, v in dist_of(rt_ach, 1.0 / HBM_PEAK_GBS).items()},
                "kernel_ms_per_block": rt_ms / rt_bpl if rt_bpl else 0.0,
                "repeats": len(rt_dts),
                "timed_seconds": sum(rt_dts),
            }
            if R.get("rt_sync"):
                # (the figures above come from calls issued back to back without a host wait; this one is the latency of ONE call for a host that
                # waits for its samples: kernel + mixer sum + launch + the wait itself)
                out["config"]["realtime"]["synchronous"] = dict(R["rt_sync"], what="one call + one stream synchronisation per callback: the call's latency as a waiting host sees it")
        if S5:
            d5 = S5["dts"][int(np.argsort(S5["dts"])[len(S5["dts"]) // 2])]
            _, (ms5, bpl5, ach5, _l5), _bt5 = S5["roofline_of"](S5["legs"])
            out["config"]["strong_c5"] = {
                "what": "BASELINE config 5: %d voices, per-voice Filter->Eq5->Delay->Reverb, split over the GPUs (strong scaling), master bus reduced per %d block(s); "
                        "the 8-GPU value over the 1-GPU value of this field is the >= 6x claim" % (S5["total_voices"], S5["M"]),
                "total_voices": S5["total_voices"], "voices_per_gpu": S5["v_per_gpu"], "n_gpus": world, "scaling": "strong",
                "ms_per_step": d5 / args.steps * 1e3, "value": S5["total_voices"] * block * args.steps / d5, "unit": "voice-frames/s",
                "roofline_frac": ach5 / HBM_PEAK_GBS, "kernel": S5["kernel"], "kernel_ms_per_block": ms5 / bpl5 if bpl5 else 0.0, "repeats": len(S5["dts"]),
                "bus_peak": S5["peak"],
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, block)
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
