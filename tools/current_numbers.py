"""Regenerates the "current numbers" table of DESIGN.md (between the current-numbers markers) from the latest round's files under profiles/
(one page of current truth, not four rounds of narrative).   python tools/current_numbers.py [rNN]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles")


def load(name):
    f = os.path.join(P, f"{R}_{name}")
    if not os.path.exists(f):
        return None
    try:
        return json.load(open(f))
    except Exception:
        return None


def jl(name):
    f = os.path.join(P, f"{R}_{name}")
    return [json.loads(l) for l in open(f) if l.strip().startswith("{")] if os.path.exists(f) else []


rows = []
add = lambda what, value, src: rows.append(f"| {what} | {value} | `profiles/{R}_{src}` |")


def bench_row(label, name, extra=""):
    d = load(name)
    if not d:
        return
    r, c = d["roofline"], d["config"]
    clk = c.get("clocks") or {}
    gm = clk.get("gpu_metrics") or {}
    state = ""
    if clk.get("sclk_mhz"):
        state = f"; sclk p50 {clk['sclk_mhz']['p50']:.0f} MHz, {clk['socket_power_w']['p50']:.0f} W" + (f", power-limit throttled {100 * gm['ppt_throttled_share']:.0f} % of the time" if gm.get("ppt_throttled_share") is not None else "")
    dist = f" (legs p10 / p50 / p90 {r['frac_p10']:.3f} / {r['frac_p50']:.3f} / {r['frac_p90']:.3f})" if "frac_p10" in r else ""
    add(label, f"{d['value'] / 1e9:.2f} G voice-frames/s, {d['ms_per_step']:.4f} ms per step; `{r['kernel'].split(' ')[0]}` {r['kernel_ms_per_block'] * 1e3:.1f} µs per block = **{r['frac']:.3f}** of the {r['bound']} roofline{dist}{state}{extra}", name)
    rt = c.get("realtime")
    if rt:
        add(label + " — one call per block", f"{rt['ms_per_step']:.4f} ms per step, **{rt['roofline_frac']:.3f}**", name)
    s5 = c.get("strong_c5")
    if s5:
        add("C5 at 8192 voices on this GPU (`config.strong_c5`)", f"{s5['value'] / 1e9:.2f} G voice-frames/s, {s5['ms_per_step']:.4f} ms per step, {s5['roofline_frac']:.3f}", name)


bench_row("H, default line (`python bench.py`)", "headline_bench.json")
bench_row("H, the driver's arguments (`--steps 20 --warmup 5`)", "headline_bench_driver_args.json")
dk = load("headline_rocprofv3_dominant_kernel.json")
if dk:
    add("H under `rocprofv3 --kernel-trace`: dominant kernel", f"{dk['avg_us_per_block']:.2f} µs per block by the profiler, {dk.get('same_run_hipevent_us_per_block', 0):.2f} by the same run's hipEvents "
        f"(ratio {dk.get('rocprof_over_hipevent', 0):.3f}); fraction {dk.get('same_run_roofline_frac_rocprof', 0):.3f} / {dk.get('same_run_roofline_frac_hipevent', 0):.3f}", "headline_rocprofv3_dominant_kernel.json")
tr = load("headline_pmc_traffic.json")
if tr and tr.get("traffic_bytes_per_block"):
    add("H: HBM traffic by PMC counters", f"{tr['traffic_bytes_per_block'] / 1e6:.1f} MB per block = × {tr['traffic_bytes_per_block'] / tr['algorithmic_bytes_per_block']:.3f} of the algorithmic {tr['algorithmic_bytes_per_block'] / 1e6:.1f} MB", "headline_pmc_traffic.json")
for b in (128, 256, 512, 1024, 2048, 4096):
    d = load(f"headline_block{b}_bench.json")
    if d:
        rt = d["config"].get("realtime", {})
        add(f"H at {b}-frame callbacks", f"offline {d['ms_per_step']:.4f} ms per step ({d['roofline']['frac']:.3f}); one call per callback {rt.get('ms_per_step', 0):.4f} ms (**{rt.get('roofline_frac', 0):.3f}**)", f"headline_block{b}_bench.json")
for w, label in (("c2", "C2 (64 voices, Eq5 + Reverb on the bus)"), ("c3", "C3 (1024 mono voices, Filter + Chorus)"), ("c4", "C4 (256 voices, bus limiter)"), ("c5", "C5 (1024 voices, Filter → Eq5 → Delay → Reverb)")):
    bench_row(label, f"{w}_bench.json")
bench_row("C5 at 8192 voices, one GPU", "c5_8192v_bench.json")
for tag, what in (("c5", "C5 (1024 voices)"), ("c5_8192v", "C5 at 8192 voices")):
    dk = load(f"{tag}_rocprofv3_dominant_kernel.json")
    if dk:
        add(f"{what} under `rocprofv3 --kernel-trace`: dominant kernel", f"{dk['avg_us_per_block']:.2f} µs per block by the profiler, {dk.get('same_run_hipevent_us_per_block', 0):.2f} by the same run's hipEvents "
            f"(ratio {dk.get('rocprof_over_hipevent', 0):.3f}); fraction {dk.get('same_run_roofline_frac_rocprof', 0):.3f} / {dk.get('same_run_roofline_frac_hipevent', 0):.3f}", f"{tag}_rocprofv3_dominant_kernel.json")
tr = load("c5_pmc_traffic.json")
if tr and tr.get("traffic_bytes_per_block"):
    add("C5: HBM traffic by PMC counters", f"{tr['traffic_bytes_per_block'] / 1e6:.1f} MB per block = × {tr['traffic_bytes_per_block'] / tr['algorithmic_bytes_per_block']:.3f} of the algorithmic {tr['algorithmic_bytes_per_block'] / 1e6:.1f} MB", "c5_pmc_traffic.json")
for b in (128, 256, 512, 1024, 2048, 4096):
    d = load(f"c5_block{b}_bench.json")
    if d:
        rt = d["config"].get("realtime", {})
        add(f"C5 at {b}-frame callbacks", f"offline {d['ms_per_step']:.4f} ms per step ({d['roofline']['frac']:.3f}); one call per callback {rt.get('ms_per_step', 0):.4f} ms (**{rt.get('roofline_frac', 0):.3f}**)", f"c5_block{b}_bench.json")
for d in jl("dynamic.jsonl"):
    c, y = d["config"], d["dyn"]
    add(f"dyn: {c['events_per_s']:g} events/s (kinds {c.get('kinds', '012')}), churn {c['churn_pct_per_s']:g} %/s, silent {c['silent_pct']:g} %",
        f"offline {y['offline']['ms_per_step']:.4f} ms per step (× {y['ratio_offline']:.2f} of steady), one call per block {y['realtime']['ms_per_step']:.4f} (× {y['ratio_realtime']:.2f}); "
        f"{100 * y['offline']['deferred_share']:.2f} % of unit-blocks on the generic kernel, {y['offline']['generic_ms_per_launch'] * 1e3:.0f} µs per generic launch", "dynamic.jsonl")
pe = jl("per_effect.jsonl")
if pe:
    steady = sorted((d for d in pe if d["case"] == "steady"), key=lambda d: -d["ms_per_block"])
    add("per effect, 1024 units, steady (slowest first)", "; ".join(f"{d['effect']} {d['ms_per_block']:.3f}" for d in steady) + " ms per block", "per_effect.jsonl")
    ramp = sorted((d for d in pe if d["case"] == "ramp"), key=lambda d: -d["ms_per_block"])
    add("per effect, a parameter command on every unit every second call", "; ".join(f"{d['effect']} {d['ms_per_block']:.3f}" for d in ramp) + " ms per block", "per_effect.jsonl")
for d in jl("generic_paths.jsonl"):
    if d.get("blocks_per_call", 1) > 1:
        add(f"layout `{d['variant']}`, {d['units']} units, {d['blocks_per_call']} blocks per call", f"{d['ms_per_block']:.4f} ms per block, {d['voice_frames_per_s'] / 1e9:.2f} G voice-frames/s", "generic_paths.jsonl")
fz = jl("fuzz_summary.jsonl")
for d in fz:
    add(f"fuzz campaign (seeds {d['seeds']}, base {d['base']})", f"{d['requested']} cases requested, {d['run']} run, {d['passed']} passed, {d['skipped_by_classifier']} skipped by the discontinuity classifier, "
        f"{d['skipped_other']} skipped otherwise, **{d['failed']} failed**" + (f"; cut by {d['cut_by']}" if d.get("cut_by") else ""), "fuzz_summary.jsonl")
table = "| what | value | file |\n|---|---|---|\n" + "\n".join(rows) + "\n"
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
a, b = "<!-- current-numbers:begin -->", "<!-- current-numbers:end -->"
if a in text and b in text:
    text = text[: text.index(a) + len(a)] + "\n" + table + text[text.index(b):]
    open(path, "w").write(text)
    print(f"DESIGN.md: {len(rows)} rows")
else:
    print(table)
