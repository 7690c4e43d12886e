"""Determinism check of the bus workloads (C2, C4): bench.py's `config.bus_peak` — the peak of the whole rendered master bus — from super-block calls
(the next launch sequence's unit kernels run under the bus chain on a second stream) against single-block calls, N times each. Any spread is a race.
    python tools/exp_bus_peak_repeat.py [N]      (GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--strong-c5-voices", "0", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    if out.returncode != 0: return ("rc", out.returncode, out.stderr[-400:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    return d["config"]["bus_peak"]
for wl in ("c2", "c4"):
    sup = [run(["--steps", "32", "--warmup", "8", "--repeats", "3", "--workload", wl, "--superblock", "16", "--no-realtime"]) for _ in range(N)]
    one = [run(["--steps", "32", "--warmup", "8", "--repeats", "3", "--workload", wl, "--superblock", "1"]) for _ in range(max(2, N // 4))]
    print(wl, "super-block 16:", sorted(set(map(repr, sup))), "| single:", sorted(set(map(repr, one))), flush=True)
d = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "4", "--voices", "64", "--min-seconds", "0.2", "--no-cpu-baseline", "--strong-c5-voices", "0"], cwd=ROOT, capture_output=True, text=True)
j = json.loads([l for l in d.stdout.splitlines() if l.startswith("{")][0])
print("min-seconds leg:", j["repeats"], j["config"]["realtime"]["timed_seconds"])
