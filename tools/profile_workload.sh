#!/bin/bash
# usage (GPU box): bash tools/profile_workload.sh <tag> <dominant kernel> <bench args...> — tools/profile_round.sh's kernel trace (with the SAME run's
# hipEvents beside it) and HBM-traffic counter passes for another workload than the default line, e.g.
#   bash tools/profile_workload.sh c5 pg_stage_fused_wide_kernel --workload c5
#   bash tools/profile_workload.sh c5_8192v pg_stage_fused_wide_kernel --workload c5 --scaling strong --total-voices 8192
# writes gpurun_out/profiles_<tag>/r05_<tag>_{bench_under_rocprofv3.json, rocprofv3_kernel_stats.csv, rocprofv3_dominant_kernel.json, pmc_FETCH_SIZE.csv,
# pmc_WRITE_SIZE.csv, pmc_traffic.json}. PROFILE_PMC=0 skips the counter passes. One rocprofv3 run per counter, never combined with other trace domains.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; K=$2; shift 2
R=r05
O=gpurun_out/profiles_$tag
rm -rf $O; mkdir -p $O
SB=16
COMMON="--no-cpu-baseline --strong-c5-voices 0 --no-realtime --superblock $SB"
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python3 bench.py $COMMON --repeats 3 --steps 96 --warmup 32 "$@" > $O/${R}_${tag}_bench_under_rocprofv3.json 2> /tmp/kt_$tag.err || { echo "bench under rocprofv3 failed"; tail -5 /tmp/kt_$tag.err; exit 1; }
cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) $O/${R}_${tag}_rocprofv3_kernel_stats.csv
python3 - $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) $SB $K $O/${R}_${tag}_bench_under_rocprofv3.json > $O/${R}_${tag}_rocprofv3_dominant_kernel.json <<'PY'
import csv, json, sys
sb, k = int(sys.argv[2]), sys.argv[3]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].split("(")[0] == k]
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
top = [x for x in d if x > 0.5 * d[-1]]   # steady-state super-block dispatches (the first rounds of a run are single blocks)
b = json.load(open(sys.argv[4])); r = b["roofline"]; c = b["config"].get("clocks", {}); gm = c.get("gpu_metrics") or {}
us = sum(top) / len(top) / 1e3 / sb
out = {"kernel": k, "workload": b["config"]["workload"], "voices_per_gpu": b["config"]["voices_per_gpu"], "dispatches": len(d), "super_block_dispatches": len(top), "blocks_per_dispatch": sb,
       "avg_us_per_dispatch": us * sb, "avg_us_per_block": us,
       "same_run_hipevent_kernel": r.get("kernel"), "same_run_hipevent_us_per_block": r["kernel_ms_per_block"] * 1e3, "rocprof_over_hipevent": us / (r["kernel_ms_per_block"] * 1e3),
       "same_run_roofline_frac_hipevent": r["frac"],
       "same_run_roofline_frac_rocprof": r["bytes_per_voice_frame"] * b["config"]["voices_per_gpu"] * b["config"]["max_frames"] / (us * 1e-6) / 1e9 / r["peak"],
       "same_run_clocks": {"sclk_mhz_p50": (c.get("sclk_mhz") or {}).get("p50"), "socket_power_w_p50": (c.get("socket_power_w") or {}).get("p50"), "ppt_throttled_share": gm.get("ppt_throttled_share")},
       "note": "rocprofv3 --kernel-trace of bench.py --superblock %d: dispatches shorter than half the longest are the single-block rounds before the steady state" % sb}
print(json.dumps(out))
PY
cat $O/${R}_${tag}_rocprofv3_dominant_kernel.json
[ "${PROFILE_PMC:-1}" = "0" ] && exit 0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -- python3 bench.py $COMMON --repeats 2 --steps 64 --warmup 32 "$@" > /tmp/pmc_${tag}_$c.log 2>&1 || { echo "counter pass $c failed"; tail -5 /tmp/pmc_${tag}_$c.log; exit 1; }
  python3 - $(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1) $SB > $O/${R}_${tag}_pmc_$c.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
sb = int(sys.argv[2])
print("kernel,counter,dispatches,super_block_dispatches,avg_per_super_block_dispatch,avg_per_block")
for (k, c), v in sorted(acc.items()):
    if not k.startswith("pg_"): continue
    top = [x for x in v if x > 0.5 * max(v)] if k.startswith("pg_stage") or k.startswith("pg_unit_kernel_fast") or k.startswith("pg_mix") else v
    per = sb if len(top) < len(v) or k.startswith("pg_stage") else 1
    print(f'"{k}",{c},{len(v)},{len(top)},{sum(top)/len(top):.1f},{sum(top)/len(top)/per:.1f}')
PY
done
python3 - $O $R $tag $K <<'PY'
import csv, json, sys, os
sys.path.insert(0, os.getcwd())
from phonic_amd import _capi
O, R, tag, k = sys.argv[1:5]
b = json.load(open(os.path.join(O, f"{R}_{tag}_bench_under_rocprofv3.json")))
def per_block(counter):
    rows = [r for r in csv.DictReader(open(os.path.join(O, f"{R}_{tag}_pmc_{counter}.csv"))) if r["kernel"] == k]
    return float(rows[0]["avg_per_block"]) if rows else None   # FETCH_SIZE / WRITE_SIZE are in KiB
f, w = per_block("FETCH_SIZE"), per_block("WRITE_SIZE")
alg = b["roofline"]["bytes_per_voice_frame"] * b["config"]["voices_per_gpu"] * b["config"]["max_frames"]
t = (2 * f + w) * 1024 if f and w else None
d = {"workload": b["config"]["workload"], "voices_per_gpu": b["config"]["voices_per_gpu"], "block_frames": b["config"]["max_frames"], "kernel": k, "source_hash": _capi.source_hash(),
     "FETCH_SIZE_KiB_per_block": f, "WRITE_SIZE_KiB_per_block": w,
     "correction": "gfx950: FETCH_SIZE reports half of the bytes read (MI355X_MICROARCH.md, HBM/rocprofv3 section): read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE exact; calibration on this library's access pattern: profiles/r04/r04_ringstream_pmc.json",
     "traffic_bytes_per_block": t, "algorithmic_bytes_per_block": alg, "traffic_over_algorithmic": (t / alg) if t else None,
     "command": "tools/profile_workload.sh: rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py <workload args> --superblock 16 --steps 64 --warmup 32 --repeats 2 --no-cpu-baseline --strong-c5-voices 0 --no-realtime (one pass per counter; per block = the 16-block dispatches / 16)"}
json.dump(d, open(os.path.join(O, f"{R}_{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(d))
PY
