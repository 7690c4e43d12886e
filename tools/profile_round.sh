# usage (GPU box): bash tools/profile_round.sh rNN — writes the judged summaries of the default bench command under gpurun_out/profiles_rNN/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r05}
O=gpurun_out/profiles_$R
rm -rf $O; mkdir -p $O
SB=16   # blocks per launch in the profiled runs (the counter passes need every steady-state dispatch to render the same number of blocks)
: > $O/bench.err
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline --strong-c5-voices 0 --no-realtime --repeats 3 --superblock $SB --steps 96 --warmup 32 > $O/${R}_headline_bench_under_rocprofv3.json 2>/tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/${R}_headline_rocprofv3_kernel_stats.csv
# per-dispatch durations of the dominant kernel: steady-state super-block dispatches only (the first rounds of a run are single blocks)
python3 - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) $SB > $O/${R}_headline_rocprofv3_dominant_kernel.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("pg_stage_fused_kernel")]
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
top = [x for x in d if x > 0.5 * d[-1]]
sb = int(sys.argv[2])
print(json.dumps({"kernel": "pg_stage_fused_kernel", "dispatches": len(d), "super_block_dispatches": len(top), "blocks_per_dispatch": sb,
                  "avg_us_per_dispatch": sum(top) / len(top) / 1e3, "avg_us_per_block": sum(top) / len(top) / 1e3 / sb,
                  "note": "rocprofv3 --kernel-trace of bench.py --superblock %d: dispatches shorter than half the longest are the single-block rounds before the steady state" % sb}))
PY
# ... and next to it what the SAME run's bench line says (hipEvents riding on the dispatches) and the clocks / power it ran at: the two must agree
python3 - $O/${R}_headline_rocprofv3_dominant_kernel.json $O/${R}_headline_bench_under_rocprofv3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
r = b["roofline"]; c = b["config"].get("clocks", {})
d["same_run_hipevent_us_per_block"] = r["kernel_ms_per_block"] * 1e3
d["rocprof_over_hipevent"] = d["avg_us_per_block"] / (r["kernel_ms_per_block"] * 1e3)
d["same_run_roofline_frac_hipevent"] = r["frac"]
d["same_run_roofline_frac_rocprof"] = r["bytes_per_voice_frame"] * b["config"]["voices_per_gpu"] * b["config"]["max_frames"] / (d["avg_us_per_block"] * 1e-6) / 1e9 / r["peak"]
gm = c.get("gpu_metrics") or {}
d["same_run_clocks"] = {"sclk_mhz_p50": (c.get("sclk_mhz") or {}).get("p50"), "socket_power_w_p50": (c.get("socket_power_w") or {}).get("p50"), "ppt_throttled_share": gm.get("ppt_throttled_share"),
                        "xcd_sclk_mhz": gm.get("xcd_sclk_mhz"), "temp_memory_c_p50": (c.get("temp_memory_c") or {}).get("p50")}
json.dump(d, open(sys.argv[1], "w"))
print(json.dumps(d))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 bench.py --superblock $SB --steps 64 --warmup 32 --repeats 2 --no-cpu-baseline --strong-c5-voices 0 --no-realtime > /tmp/pmc_$c.log 2>&1
  python3 - $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $c $SB > $O/${R}_headline_pmc_$c.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
sb = int(sys.argv[3])
print("kernel,counter,dispatches,super_block_dispatches,avg_per_super_block_dispatch,avg_per_block")
for (k, c), v in sorted(acc.items()):
    if not k.startswith("pg_"): continue
    top = [x for x in v if x > 0.5 * max(v)] if k.startswith("pg_stage") or k.startswith("pg_unit_kernel_fast") or k.startswith("pg_mix") else v
    per = sb if len(top) < len(v) or k.startswith("pg_stage") else 1
    print(f'"{k}",{c},{len(v)},{len(top)},{sum(top)/len(top):.1f},{sum(top)/len(top)/per:.1f}')
PY
done
python3 - $O $R <<'PY'
import csv, json, sys, os
sys.path.insert(0, os.getcwd())
from phonic_amd import _capi
O, R = sys.argv[1], sys.argv[2]
def per_block(counter, kernel):
    rows = [r for r in csv.DictReader(open(os.path.join(O, f"{R}_headline_pmc_{counter}.csv"))) if r["kernel"].startswith(kernel)]
    return float(rows[0]["avg_per_block"]) if rows else None   # FETCH_SIZE / WRITE_SIZE are in KiB
k = "pg_stage_fused_kernel"
f, w = per_block("FETCH_SIZE", k), per_block("WRITE_SIZE", k)
d = {"workload": "headline", "voices_per_gpu": 1024, "block_frames": 1024, "kernel": k, "source_hash": _capi.source_hash(),
     "FETCH_SIZE_KiB_per_block": f, "WRITE_SIZE_KiB_per_block": w,
     "correction": "gfx950: FETCH_SIZE reports half of the bytes read (MI355X_MICROARCH.md, HBM/rocprofv3 section, calibrated there for 16-byte lanes): read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE exact",
     "calibration": "the same factors hold for THIS kernel's access pattern (8-byte lanes, 128-frame sub-chunks, 2 KB granules): tools/ringstream with a known byte count reads FETCH_SIZE x 1.94-2.17 (the spread = whether the vibrato look-ahead of up to 15 frames per window is counted as read twice) and WRITE_SIZE x 0.98-0.99 (partial lines at window edges) - profiles/r04/r04_ringstream_pmc.json",
     "traffic_bytes_per_block": (2 * f + w) * 1024 if f and w else None,
     "algorithmic_bytes_per_block": 423.4 * 1024 * 1024,
     "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py --superblock 16 --steps 64 --warmup 32 --repeats 2 --no-cpu-baseline --strong-c5-voices 0 --no-realtime (one pass per counter; per block = the 16-block dispatches / 16)"}
json.dump(d, open(os.path.join(O, f"{R}_headline_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(d))
PY
# the bench lines last: bench.py quotes roofline.traffic from a profiles/*_pmc_traffic.json whose source hash matches the library it runs
cp $O/${R}_headline_pmc_traffic.json profiles/${R}_headline_pmc_traffic.json
python bench.py > $O/${R}_headline_bench.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/${R}_headline_bench_driver_args.json 2>> $O/bench.err
cat $O/${R}_headline_bench.json
