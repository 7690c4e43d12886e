# usage (GPU box): bash tools/profile_round.sh rNN — writes the judged summaries of the default bench command under gpurun_out/profiles_rNN/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/profiles_$R
rm -rf $O; mkdir -p $O
python bench.py > $O/${R}_headline_bench.json 2> $O/bench.err
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --no-cpu-baseline > $O/${R}_headline_bench_under_rocprofv3.json 2>/tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/${R}_headline_rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  python3 - $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $c > $O/${R}_headline_pmc_$c.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("kernel,counter,dispatches,sum,avg_per_dispatch")
for (k, c), (s, n) in sorted(acc.items()):
    if k.startswith("pg_"): print(f'"{k}",{c},{n},{s:.1f},{s/n:.1f}')
PY
done
python3 - $O $R <<'PY'
import csv, json, sys, os
O, R = sys.argv[1], sys.argv[2]
def avg(counter, kernel):
    rows = [r for r in csv.DictReader(open(os.path.join(O, f"{R}_headline_pmc_{counter}.csv"))) if r["kernel"].startswith(kernel)]
    # steady-state dispatches dominate (25 per run); FETCH_SIZE / WRITE_SIZE are in KiB
    return float(rows[0]["avg_per_dispatch"]) if rows else None
k = "pg_stage_fused_kernel"
f, w = avg("FETCH_SIZE", k), avg("WRITE_SIZE", k)
d = {"workload": "headline", "voices_per_gpu": 1024, "block_frames": 1024, "kernel": k,
     "FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w,
     "correction": "gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM/rocprofv3 section): read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE exact",
     "traffic_bytes_per_launch": (2 * f + w) * 1024 if f and w else None,
     "algorithmic_bytes_per_launch": 423.4 * 1024 * 1024,
     "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline (one pass per counter)"}
json.dump(d, open(os.path.join(O, f"{R}_headline_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(d))
PY
cat $O/${R}_headline_bench.json
