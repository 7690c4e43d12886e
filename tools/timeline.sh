# usage (GPU box): bash tools/timeline.sh [bench args] — per-dispatch timeline (start offsets / durations in us) of two steady-state rounds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("pg_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# pick two consecutive rounds late in the run
idx = [i for i, r in enumerate(rows) if (r["Kernel_Name"].startswith("pg_mix_kernel_2") or r["Kernel_Name"].startswith("pg_mix_kernel("))]
a = idx[-4] + 1; b = idx[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:34]:34s} start +{(s-t0)/1e3:8.1f} us  gap {(s-prev_end)/1e3:6.1f}  dur {(e-s)/1e3:7.1f} us  grid {r.get("Grid_Size_X","?")}')
    prev_end = e
PY
