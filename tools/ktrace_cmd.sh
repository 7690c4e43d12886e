#!/bin/bash
# usage (GPU box): bash tools/ktrace_cmd.sh <tag> <bench args...> — rocprofv3 kernel trace of `python3 bench.py <args>`: per-kernel calls / average / share,
# the bench line itself in gpurun_out/ktrace_<tag>.json, the stats CSV in gpurun_out/ktrace_<tag>_kernel_stats.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python3 bench.py "$@" > gpurun_out/ktrace_$tag.json 2> /tmp/kt_$tag.err || { echo "bench under rocprofv3 failed"; tail -5 /tmp/kt_$tag.err; }
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/ktrace_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.2: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:9.1f}  {r["Percentage"]:>6s}%')
PY
t=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$t" > gpurun_out/ktrace_${tag}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last 400 dispatches in time order: start (us), duration (us), gap to the previous end, kernel
prev_end = None
for r in rows[-400:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f'{(s - t0) / 1e3:12.1f} us  dur {(e - s) / 1e3:9.1f}  gap {gap:8.1f}  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>8s}  {r["Kernel_Name"][:48]}')
    prev_end = e
PY
