# usage (GPU box): bash tools/ktrace.sh [bench args]  — rocprofv3 kernel trace of bench.py, prints per-kernel average durations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --strong-c5-voices 0 --no-realtime "$@" > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]:>6s}%')
PY
