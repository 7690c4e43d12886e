# usage (GPU box): bash tools/pmc_c.sh <workload> "<counters pass 1>" "<counters pass 2>" ...  — tools/pmc.sh for one of the other workloads: one rocprofv3 --pmc pass per argument, per-kernel averages of the long (super-block) dispatches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=$1; shift
i=0
for c in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmcw_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcw_$i -- python3 bench.py --workload $W --superblock 16 --steps 64 --warmup 32 --repeats 2 --no-cpu-baseline --strong-c5-voices 0 --no-realtime > /tmp/pmcw_$i.log 2>&1
  f=$(find /tmp/pmcw_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("pg_"): acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    top = max(v); v = [x for x in v if x > 0.5 * top] or v   # the 16-block dispatches
    print(f"{k:32s} {c:24s} n={len(v):4d} avg/dispatch {sum(v)/len(v):18.1f}")
PY
done
