# usage (GPU box): bash tools/pmc.sh "<counters pass 1>" "<counters pass 2>" ...   — one rocprofv3 --pmc pass per argument; prints per-kernel averages
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --strong-c5-voices 0 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    if "pg_unit_kernel_fast" in k: print(f"{k:40s} {c:28s} avg/launch {s/n:16.1f}  (n={n})")
PY
done
