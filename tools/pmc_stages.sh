# usage (GPU box): bash tools/pmc_stages.sh — HBM read / write bytes per launch of each stage kernel (one launch per stage mode)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --staged 2 > /tmp/pmcs_$c.log 2>&1
  python3 - $(find /tmp/pmcs_$c -name "*counter_collection.csv" | head -1) $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("pg_"): acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
mult = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0   # gfx950: FETCH_SIZE counts half of wide coalesced reads
for k, v in sorted(acc.items()):
    v = sorted(v)[len(v)//4:]   # drop the first (deferred / cold) launches
    print(f"{sys.argv[2]:10s} {k:28s} {mult*sum(v)/len(v)*1024/1e6:8.1f} MB per launch")
PY
done
