# usage (GPU box): bash tools/c3_round.sh <tag> — C3 (1024 mono voices, Filter -> Chorus): bench with / without super-blocks, kernel trace, utilisation counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-c3}
O=gpurun_out/$T
mkdir -p $O
for sb in 1 32; do
  timeout -k 10 200 python bench.py --workload c3 --superblock $sb --steps 96 --warmup 32 --no-cpu-baseline --strong-c5-voices 0 > $O/c3_sb$sb.json 2> $O/c3_sb$sb.err || echo FAILED
  python - $O/c3_sb$sb.json $sb <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"c3 superblock {sys.argv[2]:>3}: ms/step {d['ms_per_step']:.4f} kernel ms/block {r['kernel_ms_per_block']:.4f} frac {r['frac']:.4f} kernel {r['kernel']}")
PY
done
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --workload c3 --steps 96 --warmup 32 --repeats 3 --no-realtime --no-cpu-baseline --strong-c5-voices 0 > /dev/null 2>/tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/c3_rocprofv3_kernel_stats.csv
head -6 $O/c3_rocprofv3_kernel_stats.csv
OUT=$O/c3_pmc_util.csv
echo "kernel,counter,dispatches,avg_per_dispatch" > $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); rm -rf /tmp/pmcc_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcc_$i -- python3 bench.py --workload c3 --superblock 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --strong-c5-voices 0 --no-realtime > /tmp/pmcc_$i.log 2>&1
  f=$(find /tmp/pmcc_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# pass '$c' produced no counters" >> $OUT; continue; fi
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("pg_"): acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = sorted(v)[len(v)//4:]
    print(f'"{k}",{c},{len(v)},{sum(v)/len(v):.1f}')
PY
done
cat $OUT
