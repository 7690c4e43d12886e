# usage (GPU box): bash tools/pmc_stage_insts.sh — instruction mix and busy cycles per launch of each stage kernel (one launch per stage mode,
# --staged 2), one rocprofv3 --pmc pass per counter group
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  rm -rf /tmp/pmci_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmci_$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --staged 2 --superblock 1 > /tmp/pmci_$i.log 2>&1
  f=$(find /tmp/pmci_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# pass '$c' produced no counters"; tail -3 /tmp/pmci_$i.log; continue; fi
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("pg_"): acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = sorted(v)[len(v)//4:]
    print(f"{k:28s} {c:30s} {sum(v)/len(v):14.0f} per launch")
PY
done
