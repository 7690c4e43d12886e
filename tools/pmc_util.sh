# usage (GPU box): bash tools/pmc_util.sh rNN — utilisation counters of the headline's dominant kernel, one rocprofv3 --pmc pass per group
# (separate from the kernel-trace / HBM-traffic passes of profile_round.sh) -> gpurun_out/profiles_rNN/rNN_headline_pmc_util.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r02}
O=gpurun_out/profiles_$R
mkdir -p $O
OUT=$O/${R}_headline_pmc_util.csv
echo "kernel,counter,super_block_dispatches(16 blocks),avg_per_dispatch" > $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/pmcu_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcu_$i -- python3 bench.py --superblock 16 --steps 64 --warmup 32 --repeats 2 --no-cpu-baseline --strong-c5-voices 0 --no-realtime > /tmp/pmcu_$i.log 2>&1
  f=$(find /tmp/pmcu_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# pass '$c' produced no counters" >> $OUT; continue; fi
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):   # steady-state (16-block) dispatches only: the larger half of the values; VALUBusy is a percentage
    if k.startswith("pg_stage_fused_kernel") or k.startswith("pg_mix_kernel"):
        top = v if c == "VALUBusy" else [x for x in v if x > 0.5 * max(v)]
        print(f'"{k}",{c},{len(top)},{sum(top)/len(top):.1f}')
PY
done
cat $OUT
