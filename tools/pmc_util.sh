# usage (GPU box): bash tools/pmc_util.sh rNN — utilisation counters of the headline's dominant kernel, one rocprofv3 --pmc pass per group
# (separate from the kernel-trace / HBM-traffic passes of profile_round.sh) -> gpurun_out/profiles_rNN/rNN_headline_pmc_util.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/profiles_$R
mkdir -p $O
OUT=$O/${R}_headline_pmc_util.csv
echo "kernel,counter,dispatches,avg_per_dispatch" > $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/pmcu_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcu_$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/pmcu_$i.log 2>&1
  f=$(find /tmp/pmcu_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# pass '$c' produced no counters" >> $OUT; continue; fi
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    if k.startswith("pg_stage_fused_kernel") or k.startswith("pg_mix_kernel"): print(f'"{k}",{c},{n},{s/n:.1f}')
PY
done
cat $OUT
