# usage (GPU box): bash tools/pmc_generic.sh [tag] — what the generic kernel's ONE unit of a commanded round spends its cycles on: rocprofv3 --pmc
# passes (the kernels of a round run one after the other under the profiler, so the generic kernel has the chip to itself) over
# tools/diag_cmd.py's one-command-per-block rounds -> gpurun_out/pmc_generic_<tag>.csv (per counter: mean over pg_unit_kernel's dispatches
# with work, i.e. those whose SQ_WAVES-independent value lies in the upper half)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05}
OUT=gpurun_out/pmc_generic_$T.csv
mkdir -p gpurun_out
echo "kernel,counter,dispatches,avg_per_dispatch,max" > $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_WAVES SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pmcg_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcg_$i -- python3 tools/diag_cmd.py 1024 24 > /tmp/pmcg_$i.log 2>&1
  f=$(find /tmp/pmcg_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# pass '$c' produced no counters: $(tail -1 /tmp/pmcg_$i.log | cut -c1-160)" >> $OUT; continue; fi
  cp "$f" gpurun_out/pmc_generic_${T}_pass$i.csv
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if k in ("pg_unit_kernel", "pg_stage_fused_kernel"):
        top = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
        print(f'"{k}",{c},{len(top)},{sum(top)/len(top):.1f},{max(v):.1f}')
PY
done
cat $OUT
