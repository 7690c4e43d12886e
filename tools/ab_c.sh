# usage (on the GPU box): bash tools/ab_c.sh <workload> [reps]   — like ab.sh for one of the other workloads (c2..c5); variants via PHONIC_LIB
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
for f in tools/ab_libs/*.so; do
  echo -n "[$(basename $f .so)] "
  PHONIC_LIB=$PWD/$f python bench.py --workload $1 --steps 64 --warmup 16 --no-cpu-baseline --strong-c5-voices 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'Mvf/s step', round(d['ms_per_step'],4), 'kernel/block', round(d['roofline']['kernel_ms_per_block'],4),'ms frac', round(d['roofline']['frac'],4), d['roofline']['bound'], '| one call per block:', round((d['config'].get('realtime') or {}).get('ms_per_step', 0), 4), 'ms')"
done
done
