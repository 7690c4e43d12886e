# usage (GPU box): bash tools/bus_group_sweep.sh — C2 / C4 with the bus chain launched per 4 / 8 / 16 / 32 blocks (PHONIC_BUS_GROUP), in-tree library
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for w in c2 c4; do
for g in 4 8 16 32; do
  echo -n "[$w group $g] "
  PHONIC_BUS_GROUP=$g python bench.py --workload $w --steps 64 --warmup 16 --no-cpu-baseline --strong-c5-voices 0 --no-realtime 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'kernel/block', round(d['roofline']['kernel_ms_per_block'],4), 'unit kernels', d['roofline'].get('unit_kernels'))"
done
done
done
