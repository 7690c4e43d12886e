# usage (on the GPU box): bash tools/ab_c.sh <workload> [reps]   — like ab.sh for one of the other workloads (c2..c5)
cd $GRAFT_REPO_ROOT
cp phonic_amd/csrc/libphonic_gpu.so /tmp/keep.so
for rep in $(seq 1 ${2:-2}); do
for f in tools/ab_libs/*.so; do
  cp $f phonic_amd/csrc/libphonic_gpu.so
  echo -n "[$(basename $f .so)] "
  python bench.py --workload $1 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'Mvf/s step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4),'ms frac', round(d['roofline']['frac'],4))"
done
done
cp /tmp/keep.so phonic_amd/csrc/libphonic_gpu.so
