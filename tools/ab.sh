# usage (on the GPU box): bash tools/ab.sh [reps] [bench args]  — benches every prebuilt tools/ab_libs/*.so, interleaved, on the same box
cd $GRAFT_REPO_ROOT
cp phonic_amd/csrc/libphonic_gpu.so /tmp/keep.so
reps=${1:-2}; shift
for rep in $(seq 1 $reps); do
for f in tools/ab_libs/*.so; do
  cp $f phonic_amd/csrc/libphonic_gpu.so
  echo -n "[$(basename $f .so)] "
  python bench.py --steps 96 --warmup 32 --repeats 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,1),'Mvf/s step', round(d['ms_per_step'],4), 'kernel/block', round(r['kernel_ms_per_block'],4),'ms frac', round(r['frac'],4))"
done
done
cp /tmp/keep.so phonic_amd/csrc/libphonic_gpu.so
