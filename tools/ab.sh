# usage (on the GPU box): bash tools/ab.sh [reps] [bench args]  — benches every prebuilt tools/ab_libs/*.so, interleaved, on the same box
# (the variant is selected with PHONIC_LIB: the in-tree library is never overwritten)
cd $GRAFT_REPO_ROOT
reps=${1:-2}; shift
for rep in $(seq 1 $reps); do
for f in tools/ab_libs/*.so; do
  echo -n "[$(basename $f .so)] "
  PHONIC_LIB=$PWD/$f python bench.py --steps 96 --warmup 32 --repeats 3 --no-cpu-baseline --strong-c5-voices 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; rt=d['config'].get('realtime',{}); print(round(d['value']/1e6,1),'Mvf/s step', round(d['ms_per_step'],4), 'kernel/block', round(r['kernel_ms_per_block'],4),'ms frac', round(r['frac'],4), '| real-time frac', round(rt.get('roofline_frac',0),4), 'step', round(rt.get('ms_per_step',0),4))"
done
done
