#!/bin/bash
# usage (box): bash tools/ab_blocks.sh [reps] — every tools/ab_libs/*.so at callback sizes of 128 / 256 / 512 / 1024 frames, interleaved on one box:
# offline calls and the real-time pattern (one call per callback)
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${1:-2}); do
for b in 128 256 512 1024; do
for f in tools/ab_libs/*.so; do
  echo -n "[$(basename $f .so)] block $b: "
  PHONIC_LIB=$PWD/$f python bench.py --block $b --steps 40 --warmup 10 --no-cpu-baseline --strong-c5-voices 0 --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; rt=d['config'].get('realtime',{}); print('offline step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), '| real-time step', round(rt.get('ms_per_step',0),4), 'frac', round(rt.get('roofline_frac',0),4), 'kernel/blk', round(rt.get('kernel_ms_per_block',0),4))"
done
done
done
