# usage (GPU box): bash tools/block_sweep.sh rNN — the headline at callback sizes of 1024 / 2048 / 4096 frames (max_frames stays 1024: a write is
# walked in the reference's <= 4096-frame chunks, rendered as 1024-frame pieces by the staged kernels), interleaved on one box
cd $GRAFT_REPO_ROOT
R=${1:-r04}
O=gpurun_out/profiles_$R
mkdir -p $O
for rep in 1 2; do
  for b in 1024 2048 4096; do
    python bench.py --block $b --steps $((20480 / b)) --warmup $((4096 / b * 2)) --no-cpu-baseline > $O/${R}_headline_block${b}_rep${rep}.json 2>> $O/bench.err || { echo "bench --block $b failed"; tail -5 $O/bench.err; exit 1; }
  done
done
python - $O $R <<'PY'
import json, sys, os, glob
O, R = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(O, f"{R}_headline_block*_rep*.json"))):
    d = json.load(open(f)); r = d["roofline"]; rt = d["config"].get("realtime", {})
    print(f"{os.path.basename(f):42s} value {d['value']/1e9:6.2f} G vf/s  ms/step {d['ms_per_step']:.4f}  frac {r['frac']:.3f}  kernel/piece {r['kernel_ms_per_block']:.4f}  calls of {d['config']['blocks_per_call']} steps | "
          f"real-time: {rt.get('value', 0)/1e9:6.2f} G vf/s frac {rt.get('roofline_frac', 0):.3f}")
PY
