# usage (GPU box): bash tools/voices_sweep.sh <tag> — the headline at 512 .. 4096 voices per GPU (super-blocks of 32): where the roofline fraction saturates
cd $GRAFT_REPO_ROOT
T=${1:-sweep}; O=gpurun_out/$T; mkdir -p $O
for v in 512 1024 2048 4096; do
  timeout -k 10 300 python bench.py --voices $v --steps 64 --warmup 32 --repeats 3 --no-cpu-baseline > $O/headline_${v}v.json 2> $O/headline_${v}v.err || echo FAILED
  python - $O/headline_${v}v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"voices {sys.argv[2]:>5}: ms/step {d['ms_per_step']:.4f} kernel ms/block {r['kernel_ms_per_block']:.4f} frac {r['frac']:.3f} value {d['value']/1e9:.2f} G vf/s")
PY
done
