# usage (GPU box): bash tools/ab_super.sh <tag> — the headline bench with one block per launch and with super-blocks of 4 / 16 / 32, interleaved twice
cd $GRAFT_REPO_ROOT
T=${1:-ab}
O=gpurun_out/$T
mkdir -p $O
for rep in 1 2; do
  for sb in 1 4 16 32; do
    timeout -k 10 200 python bench.py --superblock $sb --steps 96 --warmup 32 --no-cpu-baseline > $O/sb${sb}_$rep.json 2> $O/sb${sb}_$rep.err || echo "FAILED sb=$sb"
    python - $O/sb${sb}_$rep.json $sb <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"superblock {sys.argv[2]:>3}: ms/step {d['ms_per_step']:.4f}  kernel ms/block {r['kernel_ms_per_block']:.4f}  frac {r['frac']:.3f}  blocks/launch {r['blocks_per_launch']}")
PY
  done
done
