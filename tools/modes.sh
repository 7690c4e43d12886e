# usage (GPU box): bash tools/modes.sh [reps]  — bench the three reverb render modes interleaved on the same box, report the best of reps
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${1:-3}); do for m in 1 2 0; do python bench.py --steps 80 --warmup 20 --no-cpu-baseline --staged $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($m, d['ms_per_step'], d['roofline']['kernel_ms'])"; done; done > /tmp/modes.txt
python - <<'PY'
import collections
b = collections.defaultdict(lambda: [9, 9])
for l in open('/tmp/modes.txt'):
    m, s, k = l.split(); b[m][0] = min(b[m][0], float(s)); b[m][1] = min(b[m][1], float(k))
for m in sorted(b): print(f"[staged={m}] best step {b[m][0]:.4f} ms  kernel {b[m][1]:.4f} ms  frac {423.4*1024*1024/(b[m][1]*1e-3)/1e9/8000:.4f}  value {1024*1024/(b[m][0]*1e-3)/1e6:.0f} Mvf/s")
PY
