# usage: bash tools/sweep.sh "<FAST_WAVES value + extra -D flags>" ...   (runs on the GPU box; rebuilds per variant)
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  (cd phonic_amd/csrc && rm -f *.o libphonic_gpu.so && make -s FAST_WAVES="$f" 2>&1 | grep -i " error" | head -5)
  echo "== FAST_WAVES=$f"
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'Mvf/s step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4),'ms frac', round(d['roofline']['frac'],4))"
done
