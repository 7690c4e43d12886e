# usage (GPU box): bash tools/gpu_round.sh <tag> — full -m gpu suite, then the headline bench and the C5 / 8192-voice strong-scaling anchor
cd $GRAFT_REPO_ROOT
T=${1:-run}
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -25 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"; cat $O/bench_headline.json
timeout -k 10 300 python bench.py --workload c5 --scaling strong --total-voices 8192 --steps 32 --warmup 8 --no-cpu-baseline > $O/bench_c5_8192v.json 2> $O/bench_c5.err; echo "c5 rc=$?"; cat $O/bench_c5_8192v.json
