#!/bin/bash
# (box) round 5, second measurement call: tests, callback-size sweep, per-effect table, a4 / other layouts, sharded host time, C3 stamps, dyn sweep
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for b in 128 256 512 1024 2048 4096; do
  python bench.py --steps $((40960 / b > 160 ? 160 : 40960 / b)) --warmup 10 --block $b --no-cpu-baseline --strong-c5-voices 0 > $O/r05_headline_block${b}_bench.json 2>> $O/err.log; echo "block $b rc=$?"
done
python tools/per_effect.py 1024 > $O/r05_per_effect.jsonl 2>> $O/err.log; echo "per_effect rc=$?"
python tools/exp_generic_paths.py 1024 headline,stream,resampled,nested > $O/r05_generic_paths.jsonl 2>> $O/err.log; echo "generic_paths rc=$?"
( python tools/exp_sharded_host_time.py 8 1024 1; python tools/exp_sharded_host_time.py 8 1024 16; PHONIC_SHARD_THREADS=0 python tools/exp_sharded_host_time.py 8 1024 1 ) > $O/r05_sharded_host_time.jsonl 2>> $O/err.log; echo "sharded rc=$?"
( echo "== C3, single-block launches"; PHONIC_LIB=$PWD/tools/ab_libs/diag.so python tools/diag_c3.py 1024 1; echo "== C3, the last block of a 16-block launch"; PHONIC_LIB=$PWD/tools/ab_libs/diag.so python tools/diag_c3.py 1024 16 ) > $O/r05_c3_stamps.txt 2>> $O/err.log; echo "c3 stamps rc=$?"
bash tools/dyn_sweep.sh r05c 5 > $O/dyn_table.txt 2>&1; cp $O/dynamic.jsonl $O/r05_dynamic.jsonl; tail -12 $O/dyn_table.txt
