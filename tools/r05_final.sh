#!/bin/bash
# (box) round 5, the measured files behind DESIGN's table that tools/final_round.sh does not take: the full -m gpu suite, the callback-size sweep
# (128 ... 4096 frames), the dynamic legs, the per-effect table, other layouts, the sharded handle's host time, what one commanded unit costs
O=gpurun_out/r05_final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for b in 128 256 512 1024 2048 4096; do
  python bench.py --steps $((40960 / b > 160 ? 160 : 40960 / b)) --warmup 10 --block $b --no-cpu-baseline --strong-c5-voices 0 > $O/r05_headline_block${b}_bench.json 2>> $O/err.log; echo "block $b rc=$?"
done
python tools/per_effect.py 1024 > $O/r05_per_effect.jsonl 2>> $O/err.log; echo "per_effect rc=$?"
python tools/exp_generic_paths.py 1024 headline,stream,resampled,nested > $O/r05_generic_paths.jsonl 2>> $O/err.log; echo "generic_paths rc=$?"
( python tools/exp_sharded_host_time.py 8 1024 1; PHONIC_SHARD_DIRECT=0 python tools/exp_sharded_host_time.py 8 1024 1; python tools/exp_sharded_host_time.py 8 1024 16; PHONIC_SHARD_THREADS=0 python tools/exp_sharded_host_time.py 8 1024 1 ) > $O/r05_sharded_host_time.jsonl 2>> $O/err.log; echo "sharded rc=$?"
python tools/diag_cmd.py 1024 96 > $O/r05_commanded_unit_now.txt 2>> $O/err.log; echo "diag_cmd rc=$?"
bash tools/dyn_sweep.sh r05_final 5 > $O/dyn_table.txt 2>&1; cp $O/dynamic.jsonl $O/r05_dynamic.jsonl; tail -12 $O/dyn_table.txt
