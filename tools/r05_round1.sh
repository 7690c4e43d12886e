#!/bin/bash
# (box) round 5, first measurement call: the new parity tests, the default bench line with clocks, the callback-size sweep, the dynamic legs
set -o pipefail
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q -k "deferred_bus_words or dynamic_workload or sharded_host_write or sharded_graph_object" > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_new.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for b in 128 256 512 1024; do
  python bench.py --steps 40 --warmup 10 --block $b --no-cpu-baseline --strong-c5-voices 0 > $O/headline_block${b}.json 2> $O/headline_block${b}.err; echo "block $b rc=$?"
done
: > $O/dynamic.jsonl
python bench.py --workload dyn --events 0 --churn 0 --silent 0 --dyn-seconds 5 >> $O/dynamic.jsonl 2> $O/dyn.err; echo "dyn0 rc=$?"
python bench.py --workload dyn --events 480 --dyn-seconds 5 >> $O/dynamic.jsonl 2>> $O/dyn.err; echo "dyn events rc=$?"
python bench.py --workload dyn --churn 1 --dyn-seconds 5 >> $O/dynamic.jsonl 2>> $O/dyn.err; echo "dyn churn rc=$?"
python bench.py --workload dyn --silent 50 --dyn-seconds 5 >> $O/dynamic.jsonl 2>> $O/dyn.err; echo "dyn silent rc=$?"
python bench.py --workload dyn --events 480 --churn 1 --silent 25 --dyn-seconds 5 >> $O/dynamic.jsonl 2>> $O/dyn.err; echo "dyn all rc=$?"
python tools/per_effect.py 1024 > $O/per_effect.jsonl 2> $O/per_effect.err; echo "per_effect rc=$?"
