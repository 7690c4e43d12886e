#!/bin/bash
# (box) bench.py --workload dyn at the settings DESIGN quotes -> gpurun_out/<tag>/dynamic.jsonl (one JSON line per setting) + a summary table
# usage: bash tools/dyn_sweep.sh [tag] [seconds of audio per span]
T=${1:-r05}; S=${2:-5}
O=gpurun_out/$T; mkdir -p $O; : > $O/dynamic.jsonl
run() { python bench.py --workload dyn --dyn-seconds $S "$@" >> $O/dynamic.jsonl 2>> $O/dyn.err || echo "dyn $* failed"; }
run --events 0 --churn 0 --silent 0
run --events 48
run --events 480
run --events 480 --dyn-kinds 0
run --events 480 --dyn-kinds 1
run --events 480 --dyn-kinds 2
run --churn 1
run --churn 10
run --silent 50
run --silent 90
run --events 480 --churn 1 --silent 25
python - $O/dynamic.jsonl <<'PY'
import json, sys
print("events/s kinds churn%/s silent% | offline ms/step (x steady) | realtime ms/step (x steady) | deferred share | generic ms/launch | fast ms/blk (blocks/launch) | cmds/blk")
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]; y = d["dyn"]; o, r = y["offline"], y["realtime"]
    print(f"{c['events_per_s']:7g} {c.get('kinds','012'):>5s} {c['churn_pct_per_s']:6g} {c['silent_pct']:7g} | {o['ms_per_step']:.4f} (x{y['ratio_offline']:.2f}) | {r['ms_per_step']:.4f} (x{y['ratio_realtime']:.2f}) | "
          f"{o['deferred_share']:.4f} | {o['generic_ms_per_launch']:.4f} | {o['fast_kernel_ms_per_block']:.4f} ({o['fast_blocks_per_launch']:.1f}) | {o['commands_per_block']:.2f}")
PY
