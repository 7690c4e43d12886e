#!/bin/bash
# (box) which command kind costs what: the dynamic leg with one kind of command at a time
O=gpurun_out/r05_dynk; mkdir -p $O; : > $O/dyn_kinds.jsonl
for k in 0 1 2; do python bench.py --workload dyn --events 480 --dyn-kinds $k --dyn-seconds 3 >> $O/dyn_kinds.jsonl 2>> $O/err.log; done
python bench.py --workload dyn --events 48 --dyn-seconds 3 >> $O/dyn_kinds.jsonl 2>> $O/err.log
python - $O/dyn_kinds.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]; y = d["dyn"]
    for tag in ("offline", "realtime"):
        x = y[tag]
        print(f"ev {c['events_per_s']:g} {tag:8s} ms/step {x['ms_per_step']:.4f} (steady {y['steady_' + tag + '_ms_per_step']:.4f}) defer {x['deferred_share']:.4f} generic ms/launch {x['generic_ms_per_launch']:.4f} launches {x['generic_launches']} fast ms/blk {x['fast_kernel_ms_per_block']:.4f} bpl {x['fast_blocks_per_launch']:.1f} cmds/blk {x['commands_per_block']:.2f}")
PY
