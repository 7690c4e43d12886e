#!/bin/bash
# (box) the two "box states" (VERDICT r04 weak 3): the default bench line several times in a row on one box, with other leg lengths, and right behind
# a few seconds of continuous load — clocks, power and the power limit's throttle residency in every line
O=gpurun_out/r05_states; mkdir -p $O
A="--no-cpu-baseline --strong-c5-voices 0 --no-realtime"
python bench.py --steps 20 --warmup 5 $A > $O/a1_steps20.json 2>> $O/err.log
python bench.py --steps 20 --warmup 5 $A > $O/a2_steps20.json 2>> $O/err.log
python bench.py --steps 40 --warmup 10 $A > $O/b1_steps40.json 2>> $O/err.log
python bench.py --steps 20 --warmup 5 $A > $O/a3_steps20.json 2>> $O/err.log
python bench.py --steps 100 --warmup 20 $A > $O/c1_steps100.json 2>> $O/err.log
python bench.py --steps 20 --warmup 5 --min-seconds 3 $A > $O/a4_steps20_3s.json 2>> $O/err.log
python bench.py --steps 40 --warmup 10 --min-seconds 3 $A > $O/b2_steps40_3s.json 2>> $O/err.log
python bench.py --steps 20 --warmup 2000 $A > $O/a5_steps20_warm2000.json 2>> $O/err.log
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    r, c = d["roofline"], d["config"].get("clocks", {})
    gm = c.get("gpu_metrics") or {}
    print(f"{os.path.basename(f):28s} frac p10/p50/p90 {r.get('frac_p10',0):.3f} {r['frac']:.3f} {r.get('frac_p90',0):.3f}  ms/step {d['ms_per_step']:.4f} legs {d['repeats']['n']:4d} | sclk p50 {c.get('sclk_mhz',{}).get('p50')} power p50 {c.get('socket_power_w',{}).get('p50')} "
          f"ppt {gm.get('ppt_throttled_share')} xcd {gm.get('xcd_sclk_mhz')} uclk {gm.get('uclk_mhz',{}).get('p50') if gm.get('uclk_mhz') else None} umc {gm.get('umc_activity_pct',{}).get('p50') if gm.get('umc_activity_pct') else None} | by_ppt {r.get('by_power_throttle')}")
PY
