#!/bin/bash
# (box) what clock / power / temperature sources an ordinary user can read on a GPU box: sysfs, hwmon, rocm-smi, amd-smi
for d in /sys/class/drm/card*/device; do
  [ -f $d/vendor ] || continue
  echo "== $d vendor $(cat $d/vendor) device $(cat $d/device 2>/dev/null)"
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level gpu_busy_percent mem_busy_percent current_link_speed; do
    [ -r $d/$f ] && { echo "-- $f"; cat $d/$f 2>&1 | head -12; }
  done
  for h in $d/hwmon/hwmon*; do
    echo "-- $h"; ls $h | tr '\n' ' '; echo
    for f in power1_average power1_input power1_cap temp1_input temp2_input temp3_input freq1_input freq2_input; do [ -r $h/$f ] && echo "$f $(cat $h/$f 2>&1)"; done
  done
  ls -la $d/gpu_metrics 2>&1
done
echo "== rocm-smi"; timeout 30 rocm-smi --showclocks --showpower --showtemp --showperflevel --json 2>&1 | head -c 3000; echo
echo "== amd-smi"; timeout 30 amd-smi metric --clock --power --temperature --json 2>&1 | head -c 4000; echo
echo "== amd-smi throttle"; timeout 30 amd-smi metric --throttle --json 2>&1 | head -c 2000; echo
