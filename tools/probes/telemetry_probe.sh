#!/bin/bash
# What GPU telemetry an ordinary user can read on a gpurun box without touching HIP (round 6).
set +e
echo "== id"; id
echo "== drm cards"; ls -d /sys/class/drm/card*/device 2>&1
for d in /sys/class/drm/card*/device; do
  [ -e "$d/vendor" ] || continue
  echo "== $d vendor $(cat $d/vendor) device $(cat $d/device 2>/dev/null)"
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk gpu_busy_percent mem_busy_percent power_dpm_force_performance_level current_link_speed current_link_width mem_info_vram_used unique_id; do
    [ -r "$d/$f" ] && { echo "-- $f"; cat "$d/$f" 2>&1 | head -12; }
  done
  for h in $d/hwmon/hwmon*; do
    echo "-- $h"; ls $h 2>&1 | tr '\n' ' '; echo
    for f in power1_average power1_input power1_cap power1_cap_max temp1_input temp2_input temp3_input temp1_label temp2_label temp3_label freq1_input freq2_input freq1_label freq2_label; do
      [ -r "$h/$f" ] && echo "   $f = $(cat $h/$f 2>&1)"
    done
  done
  [ -r "$d/gpu_metrics" ] && { echo "-- gpu_metrics bytes: $(wc -c < $d/gpu_metrics)"; head -c 16 $d/gpu_metrics | xxd | head -2; }
done
echo "== amdsmi python"; python3 -c "import amdsmi; print(amdsmi.__file__)" 2>&1 | tail -1
ls /opt/rocm/share/amd_smi 2>&1 | head
echo "== rocm-smi"; timeout 30 rocm-smi --showclocks --showpower --showtemp --json 2>&1 | head -c 3000
echo
echo "== amd-smi metric"; timeout 30 amd-smi metric --json 2>&1 | head -c 4000
