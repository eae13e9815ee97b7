#!/bin/bash
# usage: tools/clockwatch.sh <out.log> <command...>: samples the GPU's current DPM clocks (sysfs) every
# 100 ms while the command runs; command output lines are time-stamped into the same log.
OUT=$1; shift
D=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
echo "device dir: $D" > $OUT
( while true; do
    s=$(grep -h '\*' $D/pp_dpm_sclk 2>/dev/null | tr -d '\n'); f=$(grep -h '\*' $D/pp_dpm_fclk 2>/dev/null | tr -d '\n')
    m=$(grep -h '\*' $D/pp_dpm_mclk 2>/dev/null | tr -d '\n')
    p=$(cat $D/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
    echo "clk $(date +%s.%N | cut -c1-14) sclk[$s] mclk[$m] fclk[$f] power=$p"
    sleep 0.1
  done ) >> $OUT 2>&1 &
W=$!
PYTHONUNBUFFERED=1 stdbuf -oL "$@" 2>&1 | while IFS= read -r l; do echo "run $(date +%s.%N | cut -c1-14) $l"; done >> $OUT
kill $W 2>/dev/null
