#!/bin/bash
# Samples every GPU of the node (sysfs DPM clocks, power) while tools/timeseries.py runs 4 s of sweeps:
# which card is ours, at what clocks, and what the neighbours are doing.
# sample every GPU's sclk/mclk/power through sysfs while a 4 s sweep series runs; print per-card busy maxima
( for i in $(seq 1 60); do
    for c in /sys/class/drm/card*/device; do
      [ -f $c/pp_dpm_sclk ] || continue
      s=$(grep -h '\*' $c/pp_dpm_sclk | sed 's/.*: \([0-9]*\)Mhz.*/\1/'); f=$(grep -h '\*' $c/pp_dpm_fclk 2>/dev/null | sed 's/.*: \([0-9]*\)Mhz.*/\1/'); m=$(grep -h '\*' $c/pp_dpm_mclk 2>/dev/null | sed 's/.*: \([0-9]*\)Mhz.*/\1/')
      p=$(cat $c/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
      t2=$(cat $c/hwmon/hwmon*/temp2_input 2>/dev/null | head -1); t3=$(cat $c/hwmon/hwmon*/temp3_input 2>/dev/null | head -1)
      echo "$(basename $(dirname $c)) $s $m $f $((p/1000000)) $((t2/1000)) $((t3/1000))"
    done
    sleep 0.1
  done ) > /tmp/cw_samples.txt &
W=$!
timeout -s KILL 100 python tools/timeseries.py 26 1200 2>&1 | grep ms/sweep | tail -2
wait $W
cat /sys/class/drm/card*/device/hwmon/hwmon*/temp2_label /sys/class/drm/card*/device/hwmon/hwmon*/temp3_label 2>/dev/null | sort | uniq -c
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('/tmp/cw_samples.txt'):
    p=l.split()
    if len(p)>=7 and p[1].isdigit(): d[p[0]].append((int(p[1]),p[2],p[3],int(p[4]),int(p[5]),int(p[6])))
for c,v in sorted(d.items()):
    busy=[x for x in v if x[0]>1000]
    print(c, "samples",len(v),"busy",len(busy), "sclk busy min/med/max", (min(b[0] for b in busy), sorted(b[0] for b in busy)[len(busy)//2], max(b[0] for b in busy)) if busy else None, "mclk",set(x[1] for x in v),"fclk",set(x[2] for x in v),"power max",max(x[3] for x in v),"temp2/temp3 max",max(x[4] for x in v),max(x[5] for x in v))
PY
