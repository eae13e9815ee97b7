#!/bin/bash
# round 3, GPU call 37: size of the hub groups (entries per group) against the sweep time at scale 22 / 24 / 26 and on a slice
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['config']['hub_rows_in_reference_order']; print('$1', d['ms_per_step'], d['roofline']['frac'], h['hub_groups'], h['long_chain_groups'], h['long_chain_blocks'])"; }
for s in 22 24 26; do for g in 0 131072 65536 32768; do GM_PB_HUB_GROUP=$g timeout 300 python bench.py --scale $s --cpu-sweeps 0 2>/dev/null | tail -1 | line "scale $s group $g:"; done; done
for g in 0 65536; do GM_PB_HUB_GROUP=$g timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 3 2>/dev/null | tail -1 | line "8 parts rank 3 group $g:"; done
