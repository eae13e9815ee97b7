#!/bin/bash
# round 3, GPU call 33: the long chains of the slices of an 8-way partition: how many, how many fall back
for r in 0 3 5; do GM_PB_HUB_PAR=1 timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $r 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rank $r', d['ms_per_step'], d['config']['hub_rows_in_reference_order'], d['config'].get('final_sweep_error'))"; done
