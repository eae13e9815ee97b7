#!/bin/bash
# round 3, GPU call 34: every slice of an 8-way partition, long chains walked block-parallel or not (emulated slots hold typical values)
for r in 0 1 2 3 4 5 6 7; do for p in 0 1; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $r 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['config']['hub_rows_in_reference_order']; print('rank $r par $p', d['ms_per_step'], h['hub_groups'], h['long_chain_groups'], h['long_chain_blocks'], h['long_chains_fell_back'])"; done; done
