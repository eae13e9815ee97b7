#!/bin/bash
# round 5, GPU call 17: why did the emulated 2-way rank go from 1.5 to 2.0 ms?
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$1:', d['ms_per_step'], 'hot', c['hot_sources'], c['hot_tiers'], c['hot_edges'], 'values', c['value_entries'], c['value_stream_placement'])"; }
for t in 1 0 1 0; do GM_PB_HOT_TRIM=$t timeout 600 python bench.py --emulate-parts 2 --emulate-rank 0 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "rank 0 of 2, trim $t"; done
GM_PB_HOT_TRIM=0 timeout 600 python bench.py --emulate-parts 4 --emulate-rank 0 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "rank 0 of 4, trim 0"
timeout 600 python bench.py --emulate-parts 4 --emulate-rank 0 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "rank 0 of 4, trim 1"
