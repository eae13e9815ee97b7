#!/bin/bash
# round 5, GPU call 28: how many terms a hub group of an 8-way rank holds (GM_PB_HUB_GROUP; default max(m/B, 65536) leaves
# rank 0 with ~390 walked groups of ~4.6 rows: the lane walks are the longest piece of its accumulate phase)
OUT=gpurun_out/r05x; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$1:', d['ms_per_step'], c['hub_rows_in_reference_order'], c['final_sweep_error'])"; }
for g in 0 262144 1048576 0 524288; do for r in 0 1; do GM_PB_HUB_GROUP=$g timeout 600 python bench.py --emulate-parts 8 --emulate-rank $r --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "rank $r of 8, group $g"; done; done
