#!/bin/bash
# round 5, GPU call 29: an 8-way rank's lane walks outlast its accumulate kernels by ~60 us (profiles/r05_rank0_of_8_sweep_timeline.txt).
# Environment-only probes: which rows are long (GM_PB_HUB_LONG), LDS left beside an accumulate workgroup (GM_PB_HUB_ROOM)
OUT=gpurun_out/r05x; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; h = c['hub_rows_in_reference_order']; print('$1:', d['ms_per_step'], h['hub_groups'], h['long_rows'], h['hub_seq_blocks'], c['final_sweep_error'])"; }
for cfg in "X=1" "GM_PB_HUB_LONG=8192" "GM_PB_HUB_LONG=10240" "GM_PB_HUB_LONG=6144" "GM_PB_HUB_ROOM=34304" "GM_PB_HUB_ROOM=34304 GM_PB_HUB_LONG=8192" "X=1" "GM_PB_HUB_LONG=8192" "GM_PB_HUB_LONG=10240" "GM_PB_HUB_ROOM=34304"; do for r in 0 1; do env $cfg timeout 600 python bench.py --emulate-parts 8 --emulate-rank $r --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "rank $r of 8, $cfg"; done; done
