#!/bin/bash
# round 6, GPU call 23: the hub rows' index space regrouped (walked rows first, long rows behind: groups no longer end at every long row) —
# whole GPU suite, then A/B (GM_PB_HUB_REGROUP=0/1): one GPU at scale 26 / 22, emulated ranks 0 / 1 / 6 of 8 and rank 0 of 4
OUT=gpurun_out/r06v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -4
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'), 'groups', h.get('hub_groups'), 'long', h.get('long_rows'))"; }
for r in 0 1 0 1; do GM_PB_HUB_REGROUP=$r timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 regroup=$r"; done
for r in 0 1 0 1; do GM_PB_HUB_REGROUP=$r timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 regroup=$r"; done
for rank in 0 1 6; do for r in 0 1; do GM_PB_HUB_REGROUP=$r timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $rank 2>> $OUT/bench.err | tail -1 | line "rank $rank of 8 regroup=$r"; done; done
for r in 0 1; do GM_PB_HUB_REGROUP=$r timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 4 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 4 regroup=$r"; done
