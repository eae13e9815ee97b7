#!/bin/bash
# round 6, GPU call 26: an emulated rank of 8 with more LONG hub rows (GM_PB_HUB_LONG) and with the hub rows summed beside
# their own part (GM_PR_PART_HUBS=1), alone and together
OUT=gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'), 'groups', h.get('hub_groups'), 'long', h.get('long_rows'), 'parity', d['config'].get('parity'))"; }
for rank in 0 6; do
for ph in 0 1; do for hl in 0 24576 16384 12288 8192; do
GM_PR_PART_HUBS=$ph GM_PB_HUB_LONG=$hl timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $rank 2>> $OUT/bench.err | tail -1 | line "rank $rank of 8 part_hubs=$ph hub_long=$hl"
done; done; done
for v in "0 0" "1 0" "1 12288"; do set -- $v
GM_PR_PART_HUBS=$1 GM_PB_HUB_LONG=$2 timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t_$1_$2 -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 > $OUT/t.log 2>&1
python tools/timeline.py $OUT/t_$1_$2 2 > $OUT/timeline_$1_$2.txt 2>&1; echo "== part_hubs=$1 hub_long=$2"; head -18 $OUT/timeline_$1_$2.txt | cut -c1-100
done
find $OUT -name "*.db" -delete
