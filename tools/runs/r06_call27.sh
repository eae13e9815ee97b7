#!/bin/bash
# round 6, GPU call 27: an emulated rank of 8 / 4 with the hub kernels' streams at the default and at the highest priority
# (measurement library: GM_PB_SIDE_PRIO = 1 lowest (product) / 0 default / 2 highest), and the walking wavefront's s_setprio off
OUT=gpurun_out/r06z; mkdir -p $OUT; export TMPDIR=/tmp
export GRAPH_MI355X_LIB=$PWD/graph_amd/libgraph_mi355x_measure.so
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'))"; }
for cfg in "8 0" "8 6" "4 0"; do set -- $cfg
for sp in 1 0 2 1 0 2; do
GM_PB_SIDE_PRIO=$sp timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts $1 --emulate-rank $2 2>> $OUT/bench.err | tail -1 | line "rank $2 of $1 side_prio=$sp"
done; done
for sp in 0 2; do
GM_PB_SIDE_PRIO=$sp timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t_$sp -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 > $OUT/t.log 2>&1
python tools/timeline.py $OUT/t_$sp 2 > $OUT/timeline_$sp.txt 2>&1; echo "== side_prio=$sp"; head -16 $OUT/timeline_$sp.txt | cut -c1-100
done
find $OUT -name "*.db" -delete
