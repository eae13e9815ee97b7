#!/bin/bash
# round 6, GPU call 30: smaller lane-walk groups on WHOLE graphs (measurement library, GM_PB_HUB_GROUP = terms per group; default
# max(65536, m / bins) = 262144 at scale 26, 65536 at scale 22): the synchronous sweep bench.py times against the block-Gauss-Seidel
# default call, whose 8 hub launches per iteration are its critical path (400 us of a block's 590 at scale 26)
OUT=gpurun_out/r06ac; mkdir -p $OUT; export TMPDIR=/tmp
export GRAPH_MI355X_LIB=$PWD/graph_amd/libgraph_mi355x_measure.so
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'), 'groups', h.get('hub_groups'))"; }
for rep in 1 2; do for hg in 0 131072 65536 32768; do
GM_PB_HUB_GROUP=$hg timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 sync sweep hub_group=$hg"
done; done
for hg in 0 131072 65536 32768; do GM_PB_HUB_GROUP=$hg timeout 300 python tools/gs_time.py 26 2>> $OUT/gs.err | tail -1 | cut -c1-200; done
for rep in 1 2; do for hg in 0 32768 16384; do
GM_PB_HUB_GROUP=$hg timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 sync sweep hub_group=$hg"
done; done
for hg in 0 32768 16384; do GM_PB_HUB_GROUP=$hg timeout 300 python tools/gs_time.py 22 2>> $OUT/gs.err | tail -1 | cut -c1-200; done
