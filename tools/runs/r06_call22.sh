#!/bin/bash
# round 6, GPU call 22: does a pseudo-random gap behind every bin of the value stream (GM_PB_BIN_GAP, measurement library) change which LEVEL a
# process's bin kernel runs at?  alternating fresh processes on one box, default against gaps of up to 4096 / 65536 entries
OUT=gpurun_out/r06u; mkdir -p $OUT; export TMPDIR=/tmp
export GRAPH_MI355X_LIB=$PWD/graph_amd/libgraph_mi355x_measure.so
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); v = d['config']['value_stream_placement']; print('$1', d['ms_per_step'], d['roofline']['frac'], v.get('level'), v.get('draw_best_us'), v.get('draw_worst_us'), v.get('draws_timed'))"; }
run() { env $1 timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "$1"; }
for i in 1 2 3 4 5; do run X=1; run GM_PB_BIN_GAP=4096; run GM_PB_BIN_GAP=65536; done | tee $OUT/bin_gap.txt
