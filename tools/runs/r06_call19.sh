#!/bin/bash
# round 6, GPU call 19: plan knobs of the product library against the defaults at scale 26 on one box, alternating (is any of them worth a
# change on today's boxes?): source tiles of 32768, 5 / 6 / 9 hot tiers, chunk sizes, rows per bin 13
OUT=gpurun_out/r06r; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), d['config']['value_stream_placement'].get('draw_best_us'), d['config'].get('hot_tiers'))"; }
run() { env $1 timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "$1"; }
for cfg in X=1 GM_PB_SLOG=15 X=1 GM_PB_TIERS=5 GM_PB_TIERS=6 X=1 GM_PB_TIERS=9 GM_PB_CHUNK=16384 GM_PB_CHUNK=32768 X=1 GM_PB_RB=13 GM_PB_SLOG=15 X=1; do run $cfg; done
