#!/bin/bash
# round 4, GPU call 51: one bounded experiment: does a process that allocates and frees most of the device memory first run at
# the level later processes of a box run at?  (default, default, pretouch, default, default)
OUT=gpurun_out/r04zt; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); p = d['config']['value_stream_placement']; print('$1:', d['ms_per_step'], d['roofline']['frac'], p['level'], 'draws', p['draws_timed'], p['draw_best_us'], p['draw_worst_us'])"; }
for cfg in 0 0 0.9 0 0; do
  timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --pretouch-frac $cfg 2> $OUT/err.log | tail -1 | line "pretouch $cfg"; grep pretouch $OUT/err.log
done
