#!/bin/bash
# round 4, GPU call 54: does a longer untimed clock ramp (--prewarm-ms) change the level of a fresh box's first processes?
OUT=gpurun_out/r04zw; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); p = d['config']['value_stream_placement']; print('$1:', d['ms_per_step'], d['roofline']['frac'], p['level'], 'draws', p['draws_timed'], p['draw_best_us'], p['draw_worst_us'])"; }
for rep in 1 2 3; do for pw in 400 6000; do
  timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --prewarm-ms $pw 2>/dev/null | tail -1 | line "prewarm $pw"
done; done
