#!/bin/bash
# round 4, GPU call 53: segments padded to 8 entries against 4 in alternating fresh processes: is the 8-entry layout less
# sensitive to the placement level of the process?
OUT=gpurun_out/r04zv; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); p = d['config']['value_stream_placement']; print('$1:', d['ms_per_step'], d['roofline']['frac'], p['level'], 'draws', p['draws_timed'], p['draw_best_us'], p['draw_worst_us'])"; }
for rep in 1 2 3 4; do for pad in 4 8; do
  GM_PB_SEGPAD=$pad timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "pad $pad"
done; done
