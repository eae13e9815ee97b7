#!/bin/bash
# round 4, GPU call 22: bin-kernel chunk size, finer (call 21: 24576 entries 2.62-2.64 ms against 2.71 for 32768 on one box)
export TMPDIR=/tmp
for rep in 1 2; do for ch in 32768 24576 20480 28672 22528; do
  GM_PB_CHUNK=$ch timeout 120 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('chunk $ch:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('draw_best_us'), d['config']['value_stream_placement'].get('level'))"
done; done
for sc in 22 24; do for ch in 32768 24576; do
  GM_PB_CHUNK=$ch timeout 120 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc chunk $ch:', d['ms_per_step'], d['roofline']['frac'])"
done; done
