#!/bin/bash
# round 4, GPU call 50: does the placement search find the fast level more often when it keeps looking below 4000 GB/s (GM_PB_BW_MIN)?
OUT=gpurun_out/r04zs; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); p = d['config']['value_stream_placement']; print('$1:', d['ms_per_step'], d['roofline']['frac'], p['level'], 'draws', p['draws_timed'], p['draw_best_us'], p['draw_worst_us'], 'grown', p['arena_grown_pieces'], 'plan ms', d['config']['plan_build_ms'])"; }
for rep in 1 2 3 4 5; do for bw in 3700 4000; do
  GM_PB_BW_MIN=$bw timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "bw_min $bw"
done; done
