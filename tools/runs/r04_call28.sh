#!/bin/bash
# round 4, GPU call 28: hot-record batch size of the accumulate kernel (uint4 per lane and batch: 1 / 2 / 3), same box, alternating;
# tiers 5 / 6 / 7 / 9
export TMPDIR=/tmp
for rep in 1 2 3; do for hu in 2 1 3; do
  GM_PB_ACC_HU=$hu timeout 200 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('HU $hu:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('draw_best_us'))"
done; done
for t in 5 6 9; do GM_PB_TIERS=$t timeout 200 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('tiers $t:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('draw_best_us'), d['config']['hot_edges'])"; done
