#!/bin/bash
# round 3: which later 8 GiB stretches of physical memory pair well with the pool?  (GM_PB_GROW_MAP: clean map, one candidate at a time)
OUT=gpurun_out/r03grow2_$(date +%H%M%S); mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
GM_PB_GROW_MAP=14 GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 --steps 5 --warmup 2 2> $OUT/log$rep.err > $OUT/bench$rep.json; python -c "
import json; d=json.loads(open('$OUT/bench$rep.json').read().strip().splitlines()[-1]); print('process $rep:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'])"
grep -a "value stream draw\|grow map" $OUT/log$rep.err | head -22 | cut -c16-150
done
