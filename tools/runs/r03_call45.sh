#!/bin/bash
# round 3, GPU call 45: candidate sets of different kinds for the value stream (pool spread / oldest / newest / spread, then fresh stretches):
# three fresh processes with the draws logged, then the arena + parity tests
OUT=gpurun_out/r03zi_$(date +%H%M%S); mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/log$rep.err > $OUT/bench$rep.json; python -c "
import json; d=json.loads(open('$OUT/bench$rep.json').read().strip().splitlines()[-1]); print('process $rep:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'], d['config']['plan_build_ms'])"
grep -a "value stream draw" $OUT/log$rep.err | cut -c16-120 | head -16
done
