#!/bin/bash
# round 3: how far must the arena grow before a value stream split between old and new pieces is fast?  growth forced through 96 GiB
OUT=gpurun_out/r03grow_$(date +%H%M%S); mkdir -p $OUT; export TMPDIR=/tmp
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/log0.err > $OUT/bench0.json; python -c "
import json; d=json.loads(open('$OUT/bench0.json').read().strip().splitlines()[-1]); print('default:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'])"
grep -a "value stream draw" $OUT/log0.err | head -9 | cut -c16-110
T0=$(date +%s.%N)
GM_PB_BW_MIN=9999 GM_PB_GROW_GIB=96 GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/log1.err > $OUT/bench1.json; python -c "
import json; d=json.loads(open('$OUT/bench1.json').read().strip().splitlines()[-1]); print('forced growth:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'])"
T1=$(date +%s.%N); echo "wall $(python -c "print(round($T1-$T0,1))") s"
grep -a "value stream draw" $OUT/log1.err | head -39 | cut -c16-110
