#!/bin/bash
# round 4, GPU call 36: SSSP scale 24, edges streamed per round (how small are the small rounds?)
OUT=gpurun_out/r04ze; mkdir -p $OUT; export TMPDIR=/tmp
GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2> $OUT/stats.err > $OUT/stats.json
grep "sssp round" $OUT/stats.err | tail -100 | awk '{print $3, $6, $8, $9, $10, $11, $12}' | tr '\n' ';'
echo; grep "sssp:" $OUT/stats.err | tail -2
