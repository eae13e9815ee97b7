#!/bin/bash
# round 4, GPU call 47: where the SSSP plan's time goes at scale 26 / 24
OUT=gpurun_out/r04zp; mkdir -p $OUT; export TMPDIR=/tmp
for sc in 24 26; do
  GM_SSSP_TIMES=1 timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 --sssp-scale $sc > $OUT/sssp.json 2> $OUT/sssp.err
  echo "scale $sc"; grep "sssp:" $OUT/sssp.err | head -4
done
