#!/bin/bash
# round 4, GPU call 55: scale 22 has as many bins as the chip has CUs: the accumulate kernel lasts as long as its longest item,
# and bins are only split beyond TWICE the average (GM_PB_SPLIT = entries per item).  Smaller limits:
OUT=gpurun_out/r04zx; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['workgroups_per_sweep'], d['config']['value_entries'], d['config']['hot_edges'])"; }
for sc in 22 24; do
for sp in 0 400000 300000 250000 200000 150000 100000; do
  GM_PB_SPLIT=$sp timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc split $sp"
done; done
GM_PB_SPLIT=200000 timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale 22 --steps 10 > $OUT/trace.log 2>&1
python tools/timeline.py $OUT/trace 1; rm -rf $OUT/trace
