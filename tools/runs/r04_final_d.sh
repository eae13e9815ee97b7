#!/bin/bash
# round 4, final records (d) on the final library: the sweep timeline and the scale-22 / 24 lines
OUT=gpurun_out/r04fd; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/tl -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --steps 10 > $OUT/tl.log 2>&1
python tools/timeline.py $OUT/tl 2 > $OUT/timeline26.txt 2>&1; cat $OUT/timeline26.txt
for sc in 22 24; do timeout 300 python bench.py --scale $sc --algos 0 > $OUT/bench_scale$sc.json 2>/dev/null; tail -1 $OUT/bench_scale$sc.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc:', d['ms_per_step'], d['roofline']['frac'], d['config']['parity']['max_rel_vs_reference'])"; done
find $OUT -name "*.db" -delete
