#!/bin/bash
# round 4, final records (b): every rank's slice of a 2 / 4 / 8-way partition of the scale-26 graph timed alone on one GPU
# (EMULATION: tools/partition_emulated.py), WCC / SSSP / triangle count with the sequential checkers and the threaded CPU legs
OUT=gpurun_out/r04fb; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python tools/partition_emulated.py --scale 26 --parts 1,2,4,8 > $OUT/partition_emulated.json 2> $OUT/partition.err; tail -c 1500 $OUT/partition_emulated.json; tail -3 $OUT/partition.err
timeout 900 python tools/bench_algos.py --oracle 1 > $OUT/algos.json 2> $OUT/algos.err; python -c "
import json; d = json.loads(open('$OUT/algos.json').read().strip().splitlines()[-1])
for k in ('wcc', 'sssp', 'tc'): print(k, round(d[k]['ms'], 3), d[k]['roofline']['frac'], d[k]['parity'], d[k]['cpu_baseline']['seconds'])
print(d.get('page_rank_api'))"
