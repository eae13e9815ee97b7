#!/bin/bash
# round 6, GPU call 24: the dispatches of one sweep of an emulated rank of 8 (slices' hub rows regrouped, parts in order), ranks 0 and 6
OUT=gpurun_out/r06w; mkdir -p $OUT; export TMPDIR=/tmp
for rank in 0 6; do
  timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t$rank -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $rank > $OUT/t$rank.log 2>&1
  python tools/timeline.py $OUT/t$rank 2 > $OUT/timeline_rank$rank.txt 2>&1; cat $OUT/timeline_rank$rank.txt | cut -c1-150 | head -34; tail -1 $OUT/t$rank.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('rank $rank', d['ms_per_step'], d['config']['hub_rows_in_reference_order'])"
done
find $OUT -name "*.db" -delete
