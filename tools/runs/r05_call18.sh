#!/bin/bash
# round 5, GPU call 18: timeline of an emulated 2-way rank
OUT=gpurun_out/r05p; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t2 -o t -- python bench.py --emulate-parts 2 --emulate-rank 0 --cpu-sweeps 0 --algos 0 > $OUT/t2.log 2>&1
python tools/timeline.py $OUT/t2 1 | cut -c1-130
find $OUT -name "*.db" -delete
