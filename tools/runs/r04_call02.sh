#!/bin/bash
# round 4, GPU call 2: pb_hubseq_kernel (exact left-to-right row sums for hub groups of >= 3 rows): the whole GPU suite, the
# default bench line, kernel trace
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -12 $OUT/pytest.txt | cut -c1-250
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-1500
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 12 > $OUT/kernel_stats.txt; cut -c1-150 $OUT/kernel_stats.txt
find $OUT -name "*.db" -size +20M -delete
