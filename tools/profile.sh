#!/bin/bash
# usage: tools/profile.sh <outdir> [bench args]: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of bench.py
# (rocprofv3; counters in their own passes with --kernel-trace only).  Run inside gpurun.
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 "$@" > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 30 > $OUT/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sweeps 0 "$@" > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_collect.py $OUT/pmc_raw.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
tail -1 $OUT/trace.log; head -12 $OUT/kernel_stats.txt
find $OUT -name "*.db" -size +20M -delete
