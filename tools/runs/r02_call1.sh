#!/bin/bash
# round 2, GPU call 1: robustness tests, PageRank parity at scale 24/26 with the round-1 engine (the "before"
# numbers), full-size oracle comparisons + rooflines of WCC / SSSP / TC, kernel stats + PMC traffic of those.
OUT=gpurun_out/r02a; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null; free -g >> $OUT/host.txt
timeout 600 python -m pytest tests/test_gpu_robustness.py -m gpu -x -q > $OUT/pytest_robust.log 2>&1; tail -3 $OUT/pytest_robust.log
timeout 400 python tools/parity_pagerank.py --scale 24 --mode pb > $OUT/parity24_pb.json 2> $OUT/parity24.err; tail -c 600 $OUT/parity24_pb.json
timeout 700 python tools/parity_pagerank.py --scale 26 --mode pb > $OUT/parity26_pb.json 2> $OUT/parity26.err; tail -c 600 $OUT/parity26_pb.json
timeout 900 python tools/bench_algos.py > $OUT/algos.json 2> $OUT/algos.err; tail -c 1500 $OUT/algos.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/bench_algos.py --profile 1 > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 40 > $OUT/algos_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python tools/bench_algos.py --profile 1 > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_collect.py $OUT/algos_pmc_raw.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
head -30 $OUT/algos_kernel_stats.txt
find $OUT -name "*.db" -size +20M -delete
