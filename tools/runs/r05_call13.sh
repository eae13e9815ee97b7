#!/bin/bash
# round 5, GPU call 13: kernel times of the triangle count's FIRST call (tc_symmetry_kernel flattened)
OUT=gpurun_out/r05m; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,sssp > $OUT/rec.json 2> $OUT/err.txt
python tools/algos_profile.py $OUT/rec.json $OUT/t 2>&1 | grep -a "tc_\|##" | head -20
find $OUT -name "*.db" -size +8M -delete
